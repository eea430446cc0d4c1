// Library-level entry points: version, error text, pitch helper and the fused
// front-end that chains the per-op launchers on one stream.
#include "common.hpp"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void ipa_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

IPA_EXPORT int ipa_version(void) { return 100; }

// ---- per-stage timing ---------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct ProfRec {
    int stage;
    hipEvent_t a, b;
    bool ok;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::vector<ProfRec> g_prof_open;
}  // namespace

void prof_begin(int stage, hipStream_t st)
{
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.stage = stage;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    r.ok = hipEventRecord(r.a, st) == hipSuccess;
    g_prof_open.push_back(r);
}

void prof_end(int stage, hipStream_t st)
{
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = g_prof_open.size(); i-- > 0;)
        if (g_prof_open[i].stage == stage) {
            if (hipEventRecord(g_prof_open[i].b, st) != hipSuccess) g_prof_open[i].ok = false;
            g_prof_recs.push_back(g_prof_open[i]);
            g_prof_open.erase(g_prof_open.begin() + (long)i);
            return;
        }
}

IPA_EXPORT int ipa_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return IPA_OK;
}

IPA_EXPORT int ipa_profile_stage_count(void) { return PROF_NSTAGES; }

IPA_EXPORT const char* ipa_profile_stage_name(int stage)
{
    static const char* names[PROF_NSTAGES] = {"gray_thresh_pack", "open_vertical",  "open_horizontal",
                                              "gaussian_blur",    "canny_nms",      "canny_hysteresis",
                                              "hough_compact",    "hough_vote",     "hough_peaks"};
    return (stage >= 0 && stage < PROF_NSTAGES) ? names[stage] : "";
}

IPA_EXPORT int ipa_profile_collect(double* ms, int* launches)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < PROF_NSTAGES; i++) {
        ms[i] = 0;
        launches[i] = 0;
    }
    for (auto& r : g_prof_recs) {
        float t = 0;
        if (r.ok && hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms[r.stage] += t;
            launches[r.stage]++;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof_recs.clear();
    return IPA_OK;
}
IPA_EXPORT const char* ipa_last_error(void) { return g_err; }
IPA_EXPORT size_t ipa_pitch_for(size_t row_bytes) { return align_up(row_bytes ? row_bytes : 1, 256); }

int launch_pack_bits(Plane src, u32* bits, size_t bpw, size_t bsw, int w, int h, int batch, int32_t* vmax,
                     hipStream_t st);
int launch_gauss5(Plane src, Plane dst, int w, int h, int batch, void* ws, size_t ws_bytes, hipStream_t st);

namespace {
struct FrontWs {
    size_t bits_bytes, canny_bytes, hough_bytes, blur_ws_bytes, blur_plane_bytes, blur_pitch;
};
FrontWs front_layout(int w, int h, int batch, int max_points)
{
    FrontWs L;
    L.bits_bytes = bits_pitch_words(w) * 4 * (size_t)h * batch;
    L.canny_bytes = canny_ws_bytes(w, h, batch);
    L.hough_bytes = hough_ws_bytes(w, h, batch, max_points);
    L.blur_ws_bytes = ipa_gaussian_blur_workspace_bytes(w, h, batch);
    L.blur_pitch = align_up((size_t)w, 256);
    L.blur_plane_bytes = L.blur_pitch * (size_t)h * batch;
    return L;
}
}  // namespace

static size_t chunk_ws_bytes(int width, int height, int batch, int max_points)
{
    if (width <= 0 || height <= 0 || batch <= 0) return 0;
    FrontWs L = front_layout(width, height, batch, max_points);
    Arena a(nullptr, 0);
    a.take(L.bits_bytes);
    a.take(L.canny_bytes);
    a.take(L.hough_bytes);
    a.take(L.blur_ws_bytes);
    a.take(L.blur_plane_bytes);
    return a.off + 256;
}

// Large batches run as several chunks that alternate between two streams, each chunk starting
// when the previous chunk's gray kernel is done: the chunks stay one stage apart, so that an
// HBM-bound stage of one runs beside an ALU- or LDS-bound stage of the other.  Every chunk has
// its own part of the workspace.
constexpr int IPA_MAX_CHUNKS = 8;
static int pipeline_chunks(int batch)
{
    static int pipe = -1;  // IPA_PIPELINE=<chunks>; 0 or 1: one chain on the caller's stream
    if (pipe < 0) {
        const char* e = getenv("IPA_PIPELINE");
        pipe = e ? atoi(e) : 2;
        if (pipe > IPA_MAX_CHUNKS) pipe = IPA_MAX_CHUNKS;
    }
    // per-stage timing wants the stages one after the other, each over the whole batch
    int k = g_prof_on ? 1 : pipe;
    while (k > 1 && batch / k < 32) k--;  // chunks of at least 32 images
    return k < 1 ? 1 : k;
}
// images [lo, hi) of chunk c out of k
static void chunk_range(int batch, int k, int c, int* lo, int* hi)
{
    const int base = batch / k, rem = batch % k;
    *lo = c * base + (c < rem ? c : rem);
    *hi = *lo + base + (c < rem ? 1 : 0);
}

IPA_EXPORT size_t ipa_front_end_workspace_bytes(int width, int height, int batch, int max_points)
{
    if (width <= 0 || height <= 0 || batch <= 0) return 0;
    size_t most = chunk_ws_bytes(width, height, batch, max_points);
    for (int k = 2; k <= IPA_MAX_CHUNKS; k++) {  // every way of running must fit
        size_t sum = 0;
        for (int c = 0; c < k; c++) {
            int lo, hi;
            chunk_range(batch, k, c, &lo, &hi);
            sum += align_up(chunk_ws_bytes(width, height, hi - lo, max_points), 256);
        }
        if (sum > most) most = sum;
    }
    return most;
}

namespace {
// One helper stream per device: the two opens only depend on the gray kernel and are bound by
// HBM writes, the Canny kernels by the vector ALUs, so they run side by side.  The helper
// stream is joined back into the caller's stream before the call returns.
std::mutex g_side_mu;
hipStream_t g_side_stream[2][64] = {};
// which = 0: stream of the opens, 1: stream of the second half of a pipelined batch
hipStream_t side_stream(int which)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_side_mu);
    hipStream_t& sref = g_side_stream[which][dev];
    if (!sref && hipStreamCreateWithFlags(&sref, hipStreamNonBlocking) != hipSuccess) sref = nullptr;
    return sref;
}
}  // namespace

static int front_end_chunk(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride, uint8_t* gray, uint8_t* thresh,
                           uint8_t* vlines, uint8_t* hlines, uint8_t* edges, size_t plane_pitch, size_t plane_stride,
                           int32_t* accum, size_t accum_stride, int32_t* seg_max, int max_points, int32_t* point_counts,
                           int width, int height, int batch, int thresh_val, int vert_len, int horiz_len, int iterations,
                           double canny_lo, double canny_hi, int blur5x5, void* ws, size_t ws_bytes, hipStream_t st,
                           hipEvent_t after_gray = nullptr)
{
    Plane B = mkplane(bgr, bgr_pitch, bgr_stride);
    Plane G = mkplane(gray, plane_pitch, plane_stride), T = mkplane(thresh, plane_pitch, plane_stride);
    Plane V = mkplane(vlines, plane_pitch, plane_stride), H = mkplane(hlines, plane_pitch, plane_stride);
    Plane E = mkplane(edges, plane_pitch, plane_stride);
    IPA_REQUIRE(plane_ok(B, (size_t)width * 3, batch), IPA_ERR_ALIGN, "front_end: bgr pointer/pitch misaligned");
    IPA_REQUIRE(plane_ok(G, (size_t)width, batch) && plane_ok(T, (size_t)width, batch) &&
                    plane_ok(V, (size_t)width, batch) && plane_ok(H, (size_t)width, batch) &&
                    plane_ok(E, (size_t)width, batch),
                IPA_ERR_ALIGN, "front_end: output plane pointer/pitch misaligned");
    IPA_REQUIRE(!accum || point_counts, IPA_ERR_ARG, "front_end: point_counts is NULL");
    IPA_REQUIRE(ws && ws_bytes >= chunk_ws_bytes(width, height, batch, max_points), IPA_ERR_WORKSPACE,
                "front_end: workspace too small");
    FrontWs L = front_layout(width, height, batch, max_points);
    Arena ar(ws, ws_bytes);
    u32* bits = (u32*)ar.take(L.bits_bytes);
    void* canny_ws = ar.take(L.canny_bytes);
    void* hough_ws = ar.take(L.hough_bytes);
    void* blur_ws = ar.take(L.blur_ws_bytes);
    u8* blur_plane = (u8*)ar.take(L.blur_plane_bytes);
    const size_t bpw = bits_pitch_words(width), bsw = bpw * (size_t)height;
    int tv = thresh_val < -1 ? -1 : (thresh_val > 256 ? 256 : thresh_val);

    // gray + THRESH_BINARY_INV plane + its bit plane, one pass over the BGR input
    int rc;
    {
        ProfScope ps(PROF_GRAY_THRESH, st);
        rc = launch_bgr2gray(B, G, T, bits, bpw, bsw, width, height, batch, tv, 255, 1, st);
    }
    if (rc) return rc;
    if (after_gray) IPA_HIP(hipEventRecord(after_gray, st));
    // open with (1 x vert_len) and (horiz_len x 1), iterations folded into one larger kernel
    const int KV = vert_len + (iterations - 1) * (vert_len - 1), AV = (vert_len / 2) * iterations;
    const int KH = horiz_len + (iterations - 1) * (horiz_len - 1), AH = (horiz_len / 2) * iterations;
    static int overlap = -1;  // IPA_OVERLAP=0: everything on the caller's stream
    if (overlap < 0) {
        const char* e = getenv("IPA_OVERLAP");
        overlap = e ? atoi(e) : 1;
    }
    hipStream_t so = overlap && !g_prof_on ? side_stream(0) : nullptr;  // stream of the opens
    hipEvent_t ev_gray = nullptr, ev_open = nullptr;
    if (so) {
        if (hipEventCreateWithFlags(&ev_gray, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_open, hipEventDisableTiming) != hipSuccess) {
            if (ev_gray) (void)hipEventDestroy(ev_gray);
            ev_gray = nullptr;
            so = nullptr;
        }
    }
    if (so) {
        IPA_HIP(hipEventRecord(ev_gray, st));
        IPA_HIP(hipStreamWaitEvent(so, ev_gray, 0));
    } else {
        so = st;
    }
    {
        ProfScope ps(PROF_OPEN_V, so);
        rc = launch_bitmorph(bits, bpw, bsw, V, width, height, batch, 0, 2, 1, KV, 0, AV, 255, nullptr, so);
    }
    if (!rc) {
        ProfScope ps(PROF_OPEN_H, so);
        rc = launch_bitmorph(bits, bpw, bsw, H, width, height, batch, 0, 2, KH, 1, AH, 0, 255, nullptr, so);
    }
    // join: whatever happens below, the caller's stream waits for the opens
    struct Join {
        hipStream_t st, so;
        hipEvent_t a, b;
        ~Join()
        {
            if (so != st) {
                if (hipEventRecord(b, so) == hipSuccess) (void)hipStreamWaitEvent(st, b, 0);
                (void)hipEventDestroy(a);
                (void)hipEventDestroy(b);
            }
        }
    } join = {st, so, ev_gray, ev_open};
    if (rc) return rc;
    Plane csrc = G;
    if (blur5x5) {
        Plane BL = mkplane(blur_plane, L.blur_pitch, L.blur_pitch * (size_t)height);
        ProfScope ps(PROF_BLUR, st);
        rc = launch_gauss5(G, BL, width, height, batch, blur_ws, L.blur_ws_bytes, st);
        if (rc) return rc;
        csrc = BL;
    }
    double lo = canny_lo, hi = canny_hi;
    if (lo > hi) { double t = lo; lo = hi; hi = t; }
    // with a Hough stage behind it, Canny also emits the list of edge pixels
    PointSink sink = {nullptr, 0, 0, nullptr};
    static int fused = -1;  // IPA_FUSED_POINTS=1: the Canny kernels emit the edge-pixel list themselves (default: a
                            // separate compaction pass over the edge plane, which measures faster end to end)
    if (fused < 0) {
        const char* e = getenv("IPA_FUSED_POINTS");
        fused = e ? atoi(e) : 0;
    }
    if (accum && fused) {
        sink = hough_point_sink(width, height, max_points, point_counts, hough_ws);
        IPA_HIP(hipMemsetAsync(point_counts, 0, sizeof(int32_t) * (size_t)batch, st));
    }
    rc = run_canny(csrc, E, width, height, batch, (int)floor(lo), (int)floor(hi), canny_ws, L.canny_bytes, sink, st);
    if (rc) return rc;
    if (accum)
        rc = run_hough_accum(E, width, height, batch, 1.0, M_PI / 180, accum, accum_stride, seg_max, max_points, point_counts,
                             hough_ws, L.hough_bytes, fused != 0, st);
    return rc;
}

IPA_EXPORT int ipa_front_end_u8(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride, uint8_t* gray,
                                uint8_t* thresh, uint8_t* vlines, uint8_t* hlines, uint8_t* edges, size_t plane_pitch,
                                size_t plane_stride, int32_t* accum, size_t accum_stride, int32_t* seg_max,
                                int max_points, int32_t* point_counts, int width, int height, int batch, int thresh_val,
                                int vert_len, int horiz_len, int iterations, double canny_lo, double canny_hi,
                                int blur5x5, void* ws, size_t ws_bytes, void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    if (width == 0 || height == 0 || batch == 0) return IPA_OK;
    IPA_REQUIRE(ws && ws_bytes >= ipa_front_end_workspace_bytes(width, height, batch, max_points), IPA_ERR_WORKSPACE,
                "front_end: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int k = pipeline_chunks(batch);
    hipStream_t s2 = k > 1 ? side_stream(1) : nullptr;
    if (!s2)
        return front_end_chunk(bgr, bgr_pitch, bgr_stride, gray, thresh, vlines, hlines, edges, plane_pitch, plane_stride,
                               accum, accum_stride, seg_max, max_points, point_counts, width, height, batch, thresh_val,
                               vert_len, horiz_len, iterations, canny_lo, canny_hi, blur5x5, ws, ws_bytes, st);
    hipEvent_t ev[IPA_MAX_CHUNKS + 1] = {};
    int rc = IPA_OK;
    for (int c = 0; c <= k && !rc; c++)
        if (hipEventCreateWithFlags(&ev[c], hipEventDisableTiming) != hipSuccess) {
            ipa_set_error("front_end: hipEventCreate failed");
            rc = IPA_ERR_HIP;
        }
    int na = 0, nr = 0;
    if (accum) ipa_hough_dims(width, height, 1.0, M_PI / 180, &na, &nr);
    const size_t seg_stride = accum ? (size_t)na * hough_segments(nr) : 0;
    // the helper stream starts behind the work already queued on the caller's stream
    if (!rc && (hipEventRecord(ev[k], st) != hipSuccess || hipStreamWaitEvent(s2, ev[k], 0) != hipSuccess)) rc = IPA_ERR_HIP;
    size_t ws_off = 0;
    bool used_s2 = false;
    for (int c = 0; c < k && !rc; c++) {
        int lo, hi;
        chunk_range(batch, k, c, &lo, &hi);
        hipStream_t sc = (c & 1) ? s2 : st;
        // chunk c starts when the gray kernel of chunk c-1 (on the other stream) is done
        if (c > 0 && hipStreamWaitEvent(sc, ev[c - 1], 0) != hipSuccess) {
            rc = IPA_ERR_HIP;
            break;
        }
        const size_t wsc = align_up(chunk_ws_bytes(width, height, hi - lo, max_points), 256);
        used_s2 |= (c & 1) != 0;
        rc = front_end_chunk(bgr + (size_t)lo * bgr_stride, bgr_pitch, bgr_stride, gray + (size_t)lo * plane_stride,
                             thresh + (size_t)lo * plane_stride, vlines + (size_t)lo * plane_stride,
                             hlines + (size_t)lo * plane_stride, edges + (size_t)lo * plane_stride, plane_pitch, plane_stride,
                             accum ? accum + (size_t)lo * accum_stride : nullptr, accum_stride,
                             seg_max ? seg_max + (size_t)lo * seg_stride : nullptr, max_points,
                             point_counts ? point_counts + lo : nullptr, width, height, hi - lo, thresh_val, vert_len,
                             horiz_len, iterations, canny_lo, canny_hi, blur5x5, (u8*)ws + ws_off, wsc, sc, ev[c]);
        ws_off += wsc;
    }
    // join, also after an error: the caller's stream must not run ahead of the helper stream
    if (used_s2 && ev[k] && (hipEventRecord(ev[k], s2) != hipSuccess || hipStreamWaitEvent(st, ev[k], 0) != hipSuccess)) {
        (void)hipStreamSynchronize(s2);
        if (!rc) {
            ipa_set_error("front_end: joining the helper stream failed");
            rc = IPA_ERR_HIP;
        }
    }
    for (int c = 0; c <= k; c++)
        if (ev[c]) (void)hipEventDestroy(ev[c]);
    return rc;
}
