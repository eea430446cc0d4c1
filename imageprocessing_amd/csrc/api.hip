// Library-level entry points: version, error text, pitch helper and the fused
// front-end that chains the per-op launchers on one stream.
#include "common.hpp"
#include "gray_core.hpp"

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
#include <atomic>
#include <map>
#include <mutex>
#include <vector>
namespace {
struct ProfRec {
    int stage;
    hipEvent_t a, b;
    bool ok;
};
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};
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
                                              "gaussian_blur",    "canny_nms",      "gray_thresh_canny", "canny_hysteresis",
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

// ---- run-time options ---------------------------------------------------------
// Schedule switches of the fused front-end.  Their defaults come from the environment the first
// time one is needed (IPA_PIPELINE, IPA_OVERLAP, IPA_PHASED, IPA_HEAD_START, IPA_CANNY_BANDS,
// IPA_CANNY_QUEUE); ipa_set_option changes them for the calls that follow, from any thread.
// canny_bands: -1 = rows per band of the Canny strip kernel by launch size, 16 / 64 = forced;
// canny_queue: 0 = non-maximum suppression in place in every row (A/B; results do not differ).
// canny_wide: -1 = the kernel that writes whole aligned lines (fused with the BGR conversion in the front-end
// pass) for launches large enough to fill the chip, 0 = never (default), 4 / 8 = always, with that many pixel pairs per lane.
// vote_gray: 0 (default) = in the phased schedule the gray conversion of chunk c + 1 runs as a kernel of its own on the
// helper stream beside a resident vote grid, 1 = it is done by spare waves of the vote kernel of chunk c (one launch),
// 2 = as 1, with the opens of a chunk beside that chunk's vote instead of beside its Canny kernels.
// hyst_rounds: Canny on images of more than 4096 tiles: region launches of the hysteresis queued before the kernel
// that finishes what they left on the device (default 6; 1 leaves nearly everything to that kernel: tests).
namespace {
enum { OPT_PIPELINE = 0, OPT_OVERLAP, OPT_PHASED, OPT_HEAD_START, OPT_CANNY_BANDS, OPT_CANNY_QUEUE, OPT_CANNY_WIDE, OPT_VOTE_GRAY, OPT_HYST_ROUNDS, OPT_COUNT };
const char* const g_opt_name[OPT_COUNT] = {"pipeline_chunks", "overlap", "phased", "head_start_ticks", "canny_bands", "canny_queue", "canny_wide", "vote_gray", "hyst_rounds"};
const char* const g_opt_env[OPT_COUNT] = {"IPA_PIPELINE", "IPA_OVERLAP", "IPA_PHASED", "IPA_HEAD_START", "IPA_CANNY_BANDS", "IPA_CANNY_QUEUE", "IPA_CANNY_WIDE", "IPA_VOTE_GRAY", "IPA_HYST_ROUNDS"};
const int g_opt_default[OPT_COUNT] = {-1, 1, 1, 3000, -1, 1, 0, 0, 6};
std::atomic<int> g_opt[OPT_COUNT];
std::once_flag g_opt_once;
void options_init()
{
    std::call_once(g_opt_once, [] {
        for (int i = 0; i < OPT_COUNT; i++) {
            const char* e = getenv(g_opt_env[i]);
            g_opt[i].store(e ? atoi(e) : g_opt_default[i]);
        }
    });
}
int option(int which)
{
    options_init();
    return g_opt[which].load(std::memory_order_relaxed);
}
}  // namespace

int option_canny_bands() { return option(OPT_CANNY_BANDS); }
int option_canny_queue() { return option(OPT_CANNY_QUEUE); }
int option_canny_wide() { return option(OPT_CANNY_WIDE); }
int option_hyst_rounds() { return option(OPT_HYST_ROUNDS); }

IPA_EXPORT int ipa_set_option(const char* name, int value)
{
    IPA_REQUIRE(name != nullptr, IPA_ERR_ARG, "set_option: NULL name");
    options_init();
    for (int i = 0; i < OPT_COUNT; i++)
        if (!strcmp(name, g_opt_name[i])) {
            g_opt[i].store(value);
            return IPA_OK;
        }
    ipa_set_error("set_option: unknown option '%s'", name);
    return IPA_ERR_ARG;
}

IPA_EXPORT int ipa_get_option(const char* name, int* value)
{
    IPA_REQUIRE(name != nullptr && value != nullptr, IPA_ERR_ARG, "get_option: NULL argument");
    for (int i = 0; i < OPT_COUNT; i++)
        if (!strcmp(name, g_opt_name[i])) {
            *value = option(i);
            return IPA_OK;
        }
    ipa_set_error("get_option: unknown option '%s'", name);
    return IPA_ERR_ARG;
}

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

// Large batches run as several chunks, each with its own part of the workspace, so that stages
// bound by different units of the chip can run side by side (see ipa_front_end_u8).
constexpr int IPA_MAX_CHUNKS = 8;
static int pipeline_chunks(int batch, int width, int height)
{
    // option pipeline_chunks: -1 = by size; 0 or 1 = one chain on the caller's stream; n = that many chunks
    int pipe = option(OPT_PIPELINE);
    if (pipe > IPA_MAX_CHUNKS) pipe = IPA_MAX_CHUNKS;
    // per-stage timing wants the stages one after the other, each over the whole batch
    if (g_prof_on) return 1;
    if (pipe >= 0) {  // forced: any chunk count the batch can fill
        if (pipe > batch) pipe = batch;
        return pipe < 1 ? 1 : pipe;
    }
    // Measured on 4K images (ms per batch with 1 / 2 / 3 / 4 chunks): 128 images 4.46 / 4.62 / - / 5.15,
    // 256: 8.56 / 8.30 / - / 8.94, 512: 16.4 / 15.6 / - / 15.7, 768: - / 22.8 / 22.0 / 22.5,
    // 1024: - / 29.6 / 29.4 / 29.0 -- a chunk wants about 1.5 gigapixels before overlap pays.
    int k = 1;
    const double gpx = (double)batch * width * height * 1e-9;
    if (gpx >= 7.4) k = 4;
    else if (gpx >= 5.3) k = 3;
    else if (gpx >= 1.46) k = 2;
    while (k > 1 && batch / k < 32) k--;  // chunks of at least 32 images
    return k;
}
// images [lo, hi) of chunk c out of k
static void chunk_range(int batch, int k, int c, int* lo, int* hi)
{
    const int base = batch / k, rem = batch % k;
    *lo = c * base + (c < rem ? c : rem);
    *hi = *lo + base + (c < rem ? 1 : 0);
}

IPA_EXPORT int ipa_front_end_chunks(int width, int height, int batch)
{
    if (width <= 0 || height <= 0 || batch <= 0) return 0;
    return pipeline_chunks(batch, width, height);
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
// Helper streams: the two opens only depend on the gray kernel and are bound by HBM writes, the Canny
// kernels by the vector ALUs, so they run side by side; the second helper carries the other half of a
// pipelined batch.  Every (device, caller's stream) gets a pair of its own, so that calls issued from
// several host threads on streams of their own do not wait for each other; both are joined back into
// the caller's stream before a call returns.  Beyond SIDE_MAX_PAIRS pairs per process the pairs are
// shared (correct, but such calls then queue behind each other).
constexpr size_t SIDE_MAX_PAIRS = 64;
struct SidePair {
    hipStream_t s[2] = {nullptr, nullptr};
};
std::mutex g_side_mu;
std::map<std::pair<int, hipStream_t>, SidePair> g_side;

// The device `st` belongs to must be the current one (helper streams and the library's launches go to
// the current device).  Returns the device, or -1 with the error text set.
int stream_device(hipStream_t st)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) {
        ipa_set_error("front_end: no current device");
        return -1;
    }
    if (st) {
        hipDevice_t sd = -1;
        if (hipStreamGetDevice(st, &sd) == hipSuccess && (int)sd != cur) {
            ipa_set_error("front_end: the stream belongs to device %d but device %d is current", (int)sd, cur);
            return -1;
        }
    }
    return cur;
}

// which = 0: stream of the opens, 1: stream of the second half of a pipelined batch
hipStream_t side_stream(int which, int dev, hipStream_t caller)
{
    std::lock_guard<std::mutex> lk(g_side_mu);
    auto key = std::make_pair(dev, caller);
    auto it = g_side.find(key);
    if (it == g_side.end()) {
        if (g_side.size() >= SIDE_MAX_PAIRS) {  // share the pair of another stream of this device
            for (it = g_side.begin(); it != g_side.end() && it->first.first != dev; ++it) {}
        }
        if (it == g_side.end()) it = g_side.emplace(key, SidePair()).first;
    }
    hipStream_t& sref = it->second.s[which];
    if (!sref && hipStreamCreateWithFlags(&sref, hipStreamNonBlocking) != hipSuccess) sref = nullptr;
    return sref;
}
}  // namespace

namespace {
// One chunk of a front-end call (a run of consecutive images with its own part of the workspace),
// cut into the three stages the schedules below place on streams:
//   stage_binarise  gray + threshold + bit plane, then the two opens      (bound by HBM)
//   stage_edges     [blur] Canny NMS + hysteresis, edge-pixel lists        (vector ALU / latency)
//   stage_vote      Hough accumulator                                      (LDS atomics)
struct Chunk {
    Plane B, G, T, V, H, E;
    int32_t *accum, *seg_max, *point_counts;
    size_t accum_stride;
    int max_points, width, height, batch, tv, KV, AV, KH, AH, blur5x5, lo, hi;
    int dev;             // current device = device of the caller's stream
    hipStream_t caller;  // the caller's stream (key of the helper streams)
    FrontWs L;
    u32* bits;
    u32 *vbits, *hbits;  // the two line masks as bit planes too (or NULL)
    void *canny_ws, *hough_ws, *blur_ws;
    u8* blur_plane;
    size_t bpw, bsw;
    bool precleared;  // the gray kernel zeroes the Canny tile flags and the Hough point counts (short passes)
    int wide_np;      // > 0: gray, threshold, bit plane and Canny's non-maximum suppression in one kernel (pixel pairs per lane)
};

bool fused_points();

int chunk_init(Chunk& C, const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride, uint8_t* gray, uint8_t* thresh,
               uint8_t* vlines, uint8_t* hlines, uint8_t* edges, size_t plane_pitch, size_t plane_stride, int32_t* accum,
               size_t accum_stride, int32_t* seg_max, int max_points, int32_t* point_counts, int width, int height, int batch,
               int thresh_val, int vert_len, int horiz_len, int iterations, double canny_lo, double canny_hi, int blur5x5,
               void* ws, size_t ws_bytes)
{
    C.B = mkplane(bgr, bgr_pitch, bgr_stride);
    C.G = mkplane(gray, plane_pitch, plane_stride), C.T = mkplane(thresh, plane_pitch, plane_stride);
    C.V = mkplane(vlines, plane_pitch, plane_stride), C.H = mkplane(hlines, plane_pitch, plane_stride);
    C.E = mkplane(edges, plane_pitch, plane_stride);
    IPA_REQUIRE(plane_ok(C.B, (size_t)width * 3, batch), IPA_ERR_ALIGN, "front_end: bgr pointer/pitch misaligned");
    IPA_REQUIRE(plane_ok(C.G, (size_t)width, batch) && plane_ok(C.T, (size_t)width, batch) &&
                    plane_ok(C.V, (size_t)width, batch) && plane_ok(C.H, (size_t)width, batch) &&
                    (!edges || plane_ok(C.E, (size_t)width, batch)),
                IPA_ERR_ALIGN, "front_end: output plane pointer/pitch misaligned");
    IPA_REQUIRE(edges || !accum, IPA_ERR_ARG, "front_end: the Hough stage needs the edges plane");
    IPA_REQUIRE(!accum || point_counts, IPA_ERR_ARG, "front_end: point_counts is NULL");
    IPA_REQUIRE(ws && ws_bytes >= chunk_ws_bytes(width, height, batch, max_points), IPA_ERR_WORKSPACE,
                "front_end: workspace too small");
    C.accum = accum; C.accum_stride = accum_stride; C.seg_max = seg_max; C.point_counts = point_counts;
    C.max_points = max_points; C.width = width; C.height = height; C.batch = batch; C.blur5x5 = blur5x5;
    C.L = front_layout(width, height, batch, max_points);
    Arena ar(ws, ws_bytes);
    C.bits = (u32*)ar.take(C.L.bits_bytes);
    C.canny_ws = ar.take(C.L.canny_bytes);
    C.hough_ws = ar.take(C.L.hough_bytes);
    C.blur_ws = ar.take(C.L.blur_ws_bytes);
    C.blur_plane = (u8*)ar.take(C.L.blur_plane_bytes);
    C.bpw = bits_pitch_words(width), C.bsw = C.bpw * (size_t)height;
    C.vbits = C.hbits = nullptr;  // set by the caller when wanted
    C.tv = thresh_val < -1 ? -1 : (thresh_val > 256 ? 256 : thresh_val);
    // open with (1 x vert_len) and (horiz_len x 1), iterations folded into one larger kernel
    C.KV = vert_len + (iterations - 1) * (vert_len - 1), C.AV = (vert_len / 2) * iterations;
    C.KH = horiz_len + (iterations - 1) * (horiz_len - 1), C.AH = (horiz_len / 2) * iterations;
    double lo = canny_lo, hi = canny_hi;
    if (lo > hi) { double t = lo; lo = hi; hi = t; }
    C.lo = (int)floor(lo), C.hi = (int)floor(hi);
    // short passes (a few images, or one huge one): the flags and counters (up to a megabyte) are zeroed inside the gray kernel
    static const int preclear = getenv("IPA_PRECLEAR") ? atoi(getenv("IPA_PRECLEAR")) : 1;  // 0: separate memsets (A/B)
    C.precleared = preclear && edges != nullptr && !fused_points() && canny_ws_bytes(width, height, batch) <= 1024 * 1024 && batch <= 1024;
    // one kernel from BGR to the Canny codes when nothing sits between the gray plane and Canny
    C.wide_np = 0;
    if (edges != nullptr && !blur5x5 && !fused_points()) {
        const int np = canny_wide_np(width, height, batch);
        if (np && canny_wide_fits(width, height, bgr_pitch, plane_pitch, plane_pitch, plane_pitch, np)) C.wide_np = np;
    }
    return IPA_OK;
}

// gray + THRESH_BINARY_INV plane + its bit plane, one pass over the BGR input; with C.wide_np the same kernel
// also does Canny's gradient and non-maximum suppression (the edge plane then holds the codes 0 / WEAK / 255)
int stage_gray(const Chunk& C, hipStream_t st)
{
    ClearList clr = {{nullptr, nullptr, nullptr}, {0, 0, 0}};
    if (C.precleared) {
        canny_flag_area(C.width, C.height, C.batch, C.canny_ws, &clr.p[0], &clr.n[0]);
        if (C.accum) {
            clr.p[1] = (u32*)C.point_counts;
            clr.n[1] = (u32)C.batch;
            hough_rowcount_area(C.width, C.height, C.batch, C.max_points, C.hough_ws, &clr.p[2], &clr.n[2]);
        }
    }
    if (C.wide_np) {
        // the tile flags this kernel raises itself (and the marks behind them) are zeroed before it starts
        u32* fp = nullptr;
        u32 fn = 0;
        canny_flag_area(C.width, C.height, C.batch, C.canny_ws, &fp, &fn);
        IPA_HIP(hipMemsetAsync(fp, 0, (size_t)fn * 4, st));
        clr.p[0] = nullptr;
        clr.n[0] = 0;
        ProfScope ps(PROF_GRAY_CANNY, st);
        return launch_canny_wide(true, C.wide_np, C.B, C.G, C.T, C.bits, C.bpw, C.bsw, C.E, C.width, C.height, C.batch, C.tv, 255, 1,
                                 C.lo, C.hi, C.canny_ws, clr, st);
    }
    ProfScope ps(PROF_GRAY_THRESH, st);
    return launch_bgr2gray(C.B, C.G, C.T, C.bits, C.bpw, C.bsw, C.width, C.height, C.batch, C.tv, 255, 1, st, clr);
}

int stage_opens(const Chunk& C, hipStream_t st)
{
    int rc;
    {
        ProfScope ps(PROF_OPEN_V, st);
        rc = launch_bitmorph(C.bits, C.bpw, C.bsw, C.V, C.width, C.height, C.batch, 0, 2, 1, C.KV, 0, C.AV, 255, nullptr, st, C.vbits);
    }
    if (rc) return rc;
    ProfScope ps(PROF_OPEN_H, st);
    return launch_bitmorph(C.bits, C.bpw, C.bsw, C.H, C.width, C.height, C.batch, 0, 2, C.KH, 1, C.AH, 0, 255, nullptr, st, C.hbits);
}

bool fused_points()
{
    static int fused = -1;  // IPA_FUSED_POINTS=1: the Canny kernels emit the edge-pixel list themselves (default: a
                            // separate compaction pass over the edge plane, which measures faster end to end)
    if (fused < 0) {
        const char* e = getenv("IPA_FUSED_POINTS");
        fused = e ? atoi(e) : 0;
    }
    return fused != 0;
}

// GaussianBlur 5x5 of the gray plane into the chunk's scratch plane (variant with blur only)
int stage_blur(const Chunk& C, hipStream_t st)
{
    if (!C.blur5x5 || !C.E.p) return IPA_OK;
    Plane BL = mkplane(C.blur_plane, C.L.blur_pitch, C.L.blur_pitch * (size_t)C.height);
    ProfScope ps(PROF_BLUR, st);
    return launch_gauss5(C.G, BL, C.width, C.height, C.batch, C.blur_ws, C.L.blur_ws_bytes, st);
}

// Canny (of the blurred plane if there is one); with a Hough stage behind it also the lists of edge pixels
int stage_edges(const Chunk& C, bool with_lists, hipStream_t st)
{
    int rc;
    if (!C.E.p) return IPA_OK;  // no edge map wanted
    const Plane csrc = C.blur5x5 ? mkplane(C.blur_plane, C.L.blur_pitch, C.L.blur_pitch * (size_t)C.height) : C.G;
    PointSink sink = {nullptr, 0, 0, nullptr};
    if (C.accum && fused_points()) {
        sink = hough_point_sink(C.width, C.height, C.max_points, C.point_counts, C.hough_ws);
        IPA_HIP(hipMemsetAsync(C.point_counts, 0, sizeof(int32_t) * (size_t)C.batch, st));
    }
    rc = run_canny(csrc, C.E, C.width, C.height, C.batch, C.lo, C.hi, C.canny_ws, C.L.canny_bytes, sink, st,
                   C.precleared || C.wide_np != 0, C.wide_np != 0);
    if (rc || !C.accum || !with_lists || fused_points()) return rc;
    return run_hough_accum(C.E, C.width, C.height, C.batch, 1.0, M_PI / 180, C.accum, C.accum_stride, C.seg_max, C.max_points,
                           C.point_counts, C.hough_ws, C.L.hough_bytes, HOUGH_COMPACT_ONLY, 0, st, C.precleared);
}

// lists_ready: stage_edges has already made the edge-pixel lists; job: a gray conversion that may ride along
int stage_vote(const Chunk& C, bool lists_ready, int resident_wgs, hipStream_t st, const GrayJob* job = nullptr, bool* job_done = nullptr)
{
    if (job_done) *job_done = false;
    if (!C.accum) return IPA_OK;
    return run_hough_accum(C.E, C.width, C.height, C.batch, 1.0, M_PI / 180, C.accum, C.accum_stride, C.seg_max, C.max_points,
                           C.point_counts, C.hough_ws, C.L.hough_bytes,
                           fused_points() ? HOUGH_VOTE_ONLY_FOREIGN_LISTS : (lists_ready ? HOUGH_VOTE_ONLY : HOUGH_ALL), resident_wgs,
                           st, C.precleared, job, job_done);
}

// Events that order the helper streams.  They come from a per-thread pool that lives as long as the
// thread: a wait enqueued on a stream refers to the record made before it, so an event may be
// recorded again by a later call while earlier work is still in flight.  An event belongs to the device
// that was current when it was created and can only be recorded on streams of that device, so a thread
// keeps one pool per device it has used.
struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t next = 0;
};
struct EventPools {
    std::map<int, EventPool> by_dev;
    ~EventPools()
    {
        for (auto& kv : by_dev)
            for (hipEvent_t e : kv.second.ev) (void)hipEventDestroy(e);
    }
};
thread_local EventPools g_events;
struct Events {
    int dev;  // the current device = the device of every stream of the call
    explicit Events(int device) : dev(device) {}
    hipEvent_t make()
    {
        EventPool& P = g_events.by_dev[dev];
        if (P.next == P.ev.size()) {
            if (P.ev.size() >= 64) {  // more than a call ever needs: start over
                P.next = 0;
            } else {
                hipEvent_t e = nullptr;
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    return nullptr;
                }
                P.ev.push_back(e);
            }
        }
        return P.ev[P.next++];
    }
};

// a -> b: everything queued on b from now on waits for what is queued on a so far
int stream_after(hipStream_t b, hipStream_t a, Events& E)
{
    hipEvent_t e = E.make();
    IPA_REQUIRE(e != nullptr, IPA_ERR_HIP, "front_end: hipEventCreate failed");
    if (hipEventRecord(e, a) != hipSuccess || hipStreamWaitEvent(b, e, 0) != hipSuccess) {
        // the caller may carry on without the helper stream: leave no sticky error for its next launch check
        ipa_set_error("front_end: ordering two streams failed: %s", hipGetErrorString(hipGetLastError()));
        return IPA_ERR_HIP;
    }
    return IPA_OK;
}

// Holds its stream back for about `ticks` of the 100 MHz wall clock (bounded): gives a kernel that
// was just queued on another stream time to get its workgroups onto the CUs first.
__global__ void k_head_start(unsigned ticks)
{
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < 100000 && wall_clock64() - t0 < ticks; i++) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

// One chunk, stage after stage on the caller's stream; the opens on a helper stream beside Canny.
static int front_end_chunk(const Chunk& C, hipStream_t st, hipEvent_t after_gray = nullptr)
{
    int rc = stage_gray(C, st);
    if (rc) return rc;
    if (after_gray) IPA_HIP(hipEventRecord(after_gray, st));
    // option overlap = 0: everything on the caller's stream
    hipStream_t so = option(OPT_OVERLAP) && !g_prof_on ? side_stream(0, C.dev, C.caller) : nullptr;  // stream of the opens
    Events E(C.dev);
    if (so && stream_after(so, st, E) != IPA_OK) so = nullptr;
    if (!so) so = st;
    rc = stage_opens(C, so);
    if (!rc) rc = stage_blur(C, st);
    if (!rc) rc = stage_edges(C, false, st);
    if (!rc) rc = stage_vote(C, false, 0, st);
    // join, also after an error: the caller's stream waits for the opens
    if (so != st && stream_after(st, so, E) != IPA_OK) {
        (void)hipStreamSynchronize(so);
        if (!rc) rc = IPA_ERR_HIP;
    }
    return rc;
}

static int front_end_impl(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride, uint8_t* gray,
                          uint8_t* thresh, uint8_t* vlines, uint8_t* hlines, uint8_t* edges, size_t plane_pitch,
                          size_t plane_stride, int32_t* accum, size_t accum_stride, int32_t* seg_max,
                          int max_points, int32_t* point_counts, int width, int height, int batch, int thresh_val,
                          int vert_len, int horiz_len, int iterations, double canny_lo, double canny_hi,
                          int blur5x5, void* ws, size_t ws_bytes, void* stream, uint32_t* vlines_bits, uint32_t* hlines_bits)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    if (width == 0 || height == 0 || batch == 0) return IPA_OK;
    IPA_REQUIRE(ws && ws_bytes >= ipa_front_end_workspace_bytes(width, height, batch, max_points), IPA_ERR_WORKSPACE,
                "front_end: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int dev = stream_device(st);
    if (dev < 0) return IPA_ERR_ARG;
    const int k = pipeline_chunks(batch, width, height);
    hipStream_t s2 = k > 1 ? side_stream(1, dev, st) : nullptr;
    const int nk = s2 ? k : 1;
    int na = 0, nr = 0;
    if (accum) ipa_hough_dims(width, height, 1.0, M_PI / 180, &na, &nr);
    const size_t seg_stride = accum ? (size_t)na * hough_segments(nr) : 0;
    Chunk C[IPA_MAX_CHUNKS];
    size_t ws_off = 0;
    for (int c = 0; c < nk; c++) {
        int lo = 0, hi = batch;
        if (nk > 1) chunk_range(batch, nk, c, &lo, &hi);
        const size_t wsc = nk > 1 ? align_up(chunk_ws_bytes(width, height, hi - lo, max_points), 256) : ws_bytes;
        int rc = chunk_init(C[c], bgr + (size_t)lo * bgr_stride, bgr_pitch, bgr_stride, gray + (size_t)lo * plane_stride,
                            thresh + (size_t)lo * plane_stride, vlines + (size_t)lo * plane_stride,
                            hlines + (size_t)lo * plane_stride, edges ? edges + (size_t)lo * plane_stride : nullptr, plane_pitch, plane_stride,
                            accum ? accum + (size_t)lo * accum_stride : nullptr, accum_stride,
                            seg_max ? seg_max + (size_t)lo * seg_stride : nullptr, max_points,
                            point_counts ? point_counts + lo : nullptr, width, height, hi - lo, thresh_val, vert_len,
                            horiz_len, iterations, canny_lo, canny_hi, blur5x5, (u8*)ws + ws_off, wsc);
        if (rc) return rc;
        C[c].dev = dev;
        C[c].caller = st;
        C[c].vbits = vlines_bits ? vlines_bits + (size_t)lo * C[c].bsw : nullptr;
        C[c].hbits = hlines_bits ? hlines_bits + (size_t)lo * C[c].bsw : nullptr;
        ws_off += wsc;
    }
    if (nk == 1) return front_end_chunk(C[0], st);

    Events E(dev);
    int rc = stream_after(s2, st, E);  // the helper stream starts behind the work already queued on the caller's stream
    if (rc) return rc;
    // option phased = 0: the staggered schedule also when there is a Hough stage
    if (accum && option(OPT_PHASED)) {
        // The vote of chunk c (LDS atomics) runs beside the binarisation of chunk c+1 (HBM): it is
        // launched first, as one workgroup per CU that stays until all its work is done, and the
        // streaming kernels fill the other half of every CU.  Launched the other way round the
        // streaming kernels' small workgroups would keep refilling the CUs and the vote's large ones
        // would wait.  The Canny stage (which wants the whole chip) follows when both are done.
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        cus &= ~7;
        const int head_ticks = option(OPT_HEAD_START);  // head start of the vote kernel in 10 ns ticks
        rc = stage_gray(C[0], st);
        if (!rc) rc = stream_after(s2, st, E);
        if (!rc) rc = stage_opens(C[0], s2);
        if (!rc) rc = stage_blur(C[0], st);
        int opens_due = -1;  // chunk whose opens wait for the next vote phase (option vote_gray = 2)
        for (int c = 0; c < nk && !rc; c++) {
            rc = stage_edges(C[c], true, st);
            if (rc) break;
            const bool more = c + 1 < nk;
            if (opens_due == c) {  // beside this chunk's vote instead of beside its Canny kernels
                rc = stream_after(s2, st, E);
                if (!rc) rc = stage_opens(C[c], s2);
                if (rc) break;
                opens_due = -1;
            }
            // The binarisation of chunk c + 1 (bound by HBM) beside the vote of chunk c (LDS atomics, vector instructions):
            // as spare waves inside the vote kernel, or else as a kernel of its own on the helper stream.
            bool voted = false, rode = false;
            if (more && option(OPT_VOTE_GRAY)) {
                const Chunk& N = C[c + 1];
                GrayJob job;
                if (!N.wide_np && !N.precleared &&
                    gray_job_make(&job, N.B, N.G, N.T, N.bits, N.bpw, N.bsw, N.width, N.height, N.batch, N.tv, 255, 1)) {
                    rc = stage_vote(C[c], true, 0, st, &job, &rode);
                    voted = true;
                }
            }
            if (!voted) {
                if (more) {
                    rc = stream_after(s2, st, E);
                    if (rc) break;
                    hipLaunchKernelGGL(k_head_start, dim3(1), dim3(1), 0, s2, (unsigned)head_ticks);
                }
                rc = stage_vote(C[c], true, more ? cus : 0, st);
            }
            if (rc || !more) break;
            if (rode) {
                rc = stage_blur(C[c + 1], st);
                if (option(OPT_VOTE_GRAY) == 2) {
                    opens_due = c + 1;
                    continue;
                }
                if (!rc) rc = stream_after(s2, st, E);  // the opens read the bit plane
                if (!rc) rc = stage_opens(C[c + 1], s2);
                continue;
            }
            if (voted) rc = stream_after(s2, st, E);  // (the vote kernel did not take the job: the conversion follows it)
            if (!rc) rc = stage_gray(C[c + 1], s2);
            if (!rc) rc = stage_blur(C[c + 1], s2);  // bound by HBM like the gray kernel: beside the vote as well
            if (!rc) rc = stream_after(st, s2, E);  // the next Canny needs the gray (blurred) plane, not the opens
            if (!rc) rc = stage_opens(C[c + 1], s2);
        }
    } else {
        // Chunks alternate between the two streams, each starting when the previous chunk's gray
        // kernel is done: the chunks stay one stage apart.
        hipEvent_t prev = nullptr;
        for (int c = 0; c < nk && !rc; c++) {
            hipStream_t sc = (c & 1) ? s2 : st;
            if (prev && hipStreamWaitEvent(sc, prev, 0) != hipSuccess) {
                rc = IPA_ERR_HIP;
                break;
            }
            prev = E.make();
            if (!prev) {
                rc = IPA_ERR_HIP;
                break;
            }
            rc = front_end_chunk(C[c], sc, prev);
        }
    }
    // join, also after an error: the caller's stream must not run ahead of the helper stream
    if (stream_after(st, s2, E) != IPA_OK) {
        (void)hipStreamSynchronize(s2);
        if (!rc) {
            ipa_set_error("front_end: joining the helper stream failed");
            rc = IPA_ERR_HIP;
        }
    }
    return rc;
}

IPA_EXPORT int ipa_front_end_u8(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride, uint8_t* gray,
                                uint8_t* thresh, uint8_t* vlines, uint8_t* hlines, uint8_t* edges, size_t plane_pitch,
                                size_t plane_stride, int32_t* accum, size_t accum_stride, int32_t* seg_max,
                                int max_points, int32_t* point_counts, int width, int height, int batch, int thresh_val,
                                int vert_len, int horiz_len, int iterations, double canny_lo, double canny_hi,
                                int blur5x5, void* ws, size_t ws_bytes, void* stream)
{
    return front_end_impl(bgr, bgr_pitch, bgr_stride, gray, thresh, vlines, hlines, edges, plane_pitch, plane_stride, accum,
                          accum_stride, seg_max, max_points, point_counts, width, height, batch, thresh_val, vert_len, horiz_len,
                          iterations, canny_lo, canny_hi, blur5x5, ws, ws_bytes, stream, nullptr, nullptr);
}

IPA_EXPORT size_t ipa_bits_pitch_words(int width) { return width > 0 ? bits_pitch_words(width) : 0; }

IPA_EXPORT int ipa_front_end_bits_u8(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride, uint8_t* gray,
                                     uint8_t* thresh, uint8_t* vlines, uint8_t* hlines, uint8_t* edges, size_t plane_pitch,
                                     size_t plane_stride, int32_t* accum, size_t accum_stride, int32_t* seg_max,
                                     int max_points, int32_t* point_counts, int width, int height, int batch, int thresh_val,
                                     int vert_len, int horiz_len, int iterations, double canny_lo, double canny_hi,
                                     int blur5x5, void* ws, size_t ws_bytes, void* stream, uint32_t* vlines_bits,
                                     uint32_t* hlines_bits)
{
    IPA_REQUIRE(((uintptr_t)vlines_bits & 15) == 0 && ((uintptr_t)hlines_bits & 15) == 0, IPA_ERR_ALIGN,
                "front_end: bit planes must be 16-byte aligned");
    return front_end_impl(bgr, bgr_pitch, bgr_stride, gray, thresh, vlines, hlines, edges, plane_pitch, plane_stride, accum,
                          accum_stride, seg_max, max_points, point_counts, width, height, batch, thresh_val, vert_len, horiz_len,
                          iterations, canny_lo, canny_hi, blur5x5, ws, ws_bytes, stream, vlines_bits, hlines_bits);
}
