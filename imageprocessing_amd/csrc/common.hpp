// Shared host/device helpers for the gfx950 raster front-end kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ipamd.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define IPA_EXPORT extern "C" __attribute__((visibility("default")))

// One 8-bit plane (or interleaved 3-channel image) of a batch in device memory.
struct Plane {
    u8* p;
    size_t pitch;   // bytes between rows
    size_t stride;  // bytes between images
    __host__ __device__ u8* row(int b, int y) const { return p + (size_t)b * stride + (size_t)y * pitch; }
};

static inline Plane mkplane(const void* p, size_t pitch, size_t stride)
{
    Plane r;
    r.p = (u8*)p;
    r.pitch = pitch;
    r.stride = stride;
    return r;
}

void ipa_set_error(const char* fmt, ...);

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }
static inline bool plane_ok(const Plane& pl, size_t row_bytes, int batch)
{
    if (!pl.p || !aligned16(pl.p)) return false;
    if (pl.pitch % 16 || pl.pitch < row_bytes) return false;
    if (batch > 1 && (pl.stride % 16)) return false;
    return true;
}

#define IPA_REQUIRE(cond, code, ...)          \
    do {                                      \
        if (!(cond)) {                        \
            ipa_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)

#define IPA_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ipa_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return IPA_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define IPA_LAUNCH_CHECK()                                                              \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            ipa_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return IPA_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// Extra dynamic LDS (KiB, from the environment) requested at a launch: caps how many workgroups of
// that kernel fit on a CU, which leaves room for a kernel of another stream to run beside it.
static inline size_t lds_pad_kib(const char* env)
{
    const char* e = getenv(env);
    const int v = e ? atoi(e) : 0;
    return v > 0 ? (size_t)v * 1024 : 0;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Grid for a flat streaming kernel: enough blocks to fill 256 CUs several times
// over, grid-striding beyond that.
static inline unsigned stream_grid(long long items, int block)
{
    long long g = (items + block - 1) / block;
    const long long cap = 256LL * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// Decomposition of the streaming kernels: blockDim = (64, 4); thread (tx, ty) of block
// (bx, by, bz) owns 16-pixel chunk column cx = 64 bx + tx of rows
// [(4 by + ty) * ROWS_PER_LANE, ... + ROWS_PER_LANE) of image bz.  Only 32-bit index math.
constexpr int ROWS_PER_LANE = 4;
static inline dim3 rows_block() { return dim3(64, 4, 1); }
static inline dim3 rows_grid(int cols, int h, int nb) { return dim3((unsigned)cdiv(cols, 64), (unsigned)cdiv(h, 4 * ROWS_PER_LANE), (unsigned)nb); }

// Workgroups are dealt to the 8 XCDs (each with its own L2) round-robin by their linear id.
// This bijection relabels them so that every XCD works on one contiguous range of ids:
// workgroups that read the same data (consecutive ids) then share one L2.  Speed only.
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned orig, unsigned nwg)
{
    const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// Bump allocator over the caller's workspace.
struct Arena {
    u8* base;
    size_t size;
    size_t off;
    Arena(void* p, size_t n) : base((u8*)p), size(n), off(0) {}
    void* take(size_t n)
    {
        off = align_up(off, 256);
        void* r = base ? base + off : nullptr;
        off += n;
        return r;
    }
    bool ok() const { return off <= size; }
};

// ---- optional per-stage timing (HIP events on the caller's stream) --------
enum ProfStage {
    PROF_GRAY_THRESH = 0,
    PROF_OPEN_V,
    PROF_OPEN_H,
    PROF_BLUR,
    PROF_CANNY_NMS,
    PROF_GRAY_CANNY,
    PROF_HYSTERESIS,
    PROF_HOUGH_COMPACT,
    PROF_HOUGH_VOTE,
    PROF_HOUGH_PEAKS,
    PROF_NSTAGES
};
void prof_begin(int stage, hipStream_t st);
void prof_end(int stage, hipStream_t st);
struct ProfScope {
    int stage;
    hipStream_t st;
    ProfScope(int s, hipStream_t t) : stage(s), st(t) { prof_begin(s, t); }
    ~ProfScope() { prof_end(stage, st); }
};

// ---- internal entry points shared between translation units --------------
// (all enqueue on `st`; no validation: callers have validated)
// Small scratch areas (sizes in dwords, 16-byte aligned starts) that the gray kernel zeroes on the side for the
// stages behind it, so that a short pass does not spend launches on clearing a few kilobytes.
struct ClearList {
    u32* p[3];
    u32 n[3];
};
int launch_bgr2gray(Plane bgr, Plane gray, Plane bin, u32* bits, size_t bits_pitch_words, size_t bits_stride_words,
                    int w, int h, int batch, int ithresh, int maxval, int inv, hipStream_t st,
                    ClearList clr = ClearList{{nullptr, nullptr, nullptr}, {0, 0, 0}});

// bit-plane geometry: words per row rounded to a multiple of 32 (one 128-byte cache line), so
// that the 128-byte pieces the producing waves write never share a line between workgroups
// Inclusive scan of one value per thread over a 1024-thread workgroup (16 waves): shuffles inside
// the wave, the 16 wave totals through LDS.  Two barriers; s_wave must hold 16 words.
__device__ __forceinline__ u32 block_scan_1024(u32 v, u32* s_wave, u32& total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 scan = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = (u32)__shfl_up((int)scan, o);
        if (lane >= o) scan += t;
    }
    __syncthreads();  // s_wave may still be read by the previous round
    if (lane == 63) s_wave[wv] = scan;
    __syncthreads();
    u32 before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 t = s_wave[k];
        before += k < wv ? t : 0u;
        all += t;
    }
    total = all;
    return scan + before;
}

static inline size_t bits_pitch_words(int w) { return align_up((size_t)cdiv(w, 32), 32); }

int launch_pack_bits(Plane src, u32* bits, size_t bpw, size_t bsw, int w, int h, int batch, hipStream_t st);
int launch_bitmorph(const u32* bits, size_t bpw, size_t bsw, Plane dst, int w, int h, int batch, int first_is_dilate,
                    int nops, int kx, int ky, int ax, int ay, int value, const int32_t* value_dev, hipStream_t st,
                    u32* bits_out = nullptr);  // bits_out: the result also as a bit plane of the input's pitch
int launch_morph_generic(Plane src, Plane dst, Plane tmp, int w, int h, int batch, int is_dilate, int kx, int ky,
                         int ax, int ay, hipStream_t st, const int32_t* gate = nullptr);  // gate[b] == 0: image b is skipped
int launch_gauss5(Plane src, Plane dst, int w, int h, int batch, hipStream_t st);

// Per-image lists of edge pixels (y << 16 | x), appended to by the Canny kernels when the
// Hough transform follows, so that the edge plane is not read a second time.
struct PointSink {
    u32* pts;         // image b at pts + b * stride; nullptr = no list wanted
    size_t stride;    // elements
    u32 cap;          // entries kept per image (the count keeps counting beyond it)
    int32_t* counts;  // per image, zeroed by the caller
};

size_t canny_ws_bytes(int w, int h, int batch);
// run-time options (api.hip) read by the Canny launch
int option_canny_bands();
int option_canny_queue();
int option_canny_wide();
int option_hyst_rounds();

// flags_cleared: the area canny_flag_area() names has been zeroed by the caller (on the same stream)
// nms_done: the codes (0 / WEAK / 255) and the tile flags are already in place (launch_canny_wide with a BGR source)
int run_canny(Plane src, Plane dst, int w, int h, int batch, int low, int high, void* ws, size_t ws_bytes,
              PointSink sink, hipStream_t st, bool flags_cleared = false, bool nms_done = false);
// The Canny strip kernel that writes whole aligned lines, alone (gray source) or fused with BGR -> gray /
// threshold / bit plane (bgr_in).  np: 16-bit pixel pairs per lane, from canny_wide_np() (0 = not to be used).
int canny_wide_np(int w, int h, int batch);
bool canny_wide_fits(int w, int h, size_t src_pitch, size_t dst_pitch, size_t gray_pitch, size_t bin_pitch, int np);
int launch_canny_wide(bool bgr_in, int np, Plane src, Plane gray, Plane bin, u32* bits, size_t bpw, size_t bsw, Plane dst, int w,
                      int h, int batch, int ithresh, int maxval, int inv, int low, int high, void* canny_ws, ClearList clr,
                      hipStream_t st);
void canny_flag_area(int w, int h, int batch, void* ws, u32** p, u32* dwords);

size_t hough_ws_bytes(int w, int h, int batch, int max_points);
PointSink hough_point_sink(int w, int h, int max_points, int32_t* point_counts, void* ws);
int hough_segments(int numrho);
// phase: which part of the accumulator stage to enqueue.  resident_wgs > 0: the vote kernel runs as that
// many workgroups (one per CU, each looping over its share of the work) so that it keeps half of every
// CU while a streaming kernel of another stream uses the rest.
// HOUGH_VOTE_ONLY_FOREIGN_LISTS: the lists were not made by this file's compaction (no per-row counts exist)
enum HoughPhase { HOUGH_ALL = 0, HOUGH_VOTE_ONLY = 1, HOUGH_COMPACT_ONLY = 2, HOUGH_VOTE_ONLY_FOREIGN_LISTS = 3 };
void hough_rowcount_area(int w, int h, int batch, int max_points, void* ws, u32** p, u32* dwords);
// counts_cleared: point_counts has been zeroed by the caller (on the same stream)
// job: a BGR -> gray conversion the vote kernel's spare waves may work on beside the vote (*job_done tells whether
// they did; otherwise the caller launches it itself)
struct GrayJob;
int run_hough_accum(Plane src, int w, int h, int batch, double rho, double theta, int32_t* accum, size_t accum_stride,
                    int32_t* seg_max, int max_points, int32_t* point_counts, void* ws, size_t ws_bytes, int phase,
                    int resident_wgs, hipStream_t st, bool counts_cleared = false, const GrayJob* job = nullptr,
                    bool* job_done = nullptr);
// the gray job of a front-end chunk (false: its shape needs the stand-alone kernel)
bool gray_job_make(GrayJob* J, Plane bgr, Plane gray, Plane bin, u32* bits, size_t bpw, size_t bsw, int w, int h, int batch,
                   int ithresh, int maxval, int inv);
