// GaussianBlur (8-bit fixed point, as OpenCV's 8u path) and adaptiveThreshold
// (float32 Gaussian mean or integer box mean, then a signed compare).
//
// Separable: a row pass into a 16-bit (Q8.8) / 32-bit intermediate held in the
// caller's workspace, then a column pass that finishes the pixel.  Each lane
// produces 4 adjacent pixels.  Built with -ffp-contract=off: the float path
// must round every multiply and add on its own.
#include "common.hpp"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

#define IPA_MAX_TAPS 127

struct Taps {
    int n;
    int border;
    int16_t q8[IPA_MAX_TAPS + 1];  // Q0.8 taps, sum 256
};

struct TapsF {
    int n;
    float f[IPA_MAX_TAPS + 1];
};

// cv::borderInterpolate for REPLICATE(1) / REFLECT(2) / REFLECT_101(4); CONSTANT(0) -> -1
__device__ __forceinline__ int border_idx(int p, int len, int type)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (type == 1) return p < 0 ? 0 : len - 1;
    if (type == 2 || type == 4) {
        int delta = type == 4;
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    return -1;
}

__global__ __launch_bounds__(256) void k_gauss_row(Plane src, u16* __restrict__ tmp, size_t tpitch, size_t tstride,
                                                   int w, int h, int quads, long long total, Taps T)
{
    const long long step = (long long)gridDim.x * blockDim.x;
    const int r = T.n >> 1;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += step) {
        int qx = (int)(c % quads);
        long long t = c / quads;
        int y = (int)(t % h), b = (int)(t / h);
        const u8* s = src.row(b, y);
        u32 acc[4] = {0, 0, 0, 0};
        const int x0 = qx * 4;
        for (int i = 0; i < T.n + 3; i++) {
            int xi = border_idx(x0 + i - r, w, T.border);
            u32 p = xi < 0 ? 0u : s[xi];
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (i >= q && i < T.n + q) acc[q] += p * (u32)T.q8[i - q];
        }
        u16* o = tmp + (size_t)b * tstride + (size_t)y * tpitch + x0;
        u32 lo = min(acc[0], 65535u) | (min(acc[1], 65535u) << 16);
        u32 hi = min(acc[2], 65535u) | (min(acc[3], 65535u) << 16);
        *(uint2*)o = make_uint2(lo, hi);
    }
}

__global__ __launch_bounds__(256) void k_gauss_col(const u16* __restrict__ tmp, size_t tpitch, size_t tstride,
                                                   Plane dst, int w, int h, int quads, long long total, Taps T)
{
    const long long step = (long long)gridDim.x * blockDim.x;
    const int r = T.n >> 1;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += step) {
        int qx = (int)(c % quads);
        long long t = c / quads;
        int y = (int)(t % h), b = (int)(t / h);
        u32 acc[4] = {0, 0, 0, 0};
        for (int i = 0; i < T.n; i++) {
            int yi = border_idx(y + i - r, h, T.border);
            if (yi < 0) continue;
            uint2 v = *(const uint2*)(tmp + (size_t)b * tstride + (size_t)yi * tpitch + qx * 4);
            u32 k = (u32)T.q8[i];
            acc[0] += k * (v.x & 0xffffu);
            acc[1] += k * (v.x >> 16);
            acc[2] += k * (v.y & 0xffffu);
            acc[3] += k * (v.y >> 16);
        }
        u32 o = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) o |= min((acc[q] + 32768u) >> 16, 255u) << (8 * q);
        *(u32*)(dst.row(b, y) + qx * 4) = o;
    }
}

// ---- streaming 3- and 5-tap Gaussian ------------------------------------------------
// One wave walks a 256-pixel-wide strip of the image downwards, lane = 4 adjacent pixels, all
// state in registers (no LDS, no intermediate plane).  Horizontal pass in packed 16-bit
// (Q8.8, at most 255 * 256), the last five horizontal rows are kept and combined with 32-bit
// multiply-adds: (sum + 2^15) >> 16, exactly the two fixed-point passes of the generic
// kernels.  The outermost lanes only feed their neighbours, so a strip yields 248 columns.
// Taps are symmetric: (k0, k1, k2, k1, k0), k0 = 0 for three taps.
constexpr int G5_ROWS = 60;  // a multiple of 5, the period of the rolling row registers
constexpr int G5_COLS = 248;

typedef unsigned short us2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2v g5_pair(u32 v) { return __builtin_bit_cast(us2v, v); }
__device__ __forceinline__ u32 g5_u32(us2v v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ u32 g5_from_below(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ u32 g5_from_above(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xf, 0xf, true); }
// acc + (low / high 16 bits of h) * k
__device__ __forceinline__ u32 g5_mad_lo(u32 h, u32 k, u32 acc)
{
    u32 r;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(h), "s"(k), "v"(acc));
    return r;
}
__device__ __forceinline__ u32 g5_mad_hi(u32 h, u32 k, u32 acc)
{
    u32 r;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(h), "s"(k), "v"(acc));
    return r;
}

struct G5Row {
    u32 h01, h23;  // horizontal sums of the lane's 4 pixels (16-bit pairs)
};

__global__ __launch_bounds__(256) void k_gauss5_stream(Plane src, Plane dst, int w, int h, int strips_x, int bands,
                                                       int nwaves, int border, int kx0, int kx1, int kx2, int ky0,
                                                       int ky1, int ky2)
{
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 wid = blockIdx.x * 4u + (u32)wv;
    if (wid >= (u32)nwaves) return;
    const int lane = threadIdx.x & 63;
    const u32 t = wid / (u32)strips_x;
    const int sx = (int)(wid - t * (u32)strips_x);
    const int b = (int)(t / (u32)bands), band = (int)(t - (u32)b * (u32)bands);
    const int x = sx * G5_COLS - 4 + 4 * lane;
    const int y0 = band * G5_ROWS, y1 = min(y0 + G5_ROWS, h);
    const bool writer = lane >= 1 && lane <= 62 && x < w;
    // each lane reads one 32-bit word inside the row; byte selectors realise the border rule
    // while they widen the bytes to 16-bit pairs (columns further than 2 outside never matter)
    int r4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r4[j] = border_idx(min(max(x + j, -2), w + 1), w, border);
    const int xc = min(min(min(r4[0], r4[1]), min(r4[2], r4[3])), w - 4);
    const u32 sel01 = (0x0cu << 24) | ((u32)min(r4[1] - xc, 3) << 16) | (0x0cu << 8) | (u32)min(r4[0] - xc, 3);
    const u32 sel23 = (0x0cu << 24) | ((u32)min(r4[3] - xc, 3) << 16) | (0x0cu << 8) | (u32)min(r4[2] - xc, 3);
    const u8* const simg = src.p + (size_t)b * src.stride;
    u8* const dimg = dst.p + (size_t)b * dst.stride;
    const u32 spitch = (u32)src.pitch, dpitch = (u32)dst.pitch;
    auto load_row = [&](int yy) -> u32 {  // yy may lie up to 2 rows outside: same border rule
        const int ry = border_idx(min(max(yy, -2), h + 1), h, border);
        return *(const u32*)(simg + (size_t)((u32)ry * spitch + (u32)xc));
    };
    const u32 kx0p = (u32)kx0 * 0x00010001u, kx1p = (u32)kx1 * 0x00010001u, kx2p = (u32)kx2 * 0x00010001u;
    auto hrow = [&](u32 g) -> G5Row {
        const u32 c0 = __builtin_amdgcn_perm(0, g, sel01), c1 = __builtin_amdgcn_perm(0, g, sel23);  // columns 0,1 / 2,3
        const u32 L = g5_from_below(c1), R = g5_from_above(c0);                                       // columns -2,-1 / 4,5
        const u32 m1 = __builtin_amdgcn_alignbit(c0, L, 16), p12 = __builtin_amdgcn_alignbit(c1, c0, 16),
                  p34 = __builtin_amdgcn_alignbit(R, c1, 16);
        G5Row o;
        o.h01 = g5_u32((g5_pair(L) + g5_pair(c1)) * g5_pair(kx0p) + (g5_pair(m1) + g5_pair(p12)) * g5_pair(kx1p) +
                       g5_pair(c0) * g5_pair(kx2p));
        o.h23 = g5_u32((g5_pair(c0) + g5_pair(R)) * g5_pair(kx0p) + (g5_pair(p12) + g5_pair(p34)) * g5_pair(kx1p) +
                       g5_pair(c1) * g5_pair(kx2p));
        return o;
    };
    // horizontal rows y-2 .. y+1 are carried, row y+2 is produced in the trip that outputs row y
    G5Row ra = hrow(load_row(y0 - 2)), rb = hrow(load_row(y0 - 1)), rc = hrow(load_row(y0)), rd = hrow(load_row(y0 + 1));
    u32 gq[5];  // five gray rows in flight ahead of the one being consumed
#pragma unroll
    for (int k = 0; k < 5; k++) gq[k] = load_row(y0 + 2 + k);
    const u32 k0 = (u32)ky0, k1 = (u32)ky1, k2 = (u32)ky2;
#pragma unroll 5
    for (int it = 0; it < G5_ROWS; it++) {  // fixed trip count: rows past y1 are computed but not stored
        const int y = y0 + it;
        const G5Row re = hrow(gq[it % 5]);
        gq[it % 5] = load_row(y + 7);
        u32 o[4];
        {
            u32 a;
            a = g5_mad_lo(ra.h01, k0, 32768u); a = g5_mad_lo(re.h01, k0, a); a = g5_mad_lo(rb.h01, k1, a);
            a = g5_mad_lo(rd.h01, k1, a); o[0] = g5_mad_lo(rc.h01, k2, a) >> 16;
            a = g5_mad_hi(ra.h01, k0, 32768u); a = g5_mad_hi(re.h01, k0, a); a = g5_mad_hi(rb.h01, k1, a);
            a = g5_mad_hi(rd.h01, k1, a); o[1] = g5_mad_hi(rc.h01, k2, a) >> 16;
            a = g5_mad_lo(ra.h23, k0, 32768u); a = g5_mad_lo(re.h23, k0, a); a = g5_mad_lo(rb.h23, k1, a);
            a = g5_mad_lo(rd.h23, k1, a); o[2] = g5_mad_lo(rc.h23, k2, a) >> 16;
            a = g5_mad_hi(ra.h23, k0, 32768u); a = g5_mad_hi(re.h23, k0, a); a = g5_mad_hi(rb.h23, k1, a);
            a = g5_mad_hi(rd.h23, k1, a); o[3] = g5_mad_hi(rc.h23, k2, a) >> 16;
        }
        if (writer && y < y1) *(u32*)(dimg + (size_t)((u32)y * dpitch + (u32)x)) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
        ra = rb; rb = rc; rc = rd; rd = re;
    }
}


// ---- tiled Gaussian for 7 to 31 taps -------------------------------------------------------------
// One workgroup = 256 x 32 output pixels.  The source tile with its halo (border rule applied while
// loading) sits in LDS as bytes; the horizontal pass takes four taps per v_dot4_u32_u8 and leaves the
// Q8.8 sums of all tile rows + vertical halo in LDS with two consecutive rows of a column in one dword;
// the vertical pass takes two rows per v_dot2_u32_u16 (one packing of the taps for even, one for odd
// output rows) and finishes the pixels.  Integer sums, so the result is that of the two fixed-point
// passes k_gauss_row / k_gauss_col whatever the grouping; no intermediate plane in HBM, no per-tap
// border logic.  Taps must fit a byte (the host checks; a lone tap of 256 takes the other kernels).
constexpr int GT_W = 256, GT_H = 32, GT_MAX = 31;
constexpr int GT_G4 = (GT_MAX + 3) / 4;      // groups of four horizontal taps
constexpr int GT_G2 = (GT_MAX + 1) / 2 + 1;  // pairs of vertical taps (odd rows start half a pair early)
typedef unsigned short gt_us2 __attribute__((ext_vector_type(2)));

struct GaussTileTaps {
    int nx, ny, border;
    u32 x4[GT_G4];       // taps 4j .. 4j+3 as bytes
    u32 y2e[GT_G2];      // taps (2j, 2j+1) as 16-bit halves
    u32 y2o[GT_G2];      // taps (2j-1, 2j)
};

// NG4 / NPM: compile-time copies of the number of four-tap groups along a row and of the tap pairs down a column for
// the common square sizes (0 = run-time loops): the loops then have no exits, so all LDS reads of an item are issued
// before the first is used.
template <int NG4, int NPM>
__global__ __launch_bounds__(256) void k_gauss_tile(Plane src, Plane dst, int w, int h, int tiles_x, int tiles_y, GaussTileTaps T)
{
    extern __shared__ __align__(16) u8 gt_smem[];
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    const int tx = blk % tiles_x;
    blk /= tiles_x;
    const int ty = blk % tiles_y, b = blk / tiles_y;
    const int rx = T.nx >> 1, ry = T.ny >> 1;
    const int SR = GT_H + 2 * ry;                 // source rows in LDS (even)
    const int SPW = (GT_W + 2 * rx + 3) / 4 + 2;  // source row pitch in dwords
    u32* const s32 = (u32*)gt_smem;
    u32* const hz = s32 + (size_t)SR * SPW;       // [SR / 2][GT_W]: rows 2q (low half) and 2q+1 (high half)
    const int x0 = tx * GT_W, y0 = ty * GT_H;
    const int border = T.border;
    const int ng4 = (T.nx + 3) >> 2;
    // ---- source tile: LDS column c <-> image column x0 - rx + c, LDS row r <-> image row y0 - ry + r.
    // Wave v takes rows v, v + 4, ..; all of a thread's loads are issued before the first is used.
    const bool interior = x0 - rx >= 0 && x0 - rx + 4 * SPW <= w && y0 - ry >= 0 && y0 + GT_H + ry <= h;  // (uniform)
    if (interior) {
        // no border in sight: one running pointer per lane
        constexpr int RMAX = (GT_H + GT_MAX - 1 + 3) / 4;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
        const u8* p = src.row(b, y0 - ry + wv) + (x0 - rx) + 4 * ln;
        const size_t step = 4 * src.pitch;
        const bool second = ln + 64 < SPW;
        u32 v[RMAX][2];
#pragma unroll
        for (int k = 0; k < RMAX; k++) {
            if (wv + 4 * k < SR) {
                v[k][0] = *(const u32*)p;
                if (second) v[k][1] = *(const u32*)(p + 256);
                p += step;
            }
        }
        u32* o = s32 + wv * SPW + ln;
#pragma unroll
        for (int k = 0; k < RMAX; k++) {
            if (wv + 4 * k < SR) {
                o[0] = v[k][0];
                if (second) o[64] = v[k][1];
                o += 4 * SPW;
            }
        }
    } else {
        constexpr int RMAX = (GT_H + GT_MAX - 1 + 3) / 4;  // rows per wave for the largest kernel
        const int wv = tid >> 6, ln = tid & 63;
        u32 v[RMAX][2];
#pragma unroll
        for (int k = 0; k < RMAX; k++) {
            const int r = wv + 4 * k;
            v[k][0] = v[k][1] = 0;
            if (r < SR) {  // (wave-uniform)
                const int yy = border_idx(y0 - ry + r, h, border);
                if (yy >= 0) {
                    const u8* row = src.row(b, yy);
#pragma unroll
                    for (int m = 0; m < 2; m++) {
                        const int c4 = ln + 64 * m, gx = x0 - rx + 4 * c4;
                        if (c4 < SPW && gx >= 0 && gx + 4 <= w) v[k][m] = *(const u32*)(row + gx);  // (any byte alignment)
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RMAX; k++) {
            const int r = wv + 4 * k;
            if (r < SR) {
                const int yy = border_idx(y0 - ry + r, h, border);
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    const int c4 = ln + 64 * m, gx = x0 - rx + 4 * c4;
                    if (c4 >= SPW) continue;
                    u32 val = v[k][m];
                    if (yy >= 0 && !(gx >= 0 && gx + 4 <= w)) {  // columns across the image border: byte by byte
                        const u8* row = src.row(b, yy);
                        for (int j = 0; j < 4; j++) {
                            const int xx = border_idx(gx + j, w, border);
                            if (xx >= 0) val |= (u32)row[xx] << (8 * j);
                        }
                    }
                    s32[r * SPW + c4] = val;
                }
            }
        }
    }
    __syncthreads();
    // ---- horizontal pass: item = (pair of rows, 4 adjacent columns)
    for (int e = tid; e < (SR / 2) * (GT_W / 4); e += 256) {
        const int q = e >> 6, c4 = e & 63;
        u32 acc[2][4];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u32* rowp = s32 + (2 * q + half) * SPW + c4;  // byte 0 = first tap of output column x0 + 4 c4
            u32 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            u32 cur = rowp[0];
#pragma unroll
            for (int j = 0; j < (NG4 ? NG4 : GT_G4); j++) {
                if (!NG4 && j >= ng4) break;  // (wave-uniform)
                const u32 nxt = rowp[j + 1];
                const u32 k = T.x4[j];
                a0 = __builtin_amdgcn_udot4(cur, k, a0, false);
                a1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(nxt, cur, 1), k, a1, false);
                a2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(nxt, cur, 2), k, a2, false);
                a3 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(nxt, cur, 3), k, a3, false);
                cur = nxt;
            }
            acc[half][0] = a0; acc[half][1] = a1; acc[half][2] = a2; acc[half][3] = a3;
        }
        *(uint4*)(hz + (size_t)q * GT_W + 4 * c4) = make_uint4(acc[0][0] | (acc[1][0] << 16), acc[0][1] | (acc[1][1] << 16),
                                                              acc[0][2] | (acc[1][2] << 16), acc[0][3] | (acc[1][3] << 16));
    }
    __syncthreads();
    // ---- vertical pass: item = (output row, 4 adjacent columns)
    for (int e = tid; e < GT_H * (GT_W / 4); e += 256) {
        const int r = e >> 6, c4 = e & 63;
        const int y = y0 + r, x = x0 + 4 * c4;
        if (y >= h || x >= w) continue;
        const bool odd = r & 1;
        const int np = odd ? (T.ny + 2) >> 1 : (T.ny + 1) >> 1;  // pairs that hold taps
        const u32* col = hz + (size_t)(r >> 1) * GT_W + 4 * c4;
        u32 a0 = 32768u, a1 = 32768u, a2 = 32768u, a3 = 32768u;
#pragma unroll
        for (int j = 0; j < (NPM ? NPM : GT_G2); j++) {
            if (!NPM && j >= np) break;  // (NPM: taps beyond the window are zero, the rows read for them exist)
            const uint4 v = *(const uint4*)(col + (size_t)j * GT_W);
            const gt_us2 k = __builtin_bit_cast(gt_us2, odd ? T.y2o[j] : T.y2e[j]);
            a0 = __builtin_amdgcn_udot2(__builtin_bit_cast(gt_us2, v.x), k, a0, false);
            a1 = __builtin_amdgcn_udot2(__builtin_bit_cast(gt_us2, v.y), k, a1, false);
            a2 = __builtin_amdgcn_udot2(__builtin_bit_cast(gt_us2, v.z), k, a2, false);
            a3 = __builtin_amdgcn_udot2(__builtin_bit_cast(gt_us2, v.w), k, a3, false);
        }
        *(u32*)(dst.row(b, y) + x) = (a0 >> 16) | ((a1 >> 16) << 8) | ((a2 >> 16) << 16) | ((a3 >> 16) << 24);
    }
}

// ---- median 3x3 / 5x5 -----------------------------------------------------------------------------
// cv2.medianBlur(gray, 3 | 5).  The source tile with its halo (rows and columns outside the image replaced by the
// nearest one inside) sits in LDS as bytes.  5 x 5 (k_median<5>): one workgroup = 128 x 32 output pixels, a lane takes
// two adjacent pixels of a row: the 25 window values of both travel through the selection network side by side in
// the halves of a dword (v_pk_min_u16 / v_pk_max_u16).  3 x 3: k_median3 below.
#include "median_nets.inc"
constexpr int MED_TW = 128, MED_TR = 32;
template <int K>
struct MedNet;
template <>
struct MedNet<5> {
    static constexpr int LEN = MED25_LEN, OUT = MED25_OUT;
    static constexpr int net[MED25_LEN][2] = MED25_NET;
};

__device__ __forceinline__ u32 med_min(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(gt_us2, a), __builtin_bit_cast(gt_us2, b)));
}
__device__ __forceinline__ u32 med_max(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(gt_us2, a), __builtin_bit_cast(gt_us2, b)));
}

// 3 x 3: a lane takes 4 adjacent columns of 2 adjacent rows.  The two rows travel in the halves of a dword (window
// rows y-1..y+1 in the low half, y..y+2 in the high half), every column of three is sorted once and serves the three
// windows it belongs to: median = med3(max of the columns' smallest, med3 of their middle ones, min of their largest).
// About 15 packed operations per pixel instead of 30, aligned LDS reads, one dword store per lane and row.
constexpr int M3_TW = 256, M3_TR = 32;
__device__ __forceinline__ u32 med3(u32 a, u32 b, u32 c) { return med_max(med_min(a, b), med_min(med_max(a, b), c)); }

__global__ __launch_bounds__(256) void k_median3(Plane src, Plane dst, int w, int h, int tiles_x, int tiles_y)
{
    constexpr int ROWS = M3_TR + 2, PW = M3_TW / 4 + 2;  // LDS column c <-> image column x0 - 1 + c; 66 dwords per row
    __shared__ u32 tile[ROWS * PW];
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    const int tx = blk % tiles_x;
    blk /= tiles_x;
    const int ty = blk % tiles_y, b = blk / tiles_y;
    const int x0 = tx * M3_TW, y0 = ty * M3_TR;
    if (x0 - 1 >= 0 && x0 - 1 + 4 * PW <= w) {  // (uniform) no column outside the image: every load is issued before the first is used
        constexpr int PER = (ROWS * PW + 255) / 256;
        u32 v[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int e = min(tid + 256 * q, ROWS * PW - 1);
            const int rr = e / PW, d = e - rr * PW;
            v[q] = *(const u32*)(src.row(b, min(max(y0 - 1 + rr, 0), h - 1)) + x0 - 1 + 4 * d);  // (any byte alignment)
        }
#pragma unroll
        for (int q = 0; q < PER; q++)
            if (tid + 256 * q < ROWS * PW) tile[tid + 256 * q] = v[q];
    } else {
        for (int e = tid; e < ROWS * PW; e += 256) {
            const int rr = e / PW, d = e - rr * PW;
            const u8* row = src.row(b, min(max(y0 - 1 + rr, 0), h - 1));
            const int gx = x0 - 1 + 4 * d;
            u32 v;
            if (gx >= 0 && gx + 4 <= w) {
                v = *(const u32*)(row + gx);  // (any byte alignment)
            } else {
                v = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) v |= (u32)row[min(max(gx + j, 0), w - 1)] << (8 * j);
            }
            tile[e] = v;
        }
    }
    __syncthreads();
    const int wv = tid >> 6, lane = tid & 63;
    const int x = x0 + 4 * lane;
    for (int rp = wv; rp < M3_TR / 2; rp += 4) {
        const int y = y0 + 2 * rp;
        if (y >= h) break;  // (uniform over the wave)
        // rows y-1 .. y+2 of the tile, columns x-1 .. x+6 (two dwords each)
        u32 ra[4], rb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const u32* tr = tile + (2 * rp + i) * PW + lane;
            ra[i] = tr[0];
            rb[i] = tr[1];
        }
        u32 lo[6], mid[6], hi[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            // byte j of row i in the low half, of row i + 1 in the high half
            u32 e[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const u32 r0 = j < 4 ? ra[i] : rb[i], r1 = j < 4 ? ra[i + 1] : rb[i + 1];
                e[i] = __builtin_amdgcn_perm(r1, r0, 0x0c000c00u | (u32)(j & 3) | ((u32)(4 + (j & 3)) << 16));
            }
            const u32 a = med_min(e[0], e[1]), bq = med_max(e[0], e[1]);
            lo[j] = med_min(a, e[2]);
            const u32 t = med_max(a, e[2]);
            mid[j] = med_min(bq, t);
            hi[j] = med_max(bq, t);
        }
        u32 m[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            m[k] = med3(med_max(med_max(lo[k], lo[k + 1]), lo[k + 2]), med3(mid[k], mid[k + 1], mid[k + 2]),
                        med_min(med_min(hi[k], hi[k + 1]), hi[k + 2]));
        if (x < w) {  // (bytes past the width stay inside the pitch)
            const u32 o0 = __builtin_amdgcn_perm(m[1], m[0], 0x0c0c0400u) | __builtin_amdgcn_perm(m[3], m[2], 0x04000c0cu);
            const u32 o1 = __builtin_amdgcn_perm(m[1], m[0], 0x0c0c0602u) | __builtin_amdgcn_perm(m[3], m[2], 0x06020c0cu);
            *(u32*)(dst.row(b, y) + x) = o0;
            if (y + 1 < h) *(u32*)(dst.row(b, y + 1) + x) = o1;
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_median(Plane src, Plane dst, int w, int h, int tiles_x, int tiles_y)
{
    constexpr int R = K / 2, ROWS = MED_TR + K - 1, PW = (MED_TW + K - 1 + 3) / 4 + 1;  // row pitch in dwords
    __shared__ u32 tile[ROWS * PW];
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    const int tx = blk % tiles_x;
    blk /= tiles_x;
    const int ty = blk % tiles_y, b = blk / tiles_y;
    const int x0 = tx * MED_TW, y0 = ty * MED_TR;
    // LDS column c <-> image column x0 - R + c, LDS row rr <-> image row y0 - R + rr (both clamped to the image)
    for (int e = tid; e < ROWS * PW; e += 256) {
        const int rr = e / PW, d = e - rr * PW;
        const u8* row = src.row(b, min(max(y0 - R + rr, 0), h - 1));
        const int gx = x0 - R + 4 * d;
        u32 v;
        if (gx >= 0 && gx + 4 <= w) {
            v = *(const u32*)(row + gx);  // (any byte alignment)
        } else {
            v = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) v |= (u32)row[min(max(gx + j, 0), w - 1)] << (8 * j);
        }
        tile[e] = v;
    }
    __syncthreads();
    const int wv = tid >> 6, lane = tid & 63;
    const int x = x0 + 2 * lane;
    const u32 off = (u32)(lane & 1) * 2u;  // byte of the lane's first column inside its first dword
    for (int r = wv; r < MED_TR; r += 4) {
        const int y = y0 + r;
        if (y >= h) break;  // (uniform over the wave)
        u32 p[K * K];
#pragma unroll
        for (int dy = 0; dy < K; dy++) {
            const u32* tr = tile + (r + dy) * PW + (lane >> 1);
            const u32 d0 = tr[0], d1 = tr[1], d2 = tr[2];
            const u32 q0 = __builtin_amdgcn_alignbyte(d1, d0, off), q1 = __builtin_amdgcn_alignbyte(d2, d1, off);
#pragma unroll
            for (int c = 0; c < K; c++)  // low half: column c of the first pixel's window, high half: of the second's
                p[dy * K + c] = __builtin_amdgcn_perm(q1, q0, 0x0c000c00u | (u32)c | ((u32)(c + 1) << 16));
        }
#pragma unroll
        for (int c = 0; c < MedNet<K>::LEN; c++) {
            constexpr auto& NET = MedNet<K>::net;
            const int ia = NET[c][0], ib = NET[c][1];
            const u32 lo = med_min(p[ia], p[ib]), hi = med_max(p[ia], p[ib]);
            p[ia] = lo;
            p[ib] = hi;
        }
        const u32 m = p[MedNet<K>::OUT];
        u8* o = dst.row(b, y) + x;
        if (x + 1 < w) *(unsigned short*)o = (unsigned short)((m & 0xffu) | ((m >> 16) << 8));
        else if (x < w) *o = (u8)m;
    }
}

// ---- adaptive threshold ----------------------------------------------------
// row pass: float Gaussian (taps accumulated left to right) or integer box sums
template <bool GAUSS>
__global__ __launch_bounds__(256) void k_adapt_row(Plane src, u32* __restrict__ tmp, size_t tpitch, size_t tstride,
                                                   int w, int h, int quads, long long total, TapsF T)
{
    const long long step = (long long)gridDim.x * blockDim.x;
    const int r = T.n >> 1;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += step) {
        int qx = (int)(c % quads);
        long long t = c / quads;
        int y = (int)(t % h), b = (int)(t / h);
        const u8* s = src.row(b, y);
        const int x0 = qx * 4;
        float fa[4] = {0.f, 0.f, 0.f, 0.f};
        u32 ia[4] = {0, 0, 0, 0};
        for (int i = 0; i < T.n + 3; i++) {
            int xi = border_idx(x0 + i - r, w, 1);
            u32 p = s[xi];
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (i >= q && i < T.n + q) {
                    if (GAUSS) {
                        float prod = __fmul_rn((float)p, T.f[i - q]);
                        fa[q] = __fadd_rn(fa[q], prod);
                    } else {
                        ia[q] += p;
                    }
                }
        }
        u32* o = tmp + (size_t)b * tstride + (size_t)y * tpitch + x0;
        if (GAUSS)
            *(uint4*)o = make_uint4(__float_as_uint(fa[0]), __float_as_uint(fa[1]), __float_as_uint(fa[2]),
                                    __float_as_uint(fa[3]));
        else
            *(uint4*)o = make_uint4(ia[0], ia[1], ia[2], ia[3]);
    }
}

// column pass + compare.  tab semantics: BINARY: src - mean > -idelta ; INV: src - mean <= -idelta
template <bool GAUSS>
__global__ __launch_bounds__(256) void k_adapt_col(Plane src, const u32* __restrict__ tmp, size_t tpitch,
                                                   size_t tstride, Plane dst, int w, int h, int quads, long long total,
                                                   TapsF T, double box_scale, int idelta, int inv, u32 maxv)
{
    const long long step = (long long)gridDim.x * blockDim.x;
    const int r = T.n >> 1;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += step) {
        int qx = (int)(c % quads);
        long long t = c / quads;
        int y = (int)(t % h), b = (int)(t / h);
        const u32* base = tmp + (size_t)b * tstride + qx * 4;
        int mean[4];
        if (GAUSS) {
            uint4 c0 = *(const uint4*)(base + (size_t)y * tpitch);
            float acc[4] = {__fmul_rn(T.f[r], __uint_as_float(c0.x)), __fmul_rn(T.f[r], __uint_as_float(c0.y)),
                            __fmul_rn(T.f[r], __uint_as_float(c0.z)), __fmul_rn(T.f[r], __uint_as_float(c0.w))};
            for (int i = 1; i <= r; i++) {
                uint4 a = *(const uint4*)(base + (size_t)border_idx(y + i, h, 1) * tpitch);
                uint4 d = *(const uint4*)(base + (size_t)border_idx(y - i, h, 1) * tpitch);
                float k = T.f[r + i];
                acc[0] = __fadd_rn(acc[0], __fmul_rn(k, __fadd_rn(__uint_as_float(a.x), __uint_as_float(d.x))));
                acc[1] = __fadd_rn(acc[1], __fmul_rn(k, __fadd_rn(__uint_as_float(a.y), __uint_as_float(d.y))));
                acc[2] = __fadd_rn(acc[2], __fmul_rn(k, __fadd_rn(__uint_as_float(a.z), __uint_as_float(d.z))));
                acc[3] = __fadd_rn(acc[3], __fmul_rn(k, __fadd_rn(__uint_as_float(a.w), __uint_as_float(d.w))));
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int v = __float2int_rn(acc[q]);
                mean[q] = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
        } else {
            u32 acc[4] = {0, 0, 0, 0};
            for (int i = -r; i <= r; i++) {
                uint4 a = *(const uint4*)(base + (size_t)border_idx(y + i, h, 1) * tpitch);
                acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int v = (int)rint((double)acc[q] * box_scale);
                mean[q] = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
        }
        u32 sv = *(const u32*)(src.row(b, y) + qx * 4);
        u32 o = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int d = (int)((sv >> (8 * q)) & 0xffu) - mean[q];
            bool on = inv ? (d <= -idelta) : (d > -idelta);
            o |= (on ? maxv : 0u) << (8 * q);
        }
        *(u32*)(dst.row(b, y) + qx * 4) = o;
    }
}

// ---- streaming adaptive threshold ---------------------------------------------------------
// A workgroup of four waves walks a 256-pixel-wide strip downwards, lane = 4 adjacent pixels, wave j takes the
// rows y0 + j, y0 + 4 + j, ...  A lane reads its own row window [x - R, x + 3 + R] (clamped: BORDER_REPLICATE)
// straight from memory, so no data moves between lanes; the row-pass results of the last 2R + 8 rows live in
// a ring in LDS shared by the four waves (one barrier per trip of four rows, no intermediate plane).  Sharing
// the ring keeps sixteen waves on a CU for the 31-tap window.  The float path keeps the operation order of the
// two-pass kernels: taps accumulated left to right in the row pass, symmetric pairs in the column pass, every
// multiply and add rounded on its own.
typedef float f2v __attribute__((ext_vector_type(2)));

template <int R, bool GAUSS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(R > 5 ? 4 : 6, R > 5 ? 4 : 8))) void k_adapt_stream(Plane src, Plane dst, int w, int h, int strips_x, int bands,
                                                      int band_rows, TapsF T, double box_scale, int idelta, int inv,
                                                      u32 maxv)
{
    constexpr int N = 2 * R + 1;           // taps
    constexpr int PAD = (R + 3) / 4 * 4;   // window start x - PAD, a multiple of 4 pixels
    constexpr int ND = (PAD + 4 + R + 3) / 4;  // dwords covering [x - PAD, x + 3 + R]
    // ring rows: the rows y - R .. y + 3 + R of the four output rows of a trip, and the four rows that are
    // written for the next trip while slower waves still read
    constexpr int SLOTS = N + 7;
    extern __shared__ __align__(16) u32 ring[];  // [SLOTS rows][64 lanes][4]
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const u32 t = blockIdx.x / (u32)strips_x;
    const int sx = (int)(blockIdx.x - t * (u32)strips_x);
    const int b = (int)(t / (u32)bands), band = (int)(t - (u32)b * (u32)bands);
    const int x = sx * 256 + 4 * lane;
    const int y0 = band * band_rows, y1 = min(y0 + band_rows, h);
    uint4* const myring = (uint4*)ring + lane;  // slot s at myring[s * 64]; row r lives in slot (r - (y0 - R)) mod SLOTS
    const bool live = x < w;

    // strips whose windows lie inside the row read plain dwords (wave-uniform choice)
    const bool interior = sx * 256 - PAD >= 0 && sx * 256 + 252 - PAD + 4 * ND <= w;
    // Border strips: dword k of the window starts at pixel xk = x - PAD + 4k (a multiple of four).  It is read at
    // the nearest dword inside the row (the dword that holds pixel w - 1 lies inside the pitch), and the bytes
    // left of pixel 0 / right of pixel w - 1 are replaced by copies of that pixel.
    auto load_row = [&](int yy, u32 (&d)[ND]) {
        const u8* row = src.row(b, yy);
        if (interior) {
#pragma unroll
            for (int k = 0; k < ND; k++) d[k] = *(const u32*)(row + (x - PAD + 4 * k));
        } else {
            const u32 first = row[0] * 0x01010101u, lastp = row[w - 1] * 0x01010101u;
            int xx = x;
            asm volatile("" : "+v"(xx));  // the offsets and masks below are recomputed per row, not kept in registers
#pragma unroll
            for (int k = 0; k < ND; k++) {
                const int xk = xx - PAD + 4 * k;
                const u32 v = *(const u32*)(row + min(max(xk, 0), (w - 1) & ~3));
                // bytes xk + j with 0 <= xk + j <= w - 1 come from memory
                const int over = xk + 3 - (w - 1);  // > 0: that many bytes lie right of the last pixel
                u32 keep = xk < 0 ? 0u : (over <= 0 ? 0xffffffffu : (over >= 4 ? 0u : (0xffffffffu >> (8 * over))));
                d[k] = (v & keep) | ((xk < 0 ? first : lastp) & ~keep);
            }
        }
    };
    // row pass of one image row (window dwords `d`)
    auto row_pass = [&](const u32 (&d)[ND]) -> uint4 {
        uint4 out;
        if (GAUSS) {
            float p[N + 3];
#pragma unroll
            for (int j = 0; j < N + 3; j++) {
                const int o = PAD - R + j;  // byte offset inside d
                p[j] = (float)((d[o >> 2] >> (8 * (o & 3))) & 0xffu);
            }
            f2v a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < N; i++) {
                const float k = T.f[i];
                const f2v p01 = {p[i], p[i + 1]}, p23 = {p[i + 2], p[i + 3]};
                a01 = a01 + p01 * k;  // -ffp-contract=off: separate multiply and add
                a23 = a23 + p23 * k;
                if (N > 16 && (i & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // (keeps the byte conversions near their use)
            }
            out = make_uint4(__float_as_uint(a01.x), __float_as_uint(a01.y), __float_as_uint(a23.x), __float_as_uint(a23.y));
        } else {
            u32 sacc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < N + 3; j++) {
                const int o = PAD - R + j;
                const u32 pv = (d[o >> 2] >> (8 * (o & 3))) & 0xffu;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (j >= q && j < N + q) sacc[q] += pv;
            }
            out = make_uint4(sacc[0], sacc[1], sacc[2], sacc[3]);
        }
        return out;
    };
    auto clampy = [&](int r) { return min(max(r, 0), h - 1); };

    // rows y0 - R .. y0 + R + 3 (ring slots 0 .. 2R + 3), every fourth one by this wave; rows outside the image
    // are copies of the nearest image row
    {
        u32 dn[ND];
        load_row(clampy(y0 - R + wv), dn);
        for (int k = wv; k < 2 * R + 4; k += 4) {
            u32 dc[ND];
#pragma unroll
            for (int q = 0; q < ND; q++) dc[q] = dn[q];
            if (k + 4 < 2 * R + 4) load_row(clampy(y0 - R + k + 4), dn);
            myring[k * 64] = row_pass(dc);
        }
    }
    const int trips = (y1 - y0 + 3) >> 2;
    // the window of the row pushed in a trip and the source pixels compared in it are loaded one trip ahead
    u32 dn[ND];
    u32 svn = 0;
    load_row(clampy(y0 + wv + 4 + R), dn);
    if (live && y0 + wv < y1) svn = *(const u32*)(src.row(b, y0 + wv) + x);  // the dword lies inside the pitch
    int sb = wv;  // slot of row y - R
    for (int trip = 0; trip < trips; trip++) {
        const int y = y0 + 4 * trip + wv;
        u32 dc[ND];
#pragma unroll
        for (int q = 0; q < ND; q++) dc[q] = dn[q];
        const u32 sv = svn;
        const bool push = y + 4 + R <= y1 - 1 + R + 3 && trip + 1 < trips;  // some wave reads that row in the next trip
        if (trip + 1 < trips) {
            load_row(clampy(y + 8 + R), dn);
            if (live && y + 4 < y1) svn = *(const u32*)(src.row(b, y + 4) + x);
        }
        __syncthreads();  // rows up to y0 + 4 * trip + 3 + R are in the ring
        auto slot_rel = [&](int j) -> uint4* {  // row y - R + j, 0 <= j < SLOTS
            int s2 = sb + j;
            if (s2 >= SLOTS) s2 -= SLOTS;
            return myring + s2 * 64;
        };
        // row y + 4 + R for the next trip: its slot held row y - R - 4, which nobody reads any more
        if (push) *slot_rel(N + 3) = row_pass(dc);
        if (y < y1) {
            int mean[4];
            if (GAUSS) {
                const uint4 c = *slot_rel(R);
                const float kc = T.f[R];
                f2v a01 = {__uint_as_float(c.x) * kc, __uint_as_float(c.y) * kc};
                f2v a23 = {__uint_as_float(c.z) * kc, __uint_as_float(c.w) * kc};
#pragma unroll
                for (int i = 1; i <= R; i++) {
                    const uint4 lo = *slot_rel(R - i), hi = *slot_rel(R + i);
                    const float k = T.f[R + i];
                    const f2v s01 = {__uint_as_float(hi.x) + __uint_as_float(lo.x), __uint_as_float(hi.y) + __uint_as_float(lo.y)};
                    const f2v s23 = {__uint_as_float(hi.z) + __uint_as_float(lo.z), __uint_as_float(hi.w) + __uint_as_float(lo.w)};
                    a01 = a01 + s01 * k;
                    a23 = a23 + s23 * k;
                    if ((i & 3) == 0) __builtin_amdgcn_sched_barrier(0);  // at most eight ring rows in registers
                }
                const float av[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int v = __float2int_rn(av[q]);
                    mean[q] = v < 0 ? 0 : (v > 255 ? 255 : v);
                }
            } else {
                u32 acc[4] = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < N; i++) {
                    const uint4 a = *slot_rel(i);
                    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
                    if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int v = (int)rint((double)acc[q] * box_scale);
                    mean[q] = v < 0 ? 0 : (v > 255 ? 255 : v);
                }
            }
            if (live) {
                u32 o = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int dlt = (int)((sv >> (8 * q)) & 0xffu) - mean[q];
                    const bool on = inv ? (dlt <= -idelta) : (dlt > -idelta);
                    o |= (on ? maxv : 0u) << (8 * q);
                }
                *(u32*)(dst.row(b, y) + x) = o;  // bytes past the width stay inside the pitch
            }
        }
        sb += 4;
        if (sb >= SLOTS) sb -= SLOTS;
    }
}

// ---- host-side tap tables (cv::getGaussianKernel and its Q0.8 form) ---------
void gaussian_taps_f64(int n, double sigma, double* k)
{
    static const double tab1[] = {1.0};
    static const double tab3[] = {0.25, 0.5, 0.25};
    static const double tab5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
    static const double tab7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    if ((n & 1) && n <= 7 && sigma <= 0) {
        const double* f = n == 1 ? tab1 : n == 3 ? tab3 : n == 5 ? tab5 : tab7;
        for (int i = 0; i < n; i++) k[i] = f[i];
        return;
    }
    double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2 = -0.5 / (sx * sx), sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        k[i] = exp(scale2 * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] *= sum;
}

// Q0.8 taps summing to 256: exact tables for sigma <= 0 and small n, otherwise
// quantised from the outside in, carrying the rounding error, centre = remainder.
void gaussian_taps_q8(int n, double sigma, int16_t* q)
{
    if (sigma <= 0 && n == 9) {
        static const int16_t t9[] = {4, 13, 30, 51, 60, 51, 30, 13, 4};
        for (int i = 0; i < 9; i++) q[i] = t9[i];
        return;
    }
    double kd[IPA_MAX_TAPS + 1];
    gaussian_taps_f64(n, sigma, kd);
    if (sigma <= 0 && (n & 1) && n <= 7) {
        for (int i = 0; i < n; i++) q[i] = (int16_t)lrint(kd[i] * 256.0);
        return;
    }
    int n2 = n / 2;
    double err = 0;
    long long sum = 0;
    for (int i = 0; i < n2; i++) {
        double adj = kd[i] * 256.0 + err;
        long long v = llrint(adj);
        err = adj - (double)v;
        q[i] = q[n - 1 - i] = (int16_t)v;
        sum += v;
    }
    q[n2] = (int16_t)(256 - 2 * sum);
}

size_t tmp_pitch_elems(int w) { return align_up((size_t)w, 64); }

}  // namespace

IPA_EXPORT size_t ipa_gaussian_blur_workspace_bytes(int width, int height, int batch)
{
    if (width <= 0 || height <= 0 || batch <= 0) return 0;
    return align_up(tmp_pitch_elems(width) * sizeof(u16) * (size_t)height * batch, 256) + 256;
}

static int gaussian_impl(Plane s, Plane d, int width, int height, int batch, int kw, int kh, double sigma_x,
                         double sigma_y, int border, void* ws, size_t ws_bytes, hipStream_t st)
{
    if (sigma_y <= 0) sigma_y = sigma_x;
    if (kw <= 0 && sigma_x > 0) kw = (int)lrint(sigma_x * 3 * 2 + 1) | 1;
    if (kh <= 0 && sigma_y > 0) kh = (int)lrint(sigma_y * 3 * 2 + 1) | 1;
    IPA_REQUIRE(kw > 0 && (kw & 1) && kh > 0 && (kh & 1), IPA_ERR_ARG,
                "GaussianBlur: ksize.width > 0 && ksize.width %% 2 == 1 && ksize.height > 0 && ksize.height %% 2 == 1");
    IPA_REQUIRE(kw <= IPA_MAX_TAPS && kh <= IPA_MAX_TAPS, IPA_ERR_ARG, "GaussianBlur: kernel larger than %d taps",
                IPA_MAX_TAPS);
    IPA_REQUIRE(border == 0 || border == 1 || border == 2 || border == 4, IPA_ERR_ARG,
                "GaussianBlur: unsupported borderType %d", border);
    IPA_REQUIRE(ws && ws_bytes >= ipa_gaussian_blur_workspace_bytes(width, height, batch), IPA_ERR_WORKSPACE,
                "GaussianBlur: workspace too small");
    Taps tx, ty;
    tx.n = kw; ty.n = kh; tx.border = ty.border = border;
    gaussian_taps_q8(kw, sigma_x > 0 ? sigma_x : 0, tx.q8);
    gaussian_taps_q8(kh, sigma_y > 0 ? sigma_y : 0, ty.q8);
    // 3 / 5 taps per direction, a border rule that reads inside the image, and an image the
    // 32-bit offsets and word reads of the streaming kernel can address: no intermediate plane
    static int g5 = -1;  // IPA_GAUSS_STREAM=0 forces the two-pass kernels
    if (g5 < 0) {
        const char* e = getenv("IPA_GAUSS_STREAM");
        g5 = e ? atoi(e) : 1;
    }
    if (g5 && (kw == 3 || kw == 5) && (kh == 3 || kh == 5) && border != 0 && width >= 8 && height >= 2 &&
        (unsigned long long)height * s.pitch < 0xffffffffull && (unsigned long long)height * d.pitch < 0xffffffffull) {
        const int strips_x = cdiv(width, G5_COLS), bands = cdiv(height, G5_ROWS);
        const long long nwaves = (long long)strips_x * bands * batch;
        IPA_REQUIRE(nwaves < 0x7fffffffLL, IPA_ERR_ARG, "GaussianBlur: too many strips");
        // outer, middle and centre tap of each direction (no outer tap for 3 taps)
        const int x0 = kw == 3 ? 0 : tx.q8[0], x1 = tx.q8[kw == 3 ? 0 : 1], x2 = tx.q8[kw == 3 ? 1 : 2];
        const int v0 = kh == 3 ? 0 : ty.q8[0], v1 = ty.q8[kh == 3 ? 0 : 1], v2 = ty.q8[kh == 3 ? 1 : 2];
        hipLaunchKernelGGL(k_gauss5_stream, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, s, d, width, height, strips_x,
                           bands, (int)nwaves, border, x0, x1, x2, v0, v1, v2);
        IPA_LAUNCH_CHECK();
        return IPA_OK;
    }
    bool byte_taps = kw >= 3 && kh >= 3 && kw <= GT_MAX && kh <= GT_MAX;
    for (int i = 0; i < kw && byte_taps; i++) byte_taps = tx.q8[i] >= 0 && tx.q8[i] <= 255;
    for (int i = 0; i < kh && byte_taps; i++) byte_taps = ty.q8[i] >= 0 && ty.q8[i] <= 255;
    if (g5 && byte_taps) {
        // tiled two-pass kernel: both passes in LDS, no intermediate plane
        const int tiles_x = cdiv(width, GT_W), tiles_y = cdiv(height, GT_H);
        const long long blocks = (long long)tiles_x * tiles_y * batch;
        IPA_REQUIRE(blocks < 0x7fffffffLL, IPA_ERR_ARG, "GaussianBlur: too many tiles");
        GaussTileTaps T;
        memset(&T, 0, sizeof(T));
        T.nx = kw; T.ny = kh; T.border = border;
        for (int i = 0; i < kw; i++) T.x4[i >> 2] |= (u32)tx.q8[i] << (8 * (i & 3));
        for (int i = 0; i < kh; i++) {
            T.y2e[i >> 1] |= (u32)ty.q8[i] << (16 * (i & 1));
            T.y2o[(i + 1) >> 1] |= (u32)ty.q8[i] << (16 * ((i + 1) & 1));
        }
        const int SR = GT_H + 2 * (kh >> 1), SPW = (GT_W + 2 * (kw >> 1) + 3) / 4 + 2;
        const size_t lds = (size_t)SR * SPW * 4 + (size_t)(SR / 2) * GT_W * 4;
        static bool attr_set = false;
        if (lds > 48 * 1024 && !attr_set) {
            IPA_HIP(hipFuncSetAttribute((const void*)k_gauss_tile<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
            attr_set = true;
        }
#define IPA_GT(G4, PM) hipLaunchKernelGGL((k_gauss_tile<G4, PM>), dim3((unsigned)blocks), dim3(256), lds, st, s, d, width, height, tiles_x, tiles_y, T)
        // the common square windows (their LDS need stays below 48 KiB) with unrolled tap loops
        if (kw == 7 && kh == 7) IPA_GT(2, 4);
        else if (kw == 9 && kh == 9) IPA_GT(3, 5);
        else if (kw == 11 && kh == 11) IPA_GT(3, 6);
        else if (kw == 13 && kh == 13) IPA_GT(4, 7);
        else if (kw == 15 && kh == 15) IPA_GT(4, 8);
        else IPA_GT(0, 0);
#undef IPA_GT
        IPA_LAUNCH_CHECK();
        return IPA_OK;
    }
    Arena ar(ws, ws_bytes);
    size_t tp = tmp_pitch_elems(width);
    u16* tmp = (u16*)ar.take(tp * sizeof(u16) * (size_t)height * batch);
    int quads = cdiv(width, 4);
    long long total = (long long)quads * height * batch;
    unsigned grid = stream_grid(total, 256);
    hipLaunchKernelGGL(k_gauss_row, dim3(grid), dim3(256), 0, st, s, tmp, tp, tp * (size_t)height, width, height,
                       quads, total, tx);
    IPA_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gauss_col, dim3(grid), dim3(256), 0, st, (const u16*)tmp, tp, tp * (size_t)height, d, width,
                       height, quads, total, ty);
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}

IPA_EXPORT int ipa_gaussian_blur_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst,
                                    size_t dst_pitch, size_t dst_stride, int width, int height, int batch, int kw,
                                    int kh, double sigma_x, double sigma_y, int border, void* ws, size_t ws_bytes,
                                    void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    if (width == 0 || height == 0 || batch == 0) return IPA_OK;
    Plane s = mkplane(src, src_pitch, src_stride), d = mkplane(dst, dst_pitch, dst_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch) && plane_ok(d, (size_t)width, batch), IPA_ERR_ALIGN,
                "GaussianBlur: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    return gaussian_impl(s, d, width, height, batch, kw, kh, sigma_x, sigma_y, border, ws, ws_bytes,
                         (hipStream_t)stream);
}

// 5x5, sigma 0, REFLECT_101 blur used by the fused front-end (variant with blur).
int launch_gauss5(Plane src, Plane dst, int w, int h, int batch, void* ws, size_t ws_bytes, hipStream_t st);
int launch_gauss5(Plane src, Plane dst, int w, int h, int batch, void* ws, size_t ws_bytes, hipStream_t st)
{
    return gaussian_impl(src, dst, w, h, batch, 5, 5, 0, 0, IPA_BORDER_REFLECT_101, ws, ws_bytes, st);
}

IPA_EXPORT size_t ipa_adaptive_threshold_workspace_bytes(int width, int height, int batch)
{
    if (width <= 0 || height <= 0 || batch <= 0) return 0;
    return align_up(tmp_pitch_elems(width) * sizeof(u32) * (size_t)height * batch, 256) + 256;
}

IPA_EXPORT int ipa_adaptive_threshold_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst,
                                         size_t dst_pitch, size_t dst_stride, int width, int height, int batch,
                                         double max_value, int method, int type, int block_size, double c, void* ws,
                                         size_t ws_bytes, void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    IPA_REQUIRE(block_size % 2 == 1 && block_size > 1, IPA_ERR_ARG,
                "adaptiveThreshold: blockSize %% 2 == 1 && blockSize > 1");
    IPA_REQUIRE(method == 0 || method == 1, IPA_ERR_ARG, "adaptiveThreshold: unknown adaptive method");
    IPA_REQUIRE(type == 0 || type == 1, IPA_ERR_ARG, "adaptiveThreshold: unknown threshold type");
    IPA_REQUIRE(block_size <= IPA_MAX_TAPS, IPA_ERR_ARG, "adaptiveThreshold: blockSize larger than %d", IPA_MAX_TAPS);
    if (width == 0 || height == 0 || batch == 0) return IPA_OK;
    hipStream_t st = (hipStream_t)stream;
    Plane s = mkplane(src, src_pitch, src_stride), d = mkplane(dst, dst_pitch, dst_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch) && plane_ok(d, (size_t)width, batch), IPA_ERR_ALIGN,
                "adaptiveThreshold: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    if (max_value < 0) {
        IPA_HIP(hipMemset2DAsync(dst, dst_pitch, 0, (size_t)width, (size_t)height, st));
        for (int b = 1; b < batch; b++)
            IPA_HIP(hipMemset2DAsync(dst + (size_t)b * dst_stride, dst_pitch, 0, (size_t)width, (size_t)height, st));
        return IPA_OK;
    }
    IPA_REQUIRE(ws && ws_bytes >= ipa_adaptive_threshold_workspace_bytes(width, height, batch), IPA_ERR_WORKSPACE,
                "adaptiveThreshold: workspace too small");
    TapsF T;
    T.n = block_size;
    double kd[IPA_MAX_TAPS + 1];
    gaussian_taps_f64(block_size, 0.0, kd);
    for (int i = 0; i < block_size; i++) T.f[i] = (float)kd[i];
    long mvl = lrint(max_value);
    u32 maxv = (u32)(mvl < 0 ? 0 : (mvl > 255 ? 255 : mvl));
    int idelta = type == 0 ? (int)ceil(c) : (int)floor(c);
    double box_scale = 1.0 / ((double)block_size * block_size);
    // block sizes of the reference's callers (11, 31) and the small ones: streaming kernel
    static int astream = -1;  // IPA_ADAPT_STREAM=0 forces the two-pass kernels
    if (astream < 0) {
        const char* e = getenv("IPA_ADAPT_STREAM");
        astream = e ? atoi(e) : 1;
    }
    int R = block_size / 2;
    // Gaussian windows between the built sizes run in the next larger one with zero taps at both ends: a product
    // with 0 is 0 and adding 0 changes no partial sum, so the float result is bit for bit the same.
    if (astream && method == 1 && R < 15 && !(R == 1 || R == 2 || R == 3 || R == 5)) {
        const int Rp = R < 5 ? 5 : 15;
        TapsF P;
        P.n = 2 * Rp + 1;
        for (int i = 0; i < P.n; i++) P.f[i] = 0.f;
        for (int i = 0; i < T.n; i++) P.f[i + (Rp - R)] = T.f[i];
        T = P;
        R = Rp;
        block_size = 2 * Rp + 1;
    }
    if (astream && (R == 1 || R == 2 || R == 3 || R == 5 || R == 15)) {
        // bands as tall as the launch allows (every band row-passes 2R + 4 rows before its first output row) while
        // the chip still gets two rounds of workgroups (four fit a CU)
        const int strips_x = cdiv(width, 256);
        const int min_rows = R <= 5 ? 64 : 128;
        const long long per_band = (long long)strips_x * batch;
        const int max_bands = cdiv(height, min_rows);
        static int adapt_rounds = -1;  // IPA_ADAPT_ROUNDS: rounds of workgroups a launch should reach
        if (adapt_rounds < 0) {
            const char* e = getenv("IPA_ADAPT_ROUNDS");
            adapt_rounds = e ? atoi(e) : 4;
        }
        const int fit = R <= 5 ? 8 : 4;  // workgroups per CU
        const int bands0 = (int)std::max<long long>(1, std::min<long long>(max_bands, ((long long)adapt_rounds * fit * 256 + per_band - 1) / per_band));
        const int band_rows = (cdiv(height, bands0) + 3) & ~3;
        const int bands = cdiv(height, band_rows);
        const long long nwgs = per_band * bands;
        IPA_REQUIRE(nwgs < 0x7fffffffLL, IPA_ERR_ARG, "adaptiveThreshold: too many strips");
        const size_t lds = (size_t)(block_size + 7) * 64 * 16;  // ring slots: see the kernel
        const dim3 grid((unsigned)nwgs), block(256);
#define IPA_ADAPT(RR)                                                                                                    \
    do {                                                                                                                 \
        if (method == 1) {                                                                                               \
            IPA_HIP(hipFuncSetAttribute((const void*)k_adapt_stream<RR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                                      \
            hipLaunchKernelGGL((k_adapt_stream<RR, true>), grid, block, lds, st, s, d, width, height, strips_x, bands,    \
                               band_rows, T, box_scale, idelta, type, maxv);                                             \
        } else {                                                                                                         \
            IPA_HIP(hipFuncSetAttribute((const void*)k_adapt_stream<RR, false>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                                      \
            hipLaunchKernelGGL((k_adapt_stream<RR, false>), grid, block, lds, st, s, d, width, height, strips_x, bands,   \
                               band_rows, T, box_scale, idelta, type, maxv);                                             \
        }                                                                                                                \
    } while (0)
        if (R == 1) IPA_ADAPT(1);
        else if (R == 2) IPA_ADAPT(2);
        else if (R == 3) IPA_ADAPT(3);
        else if (R == 5) IPA_ADAPT(5);
        else IPA_ADAPT(15);
#undef IPA_ADAPT
        IPA_LAUNCH_CHECK();
        return IPA_OK;
    }
    size_t tp = tmp_pitch_elems(width);
    u32* tmp = (u32*)ws;
    int quads = cdiv(width, 4);
    long long total = (long long)quads * height * batch;
    unsigned grid = stream_grid(total, 256);
    if (method == 1) {
        hipLaunchKernelGGL((k_adapt_row<true>), dim3(grid), dim3(256), 0, st, s, tmp, tp, tp * (size_t)height, width,
                           height, quads, total, T);
        IPA_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_adapt_col<true>), dim3(grid), dim3(256), 0, st, s, (const u32*)tmp, tp,
                           tp * (size_t)height, d, width, height, quads, total, T, box_scale, idelta, type, maxv);
    } else {
        hipLaunchKernelGGL((k_adapt_row<false>), dim3(grid), dim3(256), 0, st, s, tmp, tp, tp * (size_t)height, width,
                           height, quads, total, T);
        IPA_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_adapt_col<false>), dim3(grid), dim3(256), 0, st, s, (const u32*)tmp, tp,
                           tp * (size_t)height, d, width, height, quads, total, T, box_scale, idelta, type, maxv);
    }
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}

IPA_EXPORT int ipa_median_blur_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst, size_t dst_pitch,
                                  size_t dst_stride, int width, int height, int batch, int ksize, void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    IPA_REQUIRE(ksize == 3 || ksize == 5, IPA_ERR_ARG, "medianBlur: ksize must be 3 or 5 (larger windows are not built)");
    if (width == 0 || height == 0 || batch == 0) return IPA_OK;
    Plane s = mkplane(src, src_pitch, src_stride), d = mkplane(dst, dst_pitch, dst_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch) && plane_ok(d, (size_t)width, batch), IPA_ERR_ALIGN,
                "medianBlur: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    IPA_REQUIRE(src != dst, IPA_ERR_ARG, "medianBlur: in place is not supported");
    const int tiles_x = cdiv(width, MED_TW), tiles_y = cdiv(height, MED_TR);
    const long long blocks = (long long)tiles_x * tiles_y * batch;
    IPA_REQUIRE(blocks < 0x7fffffffLL, IPA_ERR_ARG, "medianBlur: too many tiles");
    if (ksize == 3) {
        const int tx3 = cdiv(width, M3_TW), ty3 = cdiv(height, M3_TR);
        const long long blocks3 = (long long)tx3 * ty3 * batch;
        hipLaunchKernelGGL(k_median3, dim3((unsigned)blocks3), dim3(256), 0, (hipStream_t)stream, s, d, width, height, tx3, ty3);
    } else
        hipLaunchKernelGGL(k_median<5>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, s, d, width, height, tiles_x, tiles_y);
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}
