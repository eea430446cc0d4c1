// Standard Hough line transform (OpenCV HoughLinesStandard semantics).
//
//  k_hough_compact : edge plane -> per-image list of non-zero pixels (y<<16 | x),
//                    one global atomic per workgroup pass (block-wide scan).
//  k_hough_vote    : LDS-privatised voting.  A workgroup owns a few consecutive
//                    angles (and, for huge images, a slice of the rho axis) of
//                    ONE image; only the rho bins an angle can reach for this
//                    image size live in LDS.  It streams the image's point list,
//                    votes with LDS atomics, then writes its accumulator rows to
//                    global memory exactly once (zeros included, so no memset).
//  k_hough_peaks   : 4-neighbour local maxima above the threshold -> 64-bit keys.
//
// rho = round_half_even(x*cos + y*sin) with float32 multiply/add rounded
// separately (the file is built with -ffp-contract=off and uses __fmul_rn /
// __fadd_rn), matching a scalar SSE build of OpenCV.
#include "common.hpp"
#include "gray_core.hpp"

#include <math.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace {

#define IPA_MAX_ANGLES 448

struct Trig {
    float s[IPA_MAX_ANGLES];
    float c[IPA_MAX_ANGLES];
};

__global__ __launch_bounds__(256) void k_hough_compact(Plane src, int w, int h, int cpr, int chunks_per_img,
                                                       u32* __restrict__ pts, size_t pts_stride, u32 cap,
                                                       int32_t* __restrict__ counts)
{
    __shared__ u32 wave_tot[4];
    __shared__ u32 s_base;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* out = pts + (size_t)b * pts_stride;
    const int step = (int)(gridDim.x * blockDim.x);  // <= 2^21; the host keeps chunks_per_img < 2^31 - 2^22
    for (int base = (int)(blockIdx.x * blockDim.x); base < chunks_per_img; base += step) {
        const int c = base + tid;
        u32 d[4] = {0, 0, 0, 0};
        int n = 0, cx = 0, y = 0;
        if (c < chunks_per_img) {
            y = c / cpr;
            cx = c - y * cpr;
            n = min(16, w - cx * 16);
            uint4 a = *(const uint4*)(src.row(b, y) + (size_t)cx * 16);
            d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
        }
        u32 mask = 0;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (i < n && ((d[i >> 2] >> (8 * (i & 3))) & 0xffu)) mask |= 1u << i;
        u32 cnt = __popc(mask);
        // inclusive scan over the wave
        u32 scan = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            u32 t = __shfl_up(scan, o);
            if (lane >= o) scan += t;
        }
        if (lane == 63) wave_tot[wave] = scan;
        __syncthreads();
        u32 before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 t = wave_tot[k];
            if (k < wave) before += t;
            total += t;
        }
        if (tid == 0) s_base = total ? (u32)atomicAdd(&counts[b], (int32_t)total) : 0u;
        __syncthreads();
        u32 pos = s_base + before + scan - cnt;
        u32 xy = ((u32)y << 16) | (u32)(cx * 16);
        while (mask) {
            int i = __ffs(mask) - 1;
            mask &= mask - 1;
            if (pos < cap) out[pos] = xy + (u32)i;
            pos++;
        }
        __syncthreads();  // wave_tot / s_base reused next round
    }
}

// Compaction with one global atomic per block of 4 rows x 4096 pixels.  blockIdx.x is the
// IMAGE, so the workgroups resident at any time append to many different lists (atomics on
// one address are served one at a time).  Wave `ty` owns one row; lane tx reads the 16-pixel
// chunks tx, tx+64, tx+128, tx+192 of the block's span, all four loads in flight together,
// and the points are listed in raster order (the voting kernel is fastest when the points
// of one vote instruction are spread along rows rather than clustered in two dimensions).
// Points are staged in LDS and appended with coalesced stores (written directly when a
// block holds more than the stage).
constexpr int CP_STAGE = 4096;

// bytes of 0 / 1 -> 4 adjacent bits (one dot product instead of a 32-bit multiply)
__device__ __forceinline__ u32 bytes_to_bits(u32 x) { return __builtin_amdgcn_udot4(x, 0x08040201u, 0u, false); }
// inclusive sum over the 64 lanes of a wave with cross-lane adds only (no LDS permutes): four shifts inside
// each row of 16 lanes, then lane 15 of a row into the next row and lane 31 into the upper half
__device__ __forceinline__ u32 wave_scan_add(u32 v)
{
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
// 1 in bit 0 of every non-zero byte
__device__ __forceinline__ u32 nonzero_bytes(u32 v) { return ((v | ((v & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u; }

// NW = waves (= image rows) per block: 4 for batches; 16 for a few images, whose blocks would otherwise
// queue up at the image's one counter.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_hough_compact2(Plane src, int w, int h, int cpr, u32* __restrict__ pts,
                                                             size_t pts_stride, u32 cap, int32_t* __restrict__ counts, int spread,
                                                             int32_t* __restrict__ rowcnt, size_t rowcnt_stride)
{
    __shared__ u32 s_pts[CP_STAGE];
    __shared__ u32 row_tot[NW * ROWS_PER_LANE];
    __shared__ u32 s_base;
    const int b = blockIdx.x;
    const int lane = threadIdx.x, wave = threadIdx.y, tid = wave * 64 + lane;
    const int cx0 = (int)blockIdx.y * 64 * ROWS_PER_LANE + lane;
    // the four rows of a block are a quarter of the image apart
    const int y = (int)blockIdx.z + wave * (int)gridDim.z;
    uint4 a[ROWS_PER_LANE];
#pragma unroll
    for (int k = 0; k < ROWS_PER_LANE; k++) {
        a[k] = make_uint4(0, 0, 0, 0);
        if (cx0 + 64 * k < cpr && y < h) a[k] = *(const uint4*)(src.row(b, y) + (size_t)(cx0 + 64 * k) * 16);
    }
    // one wave scan per 1024-pixel piece of the row
    u32 m[ROWS_PER_LANE], excl[ROWS_PER_LANE];
#pragma unroll
    for (int k = 0; k < ROWS_PER_LANE; k++) {
        const int n = w - (cx0 + 64 * k) * 16;  // pixels of the chunk inside the image
        const u32 keep = n >= 16 ? 0xffffu : (n > 0 ? (1u << n) - 1u : 0u);
        m[k] = (bytes_to_bits(nonzero_bytes(a[k].x)) | (bytes_to_bits(nonzero_bytes(a[k].y)) << 4) |
                (bytes_to_bits(nonzero_bytes(a[k].z)) << 8) | (bytes_to_bits(nonzero_bytes(a[k].w)) << 12)) & keep;
        const u32 cnt = __popc(m[k]);
        const u32 scan = wave_scan_add(cnt);
        excl[k] = scan - cnt;
        if (lane == 63) row_tot[wave * ROWS_PER_LANE + k] = scan;
    }
    __syncthreads();
    u32 total = 0, row_base[ROWS_PER_LANE] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < NW * ROWS_PER_LANE; r++) {
        const u32 t = row_tot[r];
#pragma unroll
        for (int k = 0; k < ROWS_PER_LANE; k++)
            if (r == wave * ROWS_PER_LANE + k) row_base[k] = total;
        total += t;
    }
    if (total == 0) return;  // uniform over the block
    // non-zero pixels per image row (zeroed by the host): the vote takes the angles whose bin depends on the row
    // alone from these counts instead of from the pixels
    if (rowcnt && lane == 0 && y < h) {
        u32 t = 0;
#pragma unroll
        for (int k = 0; k < ROWS_PER_LANE; k++) t += row_tot[wave * ROWS_PER_LANE + k];
        if (t) atomicAdd(&rowcnt[(size_t)b * rowcnt_stride + y], (int32_t)t);
    }
    if (tid == 0) s_base = (u32)atomicAdd(&counts[b], (int32_t)total);
    u32* out = pts + (size_t)b * pts_stride;
    const bool staged = total <= (u32)CP_STAGE;
    u32 gbase = 0;
    if (!staged) {
        __syncthreads();
        gbase = s_base;
    }
#pragma unroll
    for (int k = 0; k < ROWS_PER_LANE; k++) {
        const u32 xy = ((u32)y << 16) | (u32)((cx0 + 64 * k) * 16);
        u32 pos = gbase + row_base[k] + excl[k];
        u32 mk = m[k];
        while (mk) {
            const int i = __ffs(mk) - 1;
            mk &= mk - 1;
            if (staged)
                s_pts[pos] = xy + (u32)i;
            else if (pos < cap)
                out[pos] = xy + (u32)i;
            pos++;
        }
    }
    if (staged) {
        __syncthreads();
        const u32 base = s_base;
        // Stored entry p holds staged point (p * M) mod total, M a prime that does not divide
        // total: neighbouring list entries are then M raster positions apart, so the 64 points
        // of one vote instruction lie on different features and rarely share an LDS bin.
        u32 M = 1;
        if (spread && total >= 128) M = total % 61 ? 61 : (total % 67 ? 67 : 71);
        for (u32 i = tid; i < total; i += 64 * NW)
            if (base + i < cap) out[base + i] = s_pts[(i * M) % total];
    }
}

struct VoteArgs {
    const u32* pts;
    size_t pts_stride;
    u32 cap;
    const int32_t* counts;
    int32_t* accum;
    size_t accum_stride;
    int w, h, numangle, numrho;
    int tg;        // number of angles owned by one workgroup
    int groups;    // ceil(numangle / tg)
    int nseg;      // slices of the padded rho axis
    int seg_cols;  // columns (of numrho+2) per slice
    int span_max;  // LDS bins per angle
    int seg_walk;  // 1: each lane walks a contiguous segment of the point list
    int32_t* segmax;  // optional: per image, angle and 512-column segment the largest count
    int nsegs;        // segments per accumulator row
    int xcd_remap;    // 1: relabel workgroups so that an image's workgroups share an XCD
    int slice_major;    // k_hough_vote_s with slices > 1: 1 = work items ordered by slice first, then by angle group
    int slices;         // k_hough_vote_s: workgroups that share one (image, angle group); > 1: they add their bins to a zeroed accumulator
    int dbg_only_grp;   // experiments: only this angle group votes (-1 = all)
    u32 dbg_trip_mask;  // experiments: list trips wrapped into a small window (0xffffffff = off)
    int comb;           // k_hough_vote_s: 1 = the lanes of a 16-lane group take list entries K apart, K by the angle (see load4)
    u32 items;        // k_hough_vote_s: (image, angle group) work items = groups * images of the launch
    // k_hough_vote_s: angles whose bin is the same for every pixel of an image row (bit n of rowdet: angle n; checked
    // on the host for this image size): they are voted from the per-row pixel counts, one add per row
    const int32_t* rowcnt;
    size_t rowcnt_stride;
    u32 rowdet[IPA_MAX_ANGLES / 32];
};

__device__ __forceinline__ int rho_index(float x, float y, float c, float s)
{
    return __float2int_rn(__fadd_rn(__fmul_rn(x, c), __fmul_rn(y, s)));
}

constexpr int VOTE_THREADS = 512;
constexpr int HOUGH_SEG = VOTE_THREADS;  // columns per segment of the per-row maxima
constexpr int VOTE_MAX_TG = 16;
#ifndef VOTE_STAGGER_TRIPS
#define VOTE_STAGGER_TRIPS 4
#endif
#ifndef VOTE_PREFETCH
#define VOTE_PREFETCH 2
#endif

typedef float f2 __attribute__((ext_vector_type(2)));

// One vote.  PACK16 packs two 16-bit bins per LDS word.
template <bool PACK16>
__device__ __forceinline__ void vote1(u32* __restrict__ h, int idx)
{
    if (PACK16)
        atomicAdd(&h[idx >> 1], 1u << ((idx & 1) * 16));
    else
        atomicAdd(&h[idx], 1u);
}

// LDS-privatised voting.  One workgroup = one image x `tg` consecutive angles
// (x one slice of the rho axis for huge images).  PACK16 packs two 16-bit bins
// per LDS word (valid while no bin can reach 65536 votes).  CHECK adds the range
// test needed only when the rho axis is sliced.  The point list is read four
// points per lane with the next load in flight while the current points vote;
// rho for two points at a time with packed float multiplies/adds (each still
// rounded separately, as the scalar form).
template <bool PACK16, bool CHECK>
__global__ __launch_bounds__(VOTE_THREADS) void k_hough_vote(VoteArgs A, Trig T)
{
    extern __shared__ __align__(16) u32 hist[];
    __shared__ int s_col0[VOTE_MAX_TG];  // first padded column held in LDS, per angle of the group
    __shared__ float s_cos[VOTE_MAX_TG], s_sin[VOTE_MAX_TG];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int grp = blockIdx.x / A.nseg, seg = blockIdx.x % A.nseg;
    const int n0 = grp * A.tg;
    const int tg = min(A.tg, A.numangle - n0);
    const int half = (A.numrho - 1) / 2;
    const int seg_lo = seg * A.seg_cols, seg_hi = min(seg_lo + A.seg_cols, A.numrho + 2);
    const int span = A.span_max;
    const int span_words = PACK16 ? (span + 1) >> 1 : span;

    if (tid < tg) {
        // smallest reachable rho for this angle: the float ops are monotone, so
        // the extreme lies at a corner of the image
        float c = T.c[n0 + tid], s = T.s[n0 + tid];
        float xw = (float)(A.w - 1), yh = (float)(A.h - 1);
        int r00 = rho_index(0.f, 0.f, c, s), r10 = rho_index(xw, 0.f, c, s);
        int r01 = rho_index(0.f, yh, c, s), r11 = rho_index(xw, yh, c, s);
        int rmin = min(min(r00, r10), min(r01, r11));
        int col0 = rmin + half + 1;  // padded column of rmin
        s_col0[tid] = max(col0, seg_lo);
        s_cos[tid] = c;
        s_sin[tid] = s;
    }
    for (int i = tid; i < tg * span_words; i += VOTE_THREADS) hist[i] = 0;
    __syncthreads();

    const u32 npts = min((u32)A.counts[b], A.cap);
    const u32* pts = A.pts + (size_t)b * A.pts_stride;
    const uint4* pts4 = (const uint4*)pts;
    const u32 nfull = npts >> 2;  // whole groups of four points
    // Every lane walks its own contiguous segment of the list, so the 64 lanes of a wave
    // are always at 64 places spread over the whole image: neighbouring pixels of a line
    // (equal or adjacent rho for many angles) never vote in the same instruction, which
    // would serialise on one LDS bank.
    const u32 seglen = A.seg_walk ? (nfull + VOTE_THREADS - 1) / VOTE_THREADS : 1u;
    const u32 stride = A.seg_walk ? 1u : (u32)VOTE_THREADS;
    u32 i = A.seg_walk ? tid * seglen : tid;
    const u32 iend = A.seg_walk ? min(i + seglen, nfull) : nfull;
    uint4 cur = make_uint4(0, 0, 0, 0);
    if (i < iend) cur = pts4[i];
    while (i < iend) {
        const u32 inext = i + stride;
        uint4 nxt = make_uint4(0, 0, 0, 0);
        if (inext < iend) nxt = pts4[inext];
        const f2 x01 = {(float)(cur.x & 0xffffu), (float)(cur.y & 0xffffu)};
        const f2 x23 = {(float)(cur.z & 0xffffu), (float)(cur.w & 0xffffu)};
        const f2 y01 = {(float)(cur.x >> 16), (float)(cur.y >> 16)};
        const f2 y23 = {(float)(cur.z >> 16), (float)(cur.w >> 16)};
        for (int g = 0; g < tg; g++) {
            const float c = s_cos[g], s = s_sin[g];
            u32* hg = hist + g * span_words;
            // round-half-even through the float adder: for |r| < 2^22 the sum r + 1.5*2^23 is
            // rounded to an integer and its low mantissa bits hold that integer
            const f2 magic = {12582912.0f, 12582912.0f};
            const int off = half + 1 - s_col0[g] - 0x4B400000;
            const f2 r01 = (x01 * c + y01 * s) + magic;  // -ffp-contract=off: every op rounds on its own
            const f2 r23 = (x23 * c + y23 * s) + magic;
            const int i0 = __float_as_int(r01.x) + off, i1 = __float_as_int(r01.y) + off;
            const int i2 = __float_as_int(r23.x) + off, i3 = __float_as_int(r23.y) + off;
            if (CHECK) {
                const int lim = min(span, seg_hi - s_col0[g]);
                if ((unsigned)i0 < (unsigned)lim) vote1<PACK16>(hg, i0);
                if ((unsigned)i1 < (unsigned)lim) vote1<PACK16>(hg, i1);
                if ((unsigned)i2 < (unsigned)lim) vote1<PACK16>(hg, i2);
                if ((unsigned)i3 < (unsigned)lim) vote1<PACK16>(hg, i3);
            } else {
                vote1<PACK16>(hg, i0);
                vote1<PACK16>(hg, i1);
                vote1<PACK16>(hg, i2);
                vote1<PACK16>(hg, i3);
            }
        }
        cur = nxt;
        i = inext;
    }
    // the last 0..3 points
    for (u32 j = nfull * 4 + tid; j < npts; j += VOTE_THREADS) {
        const u32 p = pts[j];
        const float x = (float)(p & 0xffffu), y = (float)(p >> 16);
        for (int g = 0; g < tg; g++) {
            const int idx = rho_index(x, y, s_cos[g], s_sin[g]) + half + 1 - s_col0[g];
            const int lim = min(span, seg_hi - s_col0[g]);
            if ((unsigned)idx < (unsigned)lim) vote1<PACK16>(hist + g * span_words, idx);
        }
    }
    __syncthreads();

    // write this workgroup's slice of its accumulator rows (+ the two border rows)
    int32_t* acc = A.accum + (size_t)b * A.accum_stride;
    const int ncols = A.numrho + 2;
    for (int g = 0; g < tg; g++) {
        int32_t* row = acc + (size_t)(n0 + g + 1) * ncols;
        const int c0 = s_col0[g];
        const u32* hg = hist + g * span_words;
        for (int col = seg_lo + tid; col < seg_hi; col += VOTE_THREADS) {
            int idx = col - c0;
            int32_t v = 0;
            if ((unsigned)idx < (unsigned)span)
                v = PACK16 ? (int32_t)((hg[idx >> 1] >> ((idx & 1) * 16)) & 0xffffu) : (int32_t)hg[idx];
            row[col] = v;
        }
    }
    if (grp == 0)
        for (int col = seg_lo + tid; col < seg_hi; col += VOTE_THREADS) acc[col] = 0;
    if (grp == A.groups - 1)
        for (int col = seg_lo + tid; col < seg_hi; col += VOTE_THREADS) acc[(size_t)(A.numangle + 1) * ncols + col] = 0;
}

// Voting with the per-angle constants in scalar registers.  TG angles per workgroup are
// unrolled at compile time: cos/sin come straight from the kernel arguments (scalar loads),
// the LDS byte offset of each angle's row is made wave-uniform with readfirstlane, and one
// vote costs two packed float instructions (multiply pair, add, add of the rounding constant,
// each shared by two points) plus one shift-add for the LDS address.  NT = threads per
// workgroup: every workgroup streams the image's whole point list (from L2), so more angles per
// workgroup mean less list traffic, and 1024 threads keep the CU full with two such workgroups.
// One angle per workgroup (TG = 1) is the shape of very large images, whose rho window only fits
// once into a CU's LDS: that single workgroup has the registers of a whole CU, and it loads eight
// trips ahead because nothing else on the CU hides the latency of its list.
// NG > 0: the last NG waves of the workgroup do not vote.  While the others work through the point list they
// convert the workgroup's share of a BGR batch (GrayJob: the next part of a pipelined front-end pass) to gray,
// thresholded plane and bit plane -- a streaming job bound by HBM beside the vote, which is bound by LDS atomics
// and vector instructions and touches memory only for its list.  They join the others again for the parts every
// thread shares (clearing and flushing the bins).  NTV = threads that vote.
template <int TG, int NT, int NG>
__device__ __forceinline__ void vote_body(const VoteArgs& A, const Trig& T, const GrayJob* J)
{
    constexpr int NTV = NT - 64 * NG;
    static_assert(NTV >= 256 && NTV % 64 == 0, "whole waves vote");
    extern __shared__ __align__(16) u32 hist[];
    const int tid = threadIdx.x;
    // all workgroups of one image (they stream the same point list) on one XCD
    // The grid is one-dimensional and may be smaller than the number of work items (a resident
    // grid that keeps its share of every CU while kernels of other streams come and go); its size is
    // a multiple of 8 then, so that a workgroup's items stay on its XCD.
  for (u32 item = blockIdx.x; item < A.items; item += gridDim.x) {
    const u32 wg = A.xcd_remap ? xcd_contiguous_id(item, A.items) : item;
    const u32 per_image = (u32)A.groups * (u32)A.slices;
    const int b = (int)(wg / per_image);
    const u32 rem = wg - (u32)b * per_image;
    int grp = (int)(rem / (u32)A.slices), slice = (int)(rem - (u32)grp * (u32)A.slices);
    if (A.slice_major) {  // neighbouring workgroups (one XCD) take the same slice of the list at different angles
        slice = (int)(rem / (u32)A.groups);
        grp = (int)(rem - (u32)slice * (u32)A.groups);
    }
    if (A.dbg_only_grp >= 0 && grp != A.dbg_only_grp) continue;
    const int n0 = grp * TG;
    const int tg = min(TG, A.numangle - n0);
    const int half = (A.numrho - 1) / 2;
    const int span = A.span_max;
    const int ncols = A.numrho + 2;

    float cg[TG], sg[TG];
    int col0[TG];
    u32 k4[TG];  // byte offset such that (float bits of r + 1.5*2^23) * 4 + k4 addresses the bin
    bool byrow[TG];  // this angle is voted from the row counts
#pragma unroll
    for (int g = 0; g < TG; g++) {
        const int n = min(n0 + g, A.numangle - 1);
        byrow[g] = A.rowcnt != nullptr && ((A.rowdet[n >> 5] >> (n & 31)) & 1u) != 0;
        const float c = T.c[n], s = T.s[n];
        // smallest reachable rho: the float ops are monotone, so it lies at a corner
        const float xw = (float)(A.w - 1), yh = (float)(A.h - 1);
        const int r00 = rho_index(0.f, 0.f, c, s), r10 = rho_index(xw, 0.f, c, s);
        const int r01 = rho_index(0.f, yh, c, s), r11 = rho_index(xw, yh, c, s);
        const int rmin = __builtin_amdgcn_readfirstlane(min(min(r00, r10), min(r01, r11)));
        cg[g] = c;
        sg[g] = s;
        col0[g] = rmin + half + 1;
        // one opaque scalar per angle, so that the bin address is a single shift-add
        u32 k4g = (u32)__builtin_amdgcn_readfirstlane((int)((u32)(half + 1 - col0[g] - 0x4B400000 + g * span) * 4u));
        asm volatile("" : "+s"(k4g));  // keeps the compiler from splitting the sum again
        k4[g] = k4g;
    }
    for (int i = tid; i < TG * span; i += NT) hist[i] = 0;
    __syncthreads();
    // angles that vote from the points, as one scalar mask (a flag per angle kept in a vector condition cost two
    // vector instructions per angle and trip to test)
    u32 active = 0;
#pragma unroll
    for (int g = 0; g < TG; g++)
        if (g < tg && !byrow[g]) active |= 1u << g;
    active = (u32)__builtin_amdgcn_readfirstlane((int)active);

    if (NG == 0 || tid < NTV) {  // (wave-uniform) the waves that vote
    const u32 npts = min((u32)A.counts[b], A.cap);
    const u32* pts = A.pts + (size_t)b * A.pts_stride;
    const uint4* pts4 = (const uint4*)pts;
    const u32 nfull = npts >> 2;
    char* const hb = (char*)hist;
    const f2 magic = {12582912.0f, 12582912.0f};
    // Four consecutive list entries per lane and trip (one 16-byte load).  The loads run PF trips
    // ahead of the votes: the list comes from L2 when the workgroup has the chip to itself, but
    // from a busy HBM when a streaming kernel of another stream shares the CU.
    constexpr int PF = TG == 1 ? 8 : VOTE_PREFETCH;
    // trips in which every lane has four points; a slice takes its share of them, the last slice
    // also what is left over behind them
    const u32 all_trips = nfull / NTV;
    const u32 trip0 = (u32)((unsigned long long)all_trips * (u32)slice / (u32)A.slices);
    const u32 trips = (u32)((unsigned long long)all_trips * (u32)(slice + 1) / (u32)A.slices) - trip0;
    const bool last_slice = slice == A.slices - 1;
    // Workgroups of one image walk the same list.  In lockstep they would all ask the same L2
    // channel for the same lines at the same moment (measured on a single 16k x 16k image: the list
    // arrives at 1.2 TB/s, one channel's worth per XCD), so every angle group starts a few trips
    // further into the list and wraps around: neighbours still find each other's lines in L2.
    const u32 rot = trips ? ((u32)grp * VOTE_STAGGER_TRIPS) % trips : 0u;
    // Which four consecutive list entries of a trip's window (4 * NT entries) a lane takes.  The LDS serves an
    // atomic instruction 16 lanes at a time, each group as slowly as its fullest bank.  On lists in raster order
    // the entries of a horizontal stroke are consecutive pixels: with the 16 lanes of a group K entries apart,
    // K |cos theta| in (1, 2], their bins are 16 different words less than 32 apart -- no conflict -- for every
    // angle but 90 degrees (voted from the row counts).  Entries off such strokes behave like random ones as before.
    // window entry = q * 16 K + r * K + s with r = lane within its group and (q, s) = divmod(4 * group + j, K).
    u32 comb_k = 4;  // (the plain mapping is the comb with K = 4)
    if (A.comb) {
        const float ca = fabsf(cg[tg > 1 ? 1 : 0]);
        comb_k = ca > 0.5f ? 2u : ca > 0.25f ? 4u : ca > 0.125f ? 8u : ca > 0.0625f ? 16u : ca > 0.03125f ? 32u : 64u;
        comb_k = (u32)__builtin_amdgcn_readfirstlane((int)comb_k);
        while (comb_k > 4 && (4u * NTV) % (16u * comb_k) != 0) comb_k >>= 1;  // (a window holds whole combs)
    }
    const u32 lane16 = (u32)tid & 15u, grp16 = (u32)tid >> 4;
    // two 8-byte loads: K >= 4: adjacent entries; K = 2: the second pair lies 32 entries further on
    const u32 ent_a = comb_k >= 4 ? ((4u * grp16) / comb_k) * 16u * comb_k + lane16 * comb_k + ((4u * grp16) % comb_k)
                                  : (2u * grp16) * 32u + lane16 * 2u;
    const u32 pair_lo = ent_a >> 1, pair_hi = (comb_k >= 4 ? ent_a + 2u : ent_a + 32u) >> 1;
    const uint2* const pts2 = (const uint2*)pts;
    auto load4 = [&](u32 trip) -> uint4 {
        u32 t = trip + rot;  // trip < trips
        t = t >= trips ? t - trips : t;
        const uint2* p2 = pts2 + (size_t)(trip0 + (t & A.dbg_trip_mask)) * (2u * NTV);  // the trip's window
        const uint2 lo = p2[pair_lo], hi = p2[pair_hi];
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    auto cast4 = [&](const uint4 cur) {
        const f2 x01 = {(float)(cur.x & 0xffffu), (float)(cur.y & 0xffffu)};
        const f2 x23 = {(float)(cur.z & 0xffffu), (float)(cur.w & 0xffffu)};
        const f2 y01 = {(float)(cur.x >> 16), (float)(cur.y >> 16)};
        const f2 y23 = {(float)(cur.z >> 16), (float)(cur.w >> 16)};
        auto vote = [&](int g) {
            const f2 r01 = (x01 * cg[g] + y01 * sg[g]) + magic;  // every op rounds on its own
            const f2 r23 = (x23 * cg[g] + y23 * sg[g]) + magic;
            atomicAdd((u32*)(hb + (((u32)__float_as_int(r01.x) << 2) + k4[g])), 1u);
            atomicAdd((u32*)(hb + (((u32)__float_as_int(r01.y) << 2) + k4[g])), 1u);
            atomicAdd((u32*)(hb + (((u32)__float_as_int(r23.x) << 2) + k4[g])), 1u);
            atomicAdd((u32*)(hb + (((u32)__float_as_int(r23.y) << 2) + k4[g])), 1u);
        };
        if (active == (1u << TG) - 1u) {  // the usual case: all angles of the group, no branch per angle
#pragma unroll
            for (int g = 0; g < TG; g++) vote(g);
        } else {
#pragma unroll
            for (int g = 0; g < TG; g++)
                if ((active >> g) & 1u) vote(g);
        }
    };
    if (trips) {
        uint4 buf[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) buf[k] = load4(min((u32)k, trips - 1));
        u32 i = 0;
        for (; i + PF <= trips; i += PF) {
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const uint4 cur = buf[k];
                buf[k] = load4(min(i + PF + k, trips - 1));  // past the end: the last trip again, never used
                cast4(cur);
            }
        }
#pragma unroll
        for (int k = 0; k < PF; k++)
            if (i + k < trips) cast4(buf[k]);
    }
    if (last_slice && all_trips * NTV + tid < nfull) cast4(pts4[all_trips * NTV + tid]);
    const u32 tail0 = last_slice ? nfull * 4 : npts;
    // the points left over by the four-at-a-time loop
    for (u32 j = tail0 + tid; j < npts; j += NTV) {
        const u32 p = pts[j];
        const float x = (float)(p & 0xffffu), y = (float)(p >> 16);
#pragma unroll
        for (int g = 0; g < TG; g++)
            if (g < tg && !byrow[g]) atomicAdd(&hist[g * span + rho_index(x, y, cg[g], sg[g]) + half + 1 - col0[g]], 1u);
    }
    // angles whose bin depends on the row alone: every row's pixel count goes to the bin of its first pixel
    if (slice == 0) {
#pragma unroll
        for (int g = 0; g < TG; g++)
            if (g < tg && byrow[g]) {
                const int32_t* rc = A.rowcnt + (size_t)b * A.rowcnt_stride;
                for (int yy = tid; yy < A.h; yy += NTV) {
                    const u32 nrow = (u32)rc[yy];
                    if (nrow) atomicAdd(&hist[g * span + rho_index(0.f, (float)yy, cg[g], sg[g]) + half + 1 - col0[g]], nrow);
                }
            }
    }
    } else {
        // the spare waves: this item's share of the gray job, unit by unit (a unit = 64 consecutive chunks).  They
        // issue few instructions and wait for memory most of the time: with a raised priority their loads and
        // stores go out as soon as they can, which keeps HBM busy beside the voting waves.
        __builtin_amdgcn_s_setprio(3);
        const u32 u0 = (u32)((unsigned long long)J->total_units * wg / A.items), u1 = (u32)((unsigned long long)J->total_units * (wg + 1) / A.items);
        const u32 lane = (u32)tid & 63u;
        for (u32 u = u0 + (u32)((tid - NTV) >> 6); u < u1; u += (u32)NG) {
            const u32 bimg = u / J->units_per_image;
            const u32 ci = (u - bimg * J->units_per_image) * 64u + lane;
            if (ci < J->G.nchunks) gray_flat_chunk<true, true>(J->G, (int)bimg, ci);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();

    int32_t* acc = A.accum + (size_t)b * A.accum_stride;
#pragma unroll
    for (int g = 0; g < TG; g++) {
        if (g < tg) {
            int32_t* row = acc + (size_t)(n0 + g + 1) * ncols;
            const u32* hg = hist + g * span;
            if (A.slices == 1) {
                for (int col = tid; col < ncols; col += NT) {
                    const int idx = col - col0[g];
                    row[col] = (unsigned)idx < (unsigned)span ? (int32_t)hg[idx] : 0;
                }
            } else {
                // the accumulator was zeroed before the launch; slices add what they counted
                for (int idx = tid; idx < span; idx += NT) {
                    const u32 v = hg[idx];
                    const int col = idx + col0[g];
                    if (v && col < ncols) atomicAdd(&row[col], (int32_t)v);
                }
            }
        }
    }
    if (A.segmax) {
        // largest count of every HOUGH_SEG-column segment of every row: one wave per (row, segment)
        int32_t* sm = A.segmax + ((size_t)b * A.numangle + n0) * A.nsegs;
        const int lane = tid & 63, wave = tid >> 6;
        for (int q = wave; q < tg * A.nsegs; q += NT / 64) {
            const int g = q / A.nsegs, j = q - g * A.nsegs;
            int c0 = 0;
#pragma unroll
            for (int gg = 0; gg < TG; gg++)
                if (gg == g) c0 = col0[gg];
            u32 mx = 0;
#pragma unroll
            for (int t = 0; t < HOUGH_SEG / 64; t++) {
                const int idx = j * HOUGH_SEG + t * 64 + lane - c0;
                if ((unsigned)idx < (unsigned)span) mx = max(mx, hist[g * span + idx]);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (u32)__shfl_xor((int)mx, o));
            if (lane == 0) sm[q] = (int32_t)mx;
        }
    }
    if (A.slices == 1 && grp == 0)
        for (int col = tid; col < ncols; col += NT) acc[col] = 0;
    if (A.slices == 1 && grp == A.groups - 1)
        for (int col = tid; col < ncols; col += NT) acc[(size_t)(A.numangle + 1) * ncols + col] = 0;
    __syncthreads();  // the bins are cleared again for the next item
  }
}

template <int TG, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(TG == 1 ? 4 : 8, 8)))  // TG > 1: two 1024-thread workgroups per CU
void k_hough_vote_s(VoteArgs A, Trig T)
{
    vote_body<TG, NT, 0>(A, T, nullptr);
}

template <int TG, int NT, int NG>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_hough_vote_gray(VoteArgs A, Trig T, GrayJob J)
{
    vote_body<TG, NT, NG>(A, T, &J);
}

// Four cells per lane and trip (independent loads); the four neighbours are read only for
// the few cells above the threshold.
__global__ __launch_bounds__(256) void k_hough_peaks(const int32_t* __restrict__ accum, size_t accum_stride,
                                                     int numangle, int numrho, int threshold,
                                                     u64* __restrict__ keys, int capacity,
                                                     int32_t* __restrict__ counts)
{
    const int b = blockIdx.y;
    const int32_t* acc = accum + (size_t)b * accum_stride;
    const int ncols = numrho + 2;
    const int cells = numangle * numrho;
    const int lane = threadIdx.x & 63;
    const int step = gridDim.x * blockDim.x * 4;
    for (int base = blockIdx.x * blockDim.x * 4; base < cells; base += step) {
        int v[4], idx[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + k * 256 + threadIdx.x;
            v[k] = 0;
            idx[k] = 0;
            if (i < cells) {
                const int n = i / numrho, r = i - n * numrho;
                idx[k] = (n + 1) * ncols + r + 1;
                v[k] = acc[idx[k]];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            bool peak = false;
            if (v[k] > threshold) {
                const int id = idx[k];
                peak = v[k] > acc[id - 1] && v[k] >= acc[id + 1] && v[k] > acc[id - ncols] && v[k] >= acc[id + ncols];
            }
            const unsigned long long m = __ballot(peak);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                int32_t basepos = 0;
                if (lane == leader) basepos = atomicAdd(&counts[b], (int32_t)__popcll(m));
                basepos = __shfl(basepos, leader);
                if (peak) {
                    const int pos = basepos + (int)__popcll(m & ((1ull << lane) - 1ull));
                    if (pos < capacity)
                        keys[(size_t)b * capacity + pos] = ((u64)(u32)v[k] << 32) | (u64)(0xffffffffu - (u32)idx[k]);
                }
            }
        }
    }
}

// Peak search guided by the per-segment maxima the voting kernel left behind: a cell can
// only be a peak if it exceeds the threshold, so segments whose maximum does not are never
// read.  `split` waves share one accumulator row (wave s takes the segments j with j % split == s):
// 1 for large batches, more when a few images would otherwise leave most of the chip idle while
// the rows near 0 and 90 degrees walk through their many hot segments one after the other.
constexpr int PEAK_STAGE = 256;
__global__ __launch_bounds__(256) void k_hough_peaks_seg(const int32_t* __restrict__ accum, size_t accum_stride,
                                                         const int32_t* __restrict__ segmax, int nsegs, int numangle,
                                                         int numrho, int threshold, u64* __restrict__ keys, int capacity,
                                                         int32_t* __restrict__ counts, int split)
{
    const int b = blockIdx.y;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int widx = (int)blockIdx.x * 4 + wv;
    const int n = widx / split, part = widx - n * split;
    if (n >= numangle) return;
    const int lane = threadIdx.x & 63;
    const int ncols = numrho + 2;
    const int32_t* acc = accum + (size_t)b * accum_stride;
    const int32_t* row = acc + (size_t)(n + 1) * ncols;
    const int32_t* sm = segmax + ((size_t)b * numangle + n) * nsegs;
    // Peaks are collected per wave in LDS and appended PEAK_STAGE at a time: one atomic on the image's counter per
    // flush instead of one per 64 cells that hold a peak (atomics on one address are served one after the other,
    // which is what a single large, busy accumulator spends its time on otherwise).
    __shared__ u64 s_keys[4][PEAK_STAGE];
    int nq = 0;  // wave-uniform
    auto flush = [&]() {
        if (nq == 0) return;
        int32_t basepos = 0;
        if (lane == 0) basepos = atomicAdd(&counts[b], (int32_t)nq);
        basepos = __builtin_amdgcn_readfirstlane(basepos);
        for (int i = lane; i < nq; i += 64)
            if (basepos + i < capacity) keys[(size_t)b * capacity + basepos + i] = s_keys[wv][i];
        nq = 0;
    };
    for (int j0 = 0; j0 < nsegs; j0 += 64) {
      // 64 segment maxima at a time: one load, one ballot, then only the hot segments
      unsigned long long hot = __ballot(j0 + lane < nsegs && (j0 + lane) % split == part && sm[j0 + lane] > threshold);
      while (hot) {
        const int j = j0 + __ffsll((long long)hot) - 1;
        hot &= hot - 1;
        // all cells of the segment first (independent loads), then the neighbours of the few
        // cells above the threshold
        int32_t v[HOUGH_SEG / 64];
#pragma unroll
        for (int t = 0; t < HOUGH_SEG / 64; t++) {
            const int col = j * HOUGH_SEG + t * 64 + lane;
            v[t] = (col >= 1 && col <= numrho) ? row[col] : 0;
        }
#pragma unroll
        for (int t = 0; t < HOUGH_SEG / 64; t++) {
            const int col = j * HOUGH_SEG + t * 64 + lane;
            bool peak = false;
            if (v[t] > threshold && col >= 1 && col <= numrho)
                peak = v[t] > row[col - 1] && v[t] >= row[col + 1] && v[t] > row[col - ncols] && v[t] >= row[col + ncols];
            const unsigned long long m = __ballot(peak);
            if (m) {
                const int cnt = (int)__popcll(m);
                if (nq + cnt > PEAK_STAGE) flush();
                if (peak)
                    s_keys[wv][nq + (int)__popcll(m & ((1ull << lane) - 1ull))] =
                        ((u64)(u32)v[t] << 32) | (u64)(0xffffffffu - (u32)((n + 1) * ncols + col));
                nq += cnt;
            }
        }
      }
    }
    flush();
}

// Largest count of every HOUGH_SEG-column segment of every accumulator row, for the launches whose
// vote kernel cannot leave it behind (several workgroups add into one row).  One wave per segment.
__global__ __launch_bounds__(256) void k_hough_segmax(const int32_t* __restrict__ accum, size_t accum_stride, int numangle,
                                                      int numrho, int nsegs, int32_t* __restrict__ segmax)
{
    const int b = blockIdx.z, n = blockIdx.y;
    const int j = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= nsegs) return;
    const int ncols = numrho + 2;
    const int32_t* row = accum + (size_t)b * accum_stride + (size_t)(n + 1) * ncols;
    int32_t mx = 0;
#pragma unroll
    for (int t = 0; t < HOUGH_SEG / 64; t++) {
        const int col = j * HOUGH_SEG + t * 64 + lane;
        if (col < ncols) mx = max(mx, row[col]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    if (lane == 0) segmax[((size_t)b * numangle + n) * nsegs + j] = mx;
}

void hough_dims(int w, int h, double rho, double theta, int* numangle, int* numrho)
{
    float frho = (float)rho, ftheta = (float)theta;
    int max_rho = w + h, min_rho = -max_rho;
    int na = (int)floor((M_PI - 0.0) / ftheta) + 1;
    if (na > 1 && fabs(M_PI - (na - 1) * ftheta) < ftheta / 2) --na;
    *numangle = na;
    *numrho = (int)lrint(((max_rho - min_rho) + 1) / frho);
}

bool g_vote_attr_set = false;

// the row-block compaction (default) also counts the pixels of every row
bool compact_mode_rows()
{
    static int cmode = -1;  // IPA_COMPACT_MODE=0 selects the grid-stride kernel
    if (cmode < 0) {
        const char* e = getenv("IPA_COMPACT_MODE");
        cmode = e ? atoi(e) : 1;
    }
    return cmode == 1;
}

}  // namespace

IPA_EXPORT int ipa_hough_dims(int width, int height, double rho, double theta, int* numangle, int* numrho)
{
    IPA_REQUIRE(width >= 0 && height >= 0, IPA_ERR_ARG, "negative image size");
    IPA_REQUIRE(rho > 0 && theta > 0, IPA_ERR_ARG, "HoughLines: rho and theta must be positive");
    IPA_REQUIRE(numangle && numrho, IPA_ERR_ARG, "HoughLines: NULL output");
    hough_dims(width, height, rho, theta, numangle, numrho);
    return IPA_OK;
}

static u32 point_cap(int w, int h, int max_points)
{
    long long full = (long long)w * h;
    if (max_points > 0 && max_points < full) return (u32)max_points;
    return (u32)(full > 0xffffffffLL ? 0xffffffffLL : full);
}

// per-image row counts behind the point lists
static size_t rowcnt_stride_elems(int h) { return align_up((size_t)h * 4, 256) / 4; }

size_t hough_ws_bytes(int w, int h, int batch, int max_points)
{
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    return align_up((size_t)point_cap(w, h, max_points) * 4, 256) * (size_t)batch + rowcnt_stride_elems(h) * 4 * (size_t)batch + 256;
}

void hough_rowcount_area(int w, int h, int batch, int max_points, void* ws, u32** p, u32* dwords)
{
    *p = (u32*)((u8*)ws + align_up((size_t)point_cap(w, h, max_points) * 4, 256) * (size_t)batch);
    *dwords = (u32)(rowcnt_stride_elems(h) * (size_t)batch);
}

namespace {
// Angles for which every pixel of an image row falls into one bin: the float operations are monotone in x, so it is
// enough that the first and the last pixel of every row agree.  (theta = 90 degrees: cos is 4e-8, and all pixels
// of a long horizontal stroke would otherwise add to one LDS address, lane after lane.)
struct RowDetCache {
    int w = -1, h = -1, numangle = 0;
    float c0 = 0, s0 = 0, c1 = 0, s1 = 0;
    u32 mask[IPA_MAX_ANGLES / 32] = {};
};
void row_determined_angles(const Trig& T, int numangle, int w, int h, u32* mask)
{
    static std::mutex mu;
    static RowDetCache cache;
    std::lock_guard<std::mutex> lk(mu);
    const int last = numangle - 1;
    if (cache.w == w && cache.h == h && cache.numangle == numangle && cache.c0 == T.c[1 % numangle] && cache.s0 == T.s[1 % numangle] &&
        cache.c1 == T.c[last] && cache.s1 == T.s[last]) {
        memcpy(mask, cache.mask, sizeof(cache.mask));
        return;
    }
    memset(cache.mask, 0, sizeof(cache.mask));
    const volatile float xw = (float)(w - 1);
    for (int n = 0; n < numangle; n++) {
        const volatile float c = T.c[n], s = T.s[n];
        if (fabsf(c) * xw >= 0.75f) continue;
        bool same = true;
        for (int y = 0; y < h && same; y++) {
            const volatile float ys = (float)y * s;
            const volatile float xc = xw * c;
            const volatile float a = 0.0f * c + ys, b2 = xc + ys;
            same = nearbyintf(a) == nearbyintf(b2);
        }
        if (same) cache.mask[n >> 5] |= 1u << (n & 31);
    }
    cache.w = w; cache.h = h; cache.numangle = numangle;
    cache.c0 = T.c[1 % numangle]; cache.s0 = T.s[1 % numangle]; cache.c1 = T.c[last]; cache.s1 = T.s[last];
    memcpy(mask, cache.mask, sizeof(cache.mask));
}
}  // namespace

PointSink hough_point_sink(int w, int h, int max_points, int32_t* point_counts, void* ws)
{
    PointSink k;
    k.cap = point_cap(w, h, max_points);
    k.stride = align_up((size_t)k.cap * 4, 256) / 4;
    k.pts = (u32*)ws;
    k.counts = point_counts;
    return k;
}

// accum[b * stride + i] = 0 for i < cells, every image b = blockIdx.y
__global__ __launch_bounds__(256) void k_zero_i32(int32_t* __restrict__ accum, size_t stride, size_t cells)
{
    int32_t* p = accum + (size_t)blockIdx.y * stride;
    // dwords up to the first 16-byte boundary, then 16 bytes per lane, then what is left
    const size_t head = (size_t)((16 - ((uintptr_t)p & 15)) & 15) / 4;
    const size_t h0 = head < cells ? head : cells;
    const size_t n4 = (cells - h0) / 4;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
    if (gid < h0) p[gid] = 0;
    uint4* q = (uint4*)(p + h0);
    for (size_t i = gid; i < n4; i += gsz) q[i] = make_uint4(0, 0, 0, 0);
    const size_t t0 = h0 + n4 * 4;
    if (gid < cells - t0) p[t0 + gid] = 0;
}

// phase HOUGH_VOTE_ONLY: the per-image point lists (layout of hough_point_sink) and point_counts
// already exist (made by the Canny kernels or by an earlier HOUGH_COMPACT_ONLY call); otherwise they
// are compacted from `src` here.
int hough_segments(int numrho) { return cdiv(numrho + 2, HOUGH_SEG); }

// Lists in raster order (inside a compaction block) with the comb mapping of the vote kernel, or lists in the spread
// order with the plain mapping?  Measured on 4K diagrams: raster + comb wins from about 16 images on (64 images: vote
// 0.63 -> 0.54 ms, 1024 images: 9.06 -> 7.77 ms); few images, whose lists are cut into slices, are better off with
// the spread order (1 image: 48 against 53 us).  IPA_VOTE_COMB=0 switches it off (A/B).  Both phases of a pass must
// agree, so the rule uses nothing but the batch and the number of angles.
static bool raster_lists(int batch, int numangle)
{
    static int comb = -1;
    if (comb < 0) {
        const char* e = getenv("IPA_VOTE_COMB");
        comb = e ? atoi(e) : 1;
    }
    return comb != 0 && (long long)batch * cdiv(numangle, 4) >= 768;
}

int run_hough_accum(Plane src, int w, int h, int batch, double rho, double theta, int32_t* accum, size_t accum_stride,
                    int32_t* seg_max, int max_points, int32_t* point_counts, void* ws, size_t ws_bytes, int phase,
                    int resident_wgs, hipStream_t st, bool counts_cleared, const GrayJob* job, bool* job_done)
{
    if (job_done) *job_done = false;
    IPA_REQUIRE(w <= 65535 && h <= 65535, IPA_ERR_ARG, "HoughLines: image larger than 65535 pixels on a side");
    int numangle, numrho;
    hough_dims(w, h, rho, theta, &numangle, &numrho);
    IPA_REQUIRE(numangle >= 1 && numangle <= IPA_MAX_ANGLES, IPA_ERR_ARG,
                "HoughLines: %d angles (theta too small; at most %d supported)", numangle, IPA_MAX_ANGLES);
    IPA_REQUIRE(numrho >= 1, IPA_ERR_ARG, "HoughLines: rho too large");
    IPA_REQUIRE(accum_stride >= (size_t)(numangle + 2) * (numrho + 2) || batch == 1, IPA_ERR_ARG,
                "HoughLines: accumulator stride too small");
    IPA_REQUIRE(ws && ws_bytes >= hough_ws_bytes(w, h, batch, max_points), IPA_ERR_WORKSPACE,
                "HoughLines: workspace too small");
    const u32 cap = point_cap(w, h, max_points);
    const size_t pts_stride = align_up((size_t)cap * 4, 256) / 4;
    u32* pts = (u32*)ws;
    int32_t* const rowcnt = (int32_t*)((u8*)ws + pts_stride * 4 * (size_t)batch);
    const size_t rowcnt_stride = rowcnt_stride_elems(h);

    Trig T;
    {
        float frho = (float)rho, ftheta = (float)theta;
        float irho = 1 / frho;
        float ang = 0.f;
        for (int n = 0; n < numangle; ang += ftheta, n++) {
            T.s[n] = (float)(sin((double)ang) * irho);
            T.c[n] = (float)(cos((double)ang) * irho);
        }
    }
#ifdef IPA_EXPERIMENTS
    const bool revote = getenv("IPA_VOTE_ONLY") != nullptr;  // probes only: re-vote the lists of the previous call
#else
    const bool revote = false;
#endif
    if (phase != HOUGH_VOTE_ONLY && phase != HOUGH_VOTE_ONLY_FOREIGN_LISTS && !revote) {
        if (!counts_cleared) {
            IPA_HIP(hipMemsetAsync(point_counts, 0, sizeof(int32_t) * (size_t)batch, st));
            IPA_HIP(hipMemsetAsync(rowcnt, 0, rowcnt_stride * 4 * (size_t)batch, st));
        }
        ProfScope ps(PROF_HOUGH_COMPACT, st);
        const int cpr = cdiv(w, 16);
        const int cmode = compact_mode_rows() ? 1 : 0;  // IPA_COMPACT_MODE=0 selects the grid-stride kernel
        if (cmode == 1) {
            // The four rows of a block are gridDim.z apart.  An odd distance: rows a multiple of 32 apart
            // (any height that is a multiple of 128) put the bins of theta = 90 degrees of all four into
            // one LDS bank, and the vote kernel then serialises (measured: 10x on that angle).
            static int nw_env = -1;  // IPA_COMPACT_ROWS: 4 or 16 rows per block (default: by launch size)
            if (nw_env < 0) {
                const char* e = getenv("IPA_COMPACT_ROWS");
                nw_env = e ? atoi(e) : 0;
            }
            const int nw = nw_env == 4 || nw_env == 16 ? nw_env : ((long long)batch * cdiv(h, 4) < 16384 ? 16 : 4);  // rows per block
            const dim3 grid((unsigned)batch, (unsigned)cdiv(cpr, 64 * ROWS_PER_LANE), (unsigned)(cdiv(h, nw) | 1));
            static int spread_env = -2;  // IPA_COMPACT_SPREAD: 0 = raster order inside a block, 1 = spread order (default: by raster_lists)
            if (spread_env == -2) {
                const char* e = getenv("IPA_COMPACT_SPREAD");
                spread_env = e ? atoi(e) : -1;
            }
            const int spread = spread_env >= 0 ? spread_env : (raster_lists(batch, numangle) ? 0 : 1);
            static const size_t pad = lds_pad_kib("IPA_PAD_COMPACT");
            if (nw == 16)
                hipLaunchKernelGGL(k_hough_compact2<16>, grid, dim3(64, 16), pad, st, src, w, h, cpr, pts, pts_stride, cap, point_counts, spread,
                                   rowcnt, rowcnt_stride);
            else
                hipLaunchKernelGGL(k_hough_compact2<4>, grid, rows_block(), pad, st, src, w, h, cpr, pts, pts_stride, cap, point_counts, spread,
                                   rowcnt, rowcnt_stride);
            IPA_LAUNCH_CHECK();
        } else {
            IPA_REQUIRE((long long)cpr * h < 0x7fffffffLL - (1 << 22), IPA_ERR_ARG, "HoughLines: image too large");
            int cpi = cpr * h;
            unsigned gx = (unsigned)((cpi + 255) / 256);
            unsigned capx = (unsigned)(8192 / (batch < 8192 ? batch : 8192));
            if (capx < 1) capx = 1;
            if (gx > capx) gx = capx;
            for (int b0 = 0; b0 < batch; b0 += 65535) {
                int nb = batch - b0 < 65535 ? batch - b0 : 65535;
                Plane sb = src;
                sb.p += (size_t)b0 * src.stride;
                hipLaunchKernelGGL(k_hough_compact, dim3(gx, nb), dim3(256), 0, st, sb, w, h, cpr, cpi,
                                   pts + (size_t)b0 * pts_stride, pts_stride, cap, point_counts + b0);
                IPA_LAUNCH_CHECK();
            }
        }
    }
    if (phase == HOUGH_COMPACT_ONLY) return IPA_OK;
    // geometry of the vote kernel
    VoteArgs A;
    A.pts = pts; A.pts_stride = pts_stride; A.cap = cap; A.counts = point_counts;
    A.accum = accum; A.accum_stride = accum_stride;
    A.w = w; A.h = h; A.numangle = numangle; A.numrho = numrho;
    // bins an angle can reach: |x cos| + |y sin| <= (w-1 + h-1) / rho, plus rounding slack
    long long reach = (long long)ceil(((double)(w - 1) + (double)(h - 1)) / rho) + 3;
    double diag = ceil(sqrt((double)(w - 1) * (w - 1) + (double)(h - 1) * (h - 1)) / rho) + 3;
    if (diag < (double)reach) reach = (long long)diag;
    const int ncols = numrho + 2;
    if (reach > ncols) reach = ncols;
    const long long lds_budget_bins = (150 * 1024) / 4;  // up to 150 KiB of LDS per workgroup
    if (reach <= lds_budget_bins) {
        A.nseg = 1;
        A.span_max = (int)reach;
        A.seg_cols = ncols;
    } else {
        // one angle's bins do not fit: slice the padded rho axis
        A.nseg = cdiv(ncols, lds_budget_bins);
        A.seg_cols = cdiv(ncols, A.nseg);
        A.span_max = A.seg_cols;
    }
    // 16-bit bins (two per LDS word) while no bin can overflow: a bin never gets more
    // votes than there are pixels on one line through the image
    static int mode = -1;  // IPA_VOTE_MODE: 0 = 32-bit bins (default, fewest instructions per vote), 1 = 16-bit bins
    if (mode < 0) {
        const char* e = getenv("IPA_VOTE_MODE");
        mode = e ? atoi(e) : 0;
    }
    static int segw = -1;  // IPA_VOTE_SEGWALK: 1 = per-lane list segments, 0 = interleaved (default)
    if (segw < 0) {
        const char* e = getenv("IPA_VOTE_SEGWALK");
        segw = e ? atoi(e) : 0;
    }
    A.seg_walk = segw;
    const bool pack16 = ((long long)w + h) < 65535 && mode == 1;
    const bool check = A.nseg > 1;
    const int span_words = pack16 ? (A.span_max + 1) / 2 : A.span_max;
    // several angles per workgroup (up to 64 KiB of LDS: two workgroups per CU) so that
    // each pass over the point list casts more votes; fewer when the launch would
    // otherwise have too few workgroups to fill the chip
    static int lds_kb = -1;  // IPA_VOTE_LDS_KB: LDS budget per workgroup
    if (lds_kb < 0) {
        const char* e = getenv("IPA_VOTE_LDS_KB");
        lds_kb = e ? atoi(e) : 78;
        if (lds_kb < 8) lds_kb = 8;
        if (lds_kb > 150) lds_kb = 150;
    }
    int tg = (int)(((size_t)lds_kb * 1024 / 4) / span_words);
    if (tg < 1) tg = 1;
    if (tg > VOTE_MAX_TG) tg = VOTE_MAX_TG;
    while (tg > 1 && (long long)cdiv(numangle, tg) * A.nseg * batch < 1024) tg--;
    // the compile-time unrolled kernel exists for 1, 2, 4 and 8 angles per workgroup
    static int scalar_ok = -1;  // IPA_VOTE_SCALAR=0 selects the generic kernel
    if (scalar_ok < 0) {
        const char* e = getenv("IPA_VOTE_SCALAR");
        scalar_ok = e ? atoi(e) : 1;
    }
    static int vote_nt = -1;  // IPA_VOTE_THREADS: 512 or 1024 threads per workgroup
    if (vote_nt < 0) {
        const char* e = getenv("IPA_VOTE_THREADS");
        vote_nt = e ? atoi(e) : 1024;
        if (vote_nt != 512) vote_nt = 1024;
    }
    int tgs = 0;
    if (scalar_ok && !pack16 && !check && !A.seg_walk) tgs = tg >= 8 ? 8 : tg >= 4 ? 4 : tg >= 2 ? 2 : 1;
    if (tgs) tg = tgs;
    A.tg = tg;
    A.groups = cdiv(numangle, tg);
    // (measured: the list is then fetched from memory 2.4 times instead of 13.5 times per pass)
    static int remap = -1;  // IPA_VOTE_XCD=0 keeps the plain block order
    if (remap < 0) {
        const char* e = getenv("IPA_VOTE_XCD");
        remap = e ? atoi(e) : 1;
    }
    A.xcd_remap = remap;
    A.comb = phase != HOUGH_VOTE_ONLY_FOREIGN_LISTS && raster_lists(batch, numangle) ? 1 : 0;
    A.nsegs = hough_segments(numrho);
    // Few large images: (image, angle group) alone gives too few workgroups to fill the chip, and one
    // angle whose votes pile up on a few bins (theta = 90 degrees on ruled paper) holds everything
    // up.  The list is then cut into slices, one workgroup each, which add into a zeroed accumulator.
    A.slices = 1;
    if (tgs && resident_wgs == 0) {
        static int slices_env = -1;  // IPA_VOTE_SLICES: 0 = automatic
        if (slices_env < 0) {
            const char* e = getenv("IPA_VOTE_SLICES");
            slices_env = e ? atoi(e) : 0;
        }
        const long long wgs = (long long)A.groups * batch;
        if (slices_env > 0) A.slices = slices_env > 64 ? 64 : slices_env;
        else if ((long long)w * h >= (1 << 22) && wgs < 768) A.slices = (int)(1536 / wgs > 16 ? 16 : 1536 / wgs);
        if (A.slices < 1) A.slices = 1;
    }
    A.slice_major = 0;
    if (A.slices > 1) {
        // The workgroups of one XCD (consecutive work items) then stream the SAME slice of the list at their
        // different angles and find each other's lines in that XCD's L2; with the slices of one angle side by
        // side every XCD walks the whole list once per angle it owns (one 16384 x 16384 page with 14.5 M edge
        // pixels: 1.04 -> 0.76 ms).  IPA_VOTE_SLICE_MAJOR=0 restores the angle-major order (A/B).
        static int sm_env = -1;
        if (sm_env < 0) {
            const char* e = getenv("IPA_VOTE_SLICE_MAJOR");
            sm_env = e ? (atoi(e) != 0) : 1;
        }
        A.slice_major = sm_env;
    }
    if (A.slices > 1) {
        // one launch for all images (a runtime memset of a size that is no multiple of 16 bytes is two
        // kernels per image, and this path is the one of single images, whose pass is a chain of short
        // dependent launches)
        const size_t cells = (size_t)(numangle + 2) * (numrho + 2);
        hipLaunchKernelGGL(k_zero_i32, dim3((unsigned)(cells / 1024 < 2048 ? cells / 1024 + 1 : 2048), (unsigned)batch), dim3(256), 0, st,
                           accum, accum_stride, cells);
        IPA_LAUNCH_CHECK();
    }
    // angles voted from the row counts (only the kernels that know them, and only when this library's row-block
    // compaction made the lists)
    A.rowcnt = nullptr;
    A.rowcnt_stride = rowcnt_stride;
    memset(A.rowdet, 0, sizeof(A.rowdet));
    static int byrow_env = -1;  // IPA_VOTE_BYROW=0: every angle from the pixels (A/B)
    if (byrow_env < 0) {
        const char* e = getenv("IPA_VOTE_BYROW");
        byrow_env = e ? atoi(e) : 1;
    }
    if (tgs && byrow_env && phase != HOUGH_VOTE_ONLY_FOREIGN_LISTS && compact_mode_rows()) {
        row_determined_angles(T, numangle, w, h, A.rowdet);
        A.rowcnt = rowcnt;
    }
    A.segmax = (tgs && A.slices == 1) ? seg_max : nullptr;
    const bool segmax_after = seg_max && A.slices > 1;  // computed from the finished accumulator
    if (seg_max && !A.segmax && !segmax_after)  // this kernel variant does not produce the maxima: mark every segment as worth reading
        IPA_HIP(hipMemsetD32Async((hipDeviceptr_t)seg_max, 0x7fffffff, (size_t)batch * numangle * A.nsegs, st));
    static const size_t vote_pad = lds_pad_kib("IPA_PAD_VOTE");
    size_t lds = (size_t)tg * span_words * 4 + vote_pad;
    // a resident grid must not get two workgroups onto one CU: more than half of the CU's LDS each
    if (resident_wgs > 0 && lds < 81 * 1024) lds = 81 * 1024;
    if (lds > 48 * 1024 && !g_vote_attr_set) {
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote_s<1, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote_s<1, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote_s<2, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote_s<2, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote_s<4, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote_s<4, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote_s<8, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote_s<8, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote_gray<4, 1024, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        IPA_HIP(hipFuncSetAttribute((const void*)k_hough_vote_gray<4, 1024, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        g_vote_attr_set = true;
    }
    ProfScope ps(PROF_HOUGH_VOTE, st);
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        VoteArgs Ab = A;
        Ab.pts = pts + (size_t)b0 * pts_stride;
        Ab.counts = point_counts + b0;
        Ab.accum = accum + (size_t)b0 * accum_stride;
        if (A.segmax) Ab.segmax = A.segmax + (size_t)b0 * numangle * A.nsegs;
        if (A.rowcnt) Ab.rowcnt = A.rowcnt + (size_t)b0 * A.rowcnt_stride;
        const dim3 grid((unsigned)(A.groups * A.nseg), nb), block(VOTE_THREADS);
        Ab.items = (u32)A.groups * (u32)A.slices * (u32)nb;
        Ab.dbg_only_grp = -1;
        Ab.dbg_trip_mask = 0xffffffffu;
#ifdef IPA_EXPERIMENTS  // probes only (scripts/): part of the angles / of the list
        if (getenv("IPA_VOTE_ONLY_GRP")) Ab.dbg_only_grp = atoi(getenv("IPA_VOTE_ONLY_GRP"));
        if (getenv("IPA_VOTE_WRAP")) Ab.dbg_trip_mask = (u32)atoi(getenv("IPA_VOTE_WRAP"));
#endif
        static int resident_env = -1;  // IPA_VOTE_RESIDENT: force a resident grid of that many workgroups (multiple of 8)
        if (resident_env < 0) {
            const char* e = getenv("IPA_VOTE_RESIDENT");
            resident_env = e ? atoi(e) : 0;
        }
        const int resident = (resident_env > 0 ? resident_env : resident_wgs) & ~7;
        const dim3 grid_s(resident > 0 && (u32)resident < Ab.items ? (unsigned)resident : Ab.items);
        // a gray job rides along on four of the sixteen waves of the batch shape (four angles, 1024 threads, one slice)
        static int job_waves = -1;  // IPA_VOTE_GRAY_WAVES: 0 = never, 2 or 4 (default) spare waves
        if (job_waves < 0) {
            const char* e = getenv("IPA_VOTE_GRAY_WAVES");
            job_waves = e ? atoi(e) : 4;
            if (job_waves != 0 && job_waves != 2) job_waves = 4;
        }
        if (job && job_waves && tgs == 4 && vote_nt == 1024 && A.slices == 1 && resident == 0 && b0 == 0 && nb == batch) {
            if (job_waves == 4) hipLaunchKernelGGL((k_hough_vote_gray<4, 1024, 4>), grid_s, dim3(1024), lds, st, Ab, T, *job);
            else hipLaunchKernelGGL((k_hough_vote_gray<4, 1024, 2>), grid_s, dim3(1024), lds, st, Ab, T, *job);
            IPA_LAUNCH_CHECK();
            if (job_done) *job_done = true;
            continue;
        }
#define IPA_VOTE_S(TGV)                                                                               \
    do {                                                                                              \
        if (vote_nt == 1024)                                                                          \
            hipLaunchKernelGGL((k_hough_vote_s<TGV, 1024>), grid_s, dim3(1024), lds, st, Ab, T);       \
        else                                                                                          \
            hipLaunchKernelGGL((k_hough_vote_s<TGV, 512>), grid_s, dim3(512), lds, st, Ab, T);         \
    } while (0)
        if (tgs == 8) IPA_VOTE_S(8);
        else if (tgs == 4) IPA_VOTE_S(4);
        else if (tgs == 2) IPA_VOTE_S(2);
        else if (tgs == 1) IPA_VOTE_S(1);
#undef IPA_VOTE_S
        else if (pack16 && !check)
            hipLaunchKernelGGL((k_hough_vote<true, false>), grid, block, lds, st, Ab, T);
        else if (pack16)
            hipLaunchKernelGGL((k_hough_vote<true, true>), grid, block, lds, st, Ab, T);
        else if (!check)
            hipLaunchKernelGGL((k_hough_vote<false, false>), grid, block, lds, st, Ab, T);
        else
            hipLaunchKernelGGL((k_hough_vote<false, true>), grid, block, lds, st, Ab, T);
        IPA_LAUNCH_CHECK();
    }
    if (segmax_after) {
        for (int b0 = 0; b0 < batch; b0 += 65535) {
            const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
            hipLaunchKernelGGL(k_hough_segmax, dim3((unsigned)cdiv(A.nsegs, 4), (unsigned)numangle, (unsigned)nb), dim3(256), 0, st,
                               accum + (size_t)b0 * accum_stride, accum_stride, numangle, numrho, A.nsegs,
                               seg_max + (size_t)b0 * numangle * A.nsegs);
            IPA_LAUNCH_CHECK();
        }
    }
    return IPA_OK;
}

IPA_EXPORT size_t ipa_hough_workspace_bytes(int width, int height, int batch, int max_points)
{
    return hough_ws_bytes(width, height, batch, max_points);
}

IPA_EXPORT int ipa_hough_segments(int numrho) { return numrho >= 1 ? hough_segments(numrho) : 0; }

IPA_EXPORT int ipa_hough_accum_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, int width, int height,
                                  int batch, double rho, double theta, int32_t* accum, size_t accum_stride,
                                  int32_t* seg_max, int max_points, int32_t* point_counts, void* ws, size_t ws_bytes,
                                  void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    IPA_REQUIRE(rho > 0 && theta > 0, IPA_ERR_ARG, "HoughLines: rho and theta must be positive");
    if (batch == 0) return IPA_OK;
    IPA_REQUIRE(accum && point_counts, IPA_ERR_ARG, "HoughLines: NULL accumulator / counts");
    hipStream_t st = (hipStream_t)stream;
    if (width == 0 || height == 0) {
        int na, nr;
        hough_dims(width, height, rho, theta, &na, &nr);
        for (int b = 0; b < batch; b++)
            IPA_HIP(hipMemsetAsync(accum + (size_t)b * accum_stride, 0, sizeof(int32_t) * (size_t)(na + 2) * (nr + 2), st));
        IPA_HIP(hipMemsetAsync(point_counts, 0, sizeof(int32_t) * (size_t)batch, st));
        if (seg_max) IPA_HIP(hipMemsetAsync(seg_max, 0, sizeof(int32_t) * (size_t)batch * na * hough_segments(nr), st));
        return IPA_OK;
    }
    Plane s = mkplane(src, src_pitch, src_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch), IPA_ERR_ALIGN,
                "HoughLines: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    return run_hough_accum(s, width, height, batch, rho, theta, accum, accum_stride, seg_max, max_points, point_counts,
                           ws, ws_bytes, HOUGH_ALL, 0, st);
}

IPA_EXPORT int ipa_hough_peaks_i32(const int32_t* accum, size_t accum_stride, const int32_t* seg_max, int numangle,
                                   int numrho, int batch, int threshold, uint64_t* keys, int capacity,
                                   int32_t* peak_counts, void* stream)
{
    IPA_REQUIRE(numangle >= 1 && numrho >= 1 && batch >= 0 && capacity >= 0, IPA_ERR_ARG, "HoughLines: bad peak arguments");
    if (batch == 0) return IPA_OK;
    IPA_REQUIRE(accum && peak_counts && (keys || capacity == 0), IPA_ERR_ARG, "HoughLines: NULL peak buffers");
    hipStream_t st = (hipStream_t)stream;
    IPA_HIP(hipMemsetAsync(peak_counts, 0, sizeof(int32_t) * (size_t)batch, st));
    long long cells = (long long)numangle * numrho;
    IPA_REQUIRE((long long)(numangle + 2) * (numrho + 2) < 0x7fffffffLL, IPA_ERR_ARG, "HoughLines: accumulator too large");
    unsigned gx = (unsigned)((cells + 1023) / 1024);
    unsigned capx = (unsigned)(16384 / (batch < 16384 ? batch : 16384));
    if (capx < 1) capx = 1;
    if (gx > capx) gx = capx;
    ProfScope ps(PROF_HOUGH_PEAKS, st);
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        if (seg_max) {
            const int nsegs = hough_segments(numrho);
            // about 8192 waves at least
            int split = (int)(8192 / ((long long)batch * numangle));
            split = split < 1 ? 1 : (split > nsegs ? nsegs : split);
            hipLaunchKernelGGL(k_hough_peaks_seg, dim3((unsigned)cdiv(numangle * split, 4), nb), dim3(256), 0, st,
                               accum + (size_t)b0 * accum_stride, accum_stride, seg_max + (size_t)b0 * numangle * nsegs, nsegs,
                               numangle, numrho, threshold, (u64*)keys + (size_t)b0 * capacity, capacity, peak_counts + b0, split);
            IPA_LAUNCH_CHECK();
            continue;
        }
        hipLaunchKernelGGL(k_hough_peaks, dim3(gx, nb), dim3(256), 0, st, accum + (size_t)b0 * accum_stride,
                           accum_stride, numangle, numrho, threshold, (u64*)keys + (size_t)b0 * capacity, capacity,
                           peak_counts + b0);
        IPA_LAUNCH_CHECK();
    }
    return IPA_OK;
}
