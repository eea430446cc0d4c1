// Rectangular erode / dilate / open / close.
//
// Two families of kernels:
//  * bit-plane kernels for two-valued images (the thresholded planes of the
//    diagram parsers): the image is packed to 1 bit per pixel, a tile of it is
//    held in LDS, and a k-long min/max becomes log2(k)+1 shifted AND/OR steps on
//    32-pixel words (rows: funnel shifts, columns: whole-word row offsets).  Both
//    stages of an open/close run on the tile without leaving LDS; the result is
//    expanded back to bytes on the way out (16 pixels per 16-byte store).
//  * general 8-bit kernels (any gray levels): separable row / column passes.
//
// Semantics (OpenCV): window of output x is input [x-ax, x-ax+k-1]; samples
// outside the image are ignored; iterations of an all-ones kernel fold into one
// pass with k' = k + (n-1)(k-1), a' = a*n.
#include "common.hpp"

#include <stdlib.h>

namespace {

// ------------------------------------------------------------------ bit planes

// 8-bit plane -> bit plane (bit = pixel != 0), plus the largest pixel value of
// each image (the common non-zero value when the image is two-valued).
// bit 0 of each byte of x -> 4 adjacent bits
__device__ __forceinline__ u32 pb_bytes_to_bits(u32 x) { return ((x & 0x01010101u) * 0x10204080u) >> 28; }
// 1 in bit 0 of every non-zero byte
__device__ __forceinline__ u32 pb_nonzero_bytes(u32 v) { return ((v | ((v & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u; }
// per-byte maximum of two dwords, through the even / odd bytes as 16-bit pairs
__device__ __forceinline__ u32 pb_max_bytes(u32 a, u32 b)
{
    typedef unsigned short us2p __attribute__((ext_vector_type(2)));
    const us2p ae = __builtin_bit_cast(us2p, a & 0x00ff00ffu), be = __builtin_bit_cast(us2p, b & 0x00ff00ffu);
    const us2p ao = __builtin_bit_cast(us2p, (a >> 8) & 0x00ff00ffu), bo = __builtin_bit_cast(us2p, (b >> 8) & 0x00ff00ffu);
    return __builtin_bit_cast(u32, __builtin_elementwise_max(ae, be)) | (__builtin_bit_cast(u32, __builtin_elementwise_max(ao, bo)) << 8);
}

// 8-bit plane -> bit plane ("pixel is non-zero") plus the largest pixel value of each image.
// Thread (tx, ty) of block (bx, by, bz): 16-pixel chunk column 64 bx + tx of four rows; the
// wave's bits of its four rows leave through LDS as one 8-byte store per lane, and the
// maximum costs at most one atomic per workgroup.
__global__ __launch_bounds__(256) void k_pack_bits(Plane src, u32* __restrict__ bits, size_t bpw, size_t bsw, int w,
                                                   int h, int cpr, int32_t* __restrict__ vmax)
{
    __shared__ unsigned short s_bits[4][ROWS_PER_LANE][64];
    __shared__ u32 s_max[4];
    const int lane = threadIdx.x, wv = threadIdx.y;
    const int cx = blockIdx.x * 64 + lane;
    const int b = blockIdx.z;
    const int y0 = ((int)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(wv)) * ROWS_PER_LANE;
    const int n = w - cx * 16;  // pixels of the chunk inside the image
    const u32 keep = n >= 16 ? 0xffffu : (n > 0 ? (1u << n) - 1u : 0u);
    uint4 a[ROWS_PER_LANE];
#pragma unroll
    for (int i = 0; i < ROWS_PER_LANE; i++) {
        a[i] = make_uint4(0, 0, 0, 0);
        if (n > 0 && y0 + i < h) a[i] = *(const uint4*)(src.row(b, y0 + i) + (size_t)cx * 16);
    }
    u32 mx4 = 0;
#pragma unroll
    for (int i = 0; i < ROWS_PER_LANE; i++) {
        const u32 m = (pb_bytes_to_bits(pb_nonzero_bytes(a[i].x)) | (pb_bytes_to_bits(pb_nonzero_bytes(a[i].y)) << 4) |
                       (pb_bytes_to_bits(pb_nonzero_bytes(a[i].z)) << 8) | (pb_bytes_to_bits(pb_nonzero_bytes(a[i].w)) << 12)) & keep;
        s_bits[wv][i][lane] = (unsigned short)m;
        if (n >= 16) {
            mx4 = pb_max_bytes(mx4, pb_max_bytes(pb_max_bytes(a[i].x, a[i].y), pb_max_bytes(a[i].z, a[i].w)));
        } else if (n > 0) {
            const u32 d[4] = {a[i].x, a[i].y, a[i].z, a[i].w};
            for (int k = 0; k < n; k++) mx4 = max(mx4, (d[k >> 2] >> (8 * (k & 3))) & 0xffu);
        }
    }
    u32 mx = max(max(mx4 & 0xffu, (mx4 >> 8) & 0xffu), max((mx4 >> 16) & 0xffu, mx4 >> 24));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (u32)__shfl_xor((int)mx, o));
    if (lane == 0) s_max[wv] = mx;
    __syncthreads();  // also orders the s_bits writes before the reads below
    const int r = lane >> 4, c0 = (lane & 15) * 4;
    const uint2 v = *(const uint2*)&s_bits[wv][r][c0];
    const int yy = y0 + r, cc = blockIdx.x * 64 + c0;
    if (yy < h && cc < cpr) *(uint2*)((u8*)(bits + (size_t)b * bsw + (size_t)yy * bpw) + (size_t)cc * 2) = v;
    if (lane == 0 && wv == 0) {
        const u32 bm = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        // a stale read only costs a redundant atomic
        if (vmax && bm && (u32)__builtin_nontemporal_load(&vmax[b]) < bm) atomicMax(&vmax[b], (int32_t)bm);
    }
}

__device__ __forceinline__ u32 pb_min_bytes(u32 a, u32 b)
{
    typedef unsigned short us2p __attribute__((ext_vector_type(2)));
    const us2p ae = __builtin_bit_cast(us2p, a & 0x00ff00ffu), be = __builtin_bit_cast(us2p, b & 0x00ff00ffu);
    const us2p ao = __builtin_bit_cast(us2p, (a >> 8) & 0x00ff00ffu), bo = __builtin_bit_cast(us2p, (b >> 8) & 0x00ff00ffu);
    return __builtin_bit_cast(u32, __builtin_elementwise_min(ae, be)) | (__builtin_bit_cast(u32, __builtin_elementwise_min(ao, bo)) << 8);
}

// flags[b]: 0 = no non-zero pixel seen yet, v = every non-zero pixel so far equals v, -1 = two
// different non-zero values.  Each workgroup reduces the smallest and largest non-zero pixel
// of its 1024 x 16 pixels and touches the flag at most twice.
__global__ __launch_bounds__(256) void k_binary_check(Plane src, int w, int h, int cpr, int32_t* __restrict__ flags)
{
    __shared__ u32 s_mm[4][2];
    const int lane = threadIdx.x, wv = threadIdx.y;
    const int cx = blockIdx.x * 64 + lane;
    const int b = blockIdx.z;
    const int y0 = ((int)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(wv)) * ROWS_PER_LANE;
    const int n = w - cx * 16;
    u32 mx = 0, mn = 255;
#pragma unroll
    for (int i = 0; i < ROWS_PER_LANE; i++) {
        if (n <= 0 || y0 + i >= h) continue;
        const uint4 a = *(const uint4*)(src.row(b, y0 + i) + (size_t)cx * 16);
        if (n >= 16) {
            const u32 d[4] = {a.x, a.y, a.z, a.w};
            u32 hi = 0, lo = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                hi = pb_max_bytes(hi, d[q]);
                lo = pb_min_bytes(lo, d[q] | ((pb_nonzero_bytes(d[q]) ^ 0x01010101u) * 255u));  // zero bytes -> 0xff
            }
            mx = max(mx, max(max(hi & 0xffu, (hi >> 8) & 0xffu), max((hi >> 16) & 0xffu, hi >> 24)));
            mn = min(mn, min(min(lo & 0xffu, (lo >> 8) & 0xffu), min((lo >> 16) & 0xffu, lo >> 24)));
        } else {
            const u32 d[4] = {a.x, a.y, a.z, a.w};
            for (int k = 0; k < n; k++) {
                const u32 v = (d[k >> 2] >> (8 * (k & 3))) & 0xffu;
                if (v) {
                    mx = max(mx, v);
                    mn = min(mn, v);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        mx = max(mx, (u32)__shfl_xor((int)mx, o));
        mn = min(mn, (u32)__shfl_xor((int)mn, o));
    }
    if (lane == 0) {
        s_mm[wv][0] = mx;
        s_mm[wv][1] = mn;
    }
    __syncthreads();
    if (lane == 0 && wv == 0) {
        const u32 bmx = max(max(s_mm[0][0], s_mm[1][0]), max(s_mm[2][0], s_mm[3][0]));
        const u32 bmn = min(min(s_mm[0][1], s_mm[1][1]), min(s_mm[2][1], s_mm[3][1]));
        if (bmx) {
            if (bmn != bmx) {
                atomicExch(&flags[b], -1);
            } else {
                const int old = atomicCAS(&flags[b], 0, (int)bmx);
                if (old != 0 && old != (int)bmx) atomicExch(&flags[b], -1);
            }
        }
    }
}

struct BitMorphArgs {
    const u32* bits;
    size_t bpw, bsw;  // words per row / per image of the bit plane
    Plane dst;
    int w, h;
    int tww, tr;    // output tile: words x rows
    int hlw, hrw;   // halo words left / right
    int ht, hb;     // halo rows top / bottom
    int tiles_x, tiles_y;
    int first_is_dilate, nops;
    int kx, ky, ax, ay;
    int px, py;     // largest power of two <= kx / ky
    int value;
    const int32_t* value_dev;  // per-image value (overrides `value` when non-null)
};

// 32 bits of row `row` (RW words) starting at bit position `pos` (may lie outside).
__device__ __forceinline__ u32 rd_bits(const u32* row, int RW, int pos, u32 ident)
{
    int wi = pos >> 5;
    u32 s = (u32)pos & 31u;
    u32 lo = (wi >= 0 && wi < RW) ? row[wi] : ident;
    u32 hi = (wi + 1 >= 0 && wi + 1 < RW) ? row[wi + 1] : ident;
    return __builtin_amdgcn_alignbit(hi, lo, s);
}

__device__ __forceinline__ u32 comb(u32 a, u32 b, bool dil) { return dil ? (a | b) : (a & b); }

// One workgroup = one output tile of one image.  LDS: two ping-pong copies of
// the tile plus halo, RR rows x RW words each.
__global__ __launch_bounds__(256) void k_bitmorph(BitMorphArgs A)
{
    extern __shared__ __align__(16) u32 lds[];
    const int RW = A.tww + A.hlw + A.hrw;
    const int RR = A.tr + A.ht + A.hb;
    const int N = RW * RR;
    u32* cur = lds;
    u32* nxt = lds + N;
    const int tid = threadIdx.x;
    int tile = blockIdx.x;
    const int tx = tile % A.tiles_x;
    tile /= A.tiles_x;
    const int ty = tile % A.tiles_y;
    const int b = tile / A.tiles_y;
    const int gw0 = tx * A.tww - A.hlw;  // global word index of region word 0
    const int gy0 = ty * A.tr - A.ht;    // global row of region row 0
    const int words_in_row = (A.w + 31) >> 5;
    const u32* img = A.bits + (size_t)b * A.bsw;

    // in-image mask of region word (r, j)
    auto inside_mask = [&](int r, int j) -> u32 {
        int gy = gy0 + r;
        if (gy < 0 || gy >= A.h) return 0u;
        int gx = (gw0 + j) * 32;
        if (gx + 32 <= 0 || gx >= A.w) return 0u;
        if (gx < 0) return 0u;  // word-aligned: a word is never partly left of the image
        int rem = A.w - gx;
        return rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
    };

    bool dil = A.first_is_dilate != 0;
    u32 ident = dil ? 0u : 0xffffffffu;
    for (int e = tid; e < N; e += 256) {
        int r = e / RW, j = e - r * RW;
        int gy = gy0 + r, gj = gw0 + j;
        u32 v = 0;
        if (gy >= 0 && gy < A.h && gj >= 0 && gj < words_in_row) v = img[(size_t)gy * A.bpw + gj];
        u32 m = inside_mask(r, j);
        cur[e] = (v & m) | (ident & ~m);
    }
    __syncthreads();

    for (int stage = 0; stage < A.nops; stage++) {
        if (stage > 0) {
            dil = !dil;
            ident = dil ? 0u : 0xffffffffu;
            // samples outside the image take the identity of the next operation
            for (int e = tid; e < N; e += 256) {
                int r = e / RW, j = e - r * RW;
                u32 m = inside_mask(r, j);
                cur[e] = (cur[e] & m) | (ident & ~m);
            }
            __syncthreads();
        }
        if (A.kx > 1) {
            for (int m = 1; m < A.px; m <<= 1) {
                for (int e = tid; e < N; e += 256) {
                    int r = e / RW, j = e - r * RW;
                    const u32* row = cur + r * RW;
                    nxt[e] = comb(row[j], rd_bits(row, RW, 32 * j + m, ident), dil);
                }
                __syncthreads();
                u32* t = cur; cur = nxt; nxt = t;
            }
            for (int e = tid; e < N; e += 256) {
                int r = e / RW, j = e - r * RW;
                const u32* row = cur + r * RW;
                int p0 = 32 * j - A.ax;
                nxt[e] = comb(rd_bits(row, RW, p0, ident), rd_bits(row, RW, p0 + A.kx - A.px, ident), dil);
            }
            __syncthreads();
            u32* t = cur; cur = nxt; nxt = t;
        }
        if (A.ky > 1) {
            for (int m = 1; m < A.py; m <<= 1) {
                for (int e = tid; e < N; e += 256) {
                    int r = e / RW;
                    u32 o = (r + m < RR) ? cur[e + m * RW] : ident;
                    nxt[e] = comb(cur[e], o, dil);
                }
                __syncthreads();
                u32* t = cur; cur = nxt; nxt = t;
            }
            for (int e = tid; e < N; e += 256) {
                int r = e / RW, j = e - r * RW;
                int r0 = r - A.ay, r1 = r0 + A.ky - A.py;
                u32 a = (r0 >= 0 && r0 < RR) ? cur[r0 * RW + j] : ident;
                u32 c = (r1 >= 0 && r1 < RR) ? cur[r1 * RW + j] : ident;
                nxt[e] = comb(a, c, dil);
            }
            __syncthreads();
            u32* t = cur; cur = nxt; nxt = t;
        }
    }

    // expand the tile interior to bytes: one lane = 16 pixels = one 16-byte store
    const u32 value = A.value_dev ? (u32)A.value_dev[b] : (u32)A.value;
    const int halves = A.tww * 2;
    for (int e = tid; e < A.tr * halves; e += 256) {
        int r = e / halves, hx = e - r * halves;
        int gy = ty * A.tr + r;
        int gx = (tx * A.tww) * 32 + hx * 16;
        if (gy >= A.h || gx >= A.w) continue;
        u32 wv = cur[(r + A.ht) * RW + A.hlw + (hx >> 1)];
        u32 hbits = (wv >> ((hx & 1) * 16)) & 0xffffu;
        u32 o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 nib = (hbits >> (4 * q)) & 0xfu;
            o[q] = ((nib * 0x00204081u) & 0x01010101u) * value;
        }
        *(uint4*)(A.dst.row(b, gy) + gx) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- register-level bit-plane kernels (the production shapes) ---------------------

__device__ __forceinline__ void expand16(u32 hbits, u32 value, u8* out)
{
    u32 o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u32 nib = (hbits >> (4 * q)) & 0xfu;
        o[q] = ((nib * 0x00204081u) & 0x01010101u) * value;
    }
    *(uint4*)out = make_uint4(o[0], o[1], o[2], o[3]);
}

// Bits -> bytes through a 256-entry table in LDS (8 result bytes per byte of bits): two table
// reads and one 16-byte store per 16 pixels, instead of shifts, masks and 32-bit multiplies.
// The table holds 0x00 / 0xff bytes; `vmask` (the output value in all four bytes) is applied
// after the lookup.  Built by the first 256 threads of the workgroup (which must have at least
// that many); ends with a barrier.
__device__ __forceinline__ void build_expand_lut(uint2* lut)
{
    const u32 i = threadIdx.y * blockDim.x + threadIdx.x;
    if (i < 256) {
        const u32 lo = (((i & 15u) * 0x00204081u) & 0x01010101u) * 255u;
        const u32 hi = (((i >> 4) * 0x00204081u) & 0x01010101u) * 255u;
        lut[i] = make_uint2(lo, hi);
    }
    __syncthreads();
}
__device__ __forceinline__ void expand16_lut(const uint2* lut, u32 hbits, u32 vmask, u8* out)
{
    const uint2 e0 = lut[hbits & 0xffu], e1 = lut[(hbits >> 8) & 0xffu];
    *(uint4*)out = make_uint4(e0.x & vmask, e0.y & vmask, e1.x & vmask, e1.y & vmask);
}

struct Win3 {
    u32 a0, a1, a2;  // 96 consecutive pixels: a0 = lowest x
};

// window seen `d` pixels to the right (d in 1..31): out[p] = in[p + d]; `fill` enters at the top
__device__ __forceinline__ Win3 look_right(Win3 v, int d, u32 fill)
{
    Win3 r;
    r.a0 = __builtin_amdgcn_alignbit(v.a1, v.a0, d);
    r.a1 = __builtin_amdgcn_alignbit(v.a2, v.a1, d);
    r.a2 = __builtin_amdgcn_alignbit(fill, v.a2, d);
    return r;
}
// window seen `d` pixels to the left (d in 1..31): out[p] = in[p - d]
__device__ __forceinline__ Win3 look_left(Win3 v, int d, u32 fill)
{
    Win3 r;
    r.a2 = __builtin_amdgcn_alignbit(v.a2, v.a1, 32 - d);
    r.a1 = __builtin_amdgcn_alignbit(v.a1, v.a0, 32 - d);
    r.a0 = __builtin_amdgcn_alignbit(v.a0, fill, 32 - d);
    return r;
}
// out[p] = in[p + d] for d in [-32, 32]
__device__ __forceinline__ Win3 look(Win3 v, int d, u32 fill)
{
    if (d == 0) return v;
    if (d == 32) { Win3 r; r.a0 = v.a1; r.a1 = v.a2; r.a2 = fill; return r; }
    if (d == -32) { Win3 r; r.a0 = fill; r.a1 = v.a0; r.a2 = v.a1; return r; }
    return d > 0 ? look_right(v, d, fill) : look_left(v, -d, fill);
}
__device__ __forceinline__ Win3 comb3(Win3 a, Win3 b, bool dil)
{
    Win3 r;
    r.a0 = comb(a.a0, b.a0, dil); r.a1 = comb(a.a1, b.a1, dil); r.a2 = comb(a.a2, b.a2, dil);
    return r;
}

// Horizontal erode / dilate / open / close of a bit plane whose whole reach fits in the
// neighbouring words (nops*ax <= 32 and nops*(kx-1-ax) <= 32: e.g. the 29-pixel line
// kernel of the parsers).  One lane = one 32-pixel word of 4 consecutive rows; it holds
// words j-1, j, j+1 of a row in registers, runs the doubling steps there and stores the
// 32 result bytes.
// NOPS / KX / AX are compile-time copies of nops / kx / ax for the common shapes (0 = use the
// run-time arguments): every shift becomes an immediate and the loops unroll.
template <int NOPS, int KX, int AX>
__global__ __launch_bounds__(256) void k_bitmorph_h3(const u32* __restrict__ bits, size_t bpw, size_t bsw, Plane dst,
                                                     int w, int h, int first_is_dilate, int nops_rt, int kx_rt, int ax_rt,
                                                     int px_rt, u32 value, const int32_t* __restrict__ value_dev,
                                                     u32* __restrict__ bits_out)
{
    __shared__ uint2 lut[256];
    build_expand_lut(lut);
    const int nops = NOPS ? NOPS : nops_rt, kx = NOPS ? KX : kx_rt, ax = NOPS ? AX : ax_rt;
    int px = px_rt;
    if (NOPS) {
        px = 1;
        while (px * 2 <= KX) px *= 2;
    }
    const int words = (w + 31) >> 5;
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int b = blockIdx.z;
    const int y0 = (blockIdx.y * 4 + threadIdx.y) * ROWS_PER_LANE;
    const bool live = j < words;  // idle lanes stay: the stores below are rearranged across the wave
    // in-image masks of the three words
    Win3 m;
    m.a0 = j >= 1 ? 0xffffffffu : 0u;
    int rem = w - 32 * j;
    m.a1 = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
    m.a2 = rem - 32 >= 32 ? 0xffffffffu : (rem - 32 > 0 ? ((1u << (rem - 32)) - 1u) : 0u);
    const u32 vmask = (value_dev ? (u32)value_dev[b] : value) * 0x01010101u;
    Win3 in[ROWS_PER_LANE];
#pragma unroll
    for (int i = 0; i < ROWS_PER_LANE; i++) {
        const int y = y0 + i;
        in[i].a0 = in[i].a1 = in[i].a2 = 0;
        if (y < h && live) {
            const u32* row = bits + (size_t)b * bsw + (size_t)y * bpw;
            in[i].a0 = j >= 1 ? row[j - 1] : 0u;
            in[i].a1 = row[j];
            in[i].a2 = j + 1 < words ? row[j + 1] : 0u;
        }
    }
#pragma unroll
    for (int i = 0; i < ROWS_PER_LANE; i++) {
        const int y = y0 + i;
        if (y >= h) break;  // uniform over the wave (one row per wave and i)
        Win3 v = in[i];
        bool dil = first_is_dilate != 0;
#pragma unroll
        for (int stage = 0; stage < (NOPS ? NOPS : 2); stage++) {
            if (stage >= nops) break;
            const u32 ident = dil ? 0u : 0xffffffffu;
            // pixels outside the image take the identity of this stage
            v.a0 = (v.a0 & m.a0) | (ident & ~m.a0);
            v.a1 = (v.a1 & m.a1) | (ident & ~m.a1);
            v.a2 = (v.a2 & m.a2) | (ident & ~m.a2);
#pragma unroll
            for (int mm = 1; mm <= 32; mm <<= 1)  // px = 64 (kx = 64, 65) needs the whole-word step too
                if (mm < px) v = comb3(v, look(v, mm, ident), dil);
            v = comb3(look(v, -ax, ident), look(v, kx - px - ax, ident), dil);
            dil = !dil;
        }
        // The wave's 64 words are 2 KiB of output.  Lane L stores 16-byte piece L of the first
        // KiB and piece L of the second, so that each store instruction writes 1 KiB without
        // gaps (32 bytes per lane would leave every instruction half of each cache line).
        const int lane = threadIdx.x;
        const u32 r1 = v.a1 & m.a1;
        if (bits_out && live) bits_out[(size_t)b * bsw + (size_t)y * bpw + j] = r1;  // the result as a bit plane too
        const u32 wa = (u32)__shfl((int)r1, lane >> 1), wb = (u32)__shfl((int)r1, 32 + (lane >> 1));
        const int sh = (lane & 1) * 16;
        const int xa = blockIdx.x * 2048 + lane * 16, xb = xa + 1024;  // first pixel of each piece
        u8* orow = dst.row(b, y);
        if (xa < w) expand16_lut(lut, (wa >> sh) & 0xffffu, vmask, orow + xa);
        if (xb < w) expand16_lut(lut, (wb >> sh) & 0xffffu, vmask, orow + xb);
    }
}

// Bit-sliced run counter: `d` (n slices) = how many more consecutive 1-rows are needed for a
// full window of k; one update per row.  Returns the mask of columns whose window is full.
template <int NS>
__device__ __forceinline__ u32 run_update(u32 (&d)[NS], u32 in, int k)
{
    // decrement where non-zero
    u32 nz = d[0];
#pragma unroll
    for (int i = 1; i < NS; i++) nz |= d[i];
    u32 borrow = nz;
#pragma unroll
    for (int i = 0; i < NS; i++) {
        const u32 di = d[i];
        d[i] = di ^ borrow;
        borrow &= ~di;
    }
    // reload k where the input row is 0
    const u32 nin = ~in;
    u32 any = 0;
#pragma unroll
    for (int i = 0; i < NS; i++) {
        d[i] = ((k >> i) & 1) ? (d[i] | nin) : (d[i] & in);
        any |= d[i];
    }
    return ~any;
}

// Vertical erode / dilate / open / close of a bit plane (kx == 1, any ky < 2^NS).  One lane
// streams one 32-pixel word column down a strip of rows; the k-row window is a bit-sliced
// run counter per stage, so the cost per row does not depend on k and nothing but the
// counters is kept.  A dilation is the complement of an erosion of the complement.
template <int NS>
__global__ __launch_bounds__(256) void k_bitmorph_v(const u32* __restrict__ bits, size_t bpw, size_t bsw, Plane dst,
                                                    int w, int h, int words, int strips, int strip_rows,
                                                    int first_is_dilate, int nops, int ky, int ay,
                                                    u32 value, const int32_t* __restrict__ value_dev,
                                                    u32* __restrict__ bits_out)
{
    __shared__ uint2 lut[256];
    build_expand_lut(lut);
    // wave = 64 consecutive word columns of one strip of one image
    const int lane = threadIdx.x;
    const int j = blockIdx.x * 64 + lane;
    const int sidx = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (sidx >= strips) return;
    const bool live = j < words;  // idle lanes stay: the stores are rearranged across the wave
    const int below = ky - 1 - ay;  // rows a window reaches below its output row
    const int r0 = sidx * strip_rows, r1 = min(r0 + strip_rows, h);
    const u32* col = bits + (size_t)b * bsw + j;
    const int rem = w - 32 * j;
    const u32 wmask = rem >= 32 ? 0xffffffffu : (rem > 0 ? ((1u << rem) - 1u) : 0u);
    const u32 vmask = (value_dev ? (u32)value_dev[b] : value) * 0x01010101u;
    const bool dil0 = first_is_dilate != 0;
    // Lane L stores 16-byte piece L of the first KiB of the wave's 2 KiB row segment and piece L
    // of the second: every store instruction then writes 1 KiB without gaps.
    const int sh = (lane & 1) * 16;
    const int xa = blockIdx.x * 2048 + lane * 16, xb = xa + 1024;
    u32 d1[NS], d2[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        d1[i] = ((ky >> i) & 1) ? 0xffffffffu : 0u;  // "no row seen yet"
        d2[i] = d1[i];
    }
    // stage-1 value for row c1 is known once input row c1 + below has been read; stage-2
    // (if any) for row r once stage-1 row r + below is known.
    const int lag1 = below, lag2 = nops == 2 ? below : 0;
    const int y_first = r0 - nops * ay, y_last = r1 - 1 + lag1 + lag2;
    auto step = [&](int y, u32 raw) {  // y is the same in all lanes
        // stage 1 input: image row, or the stage's identity outside the image
        // (erode: ones; dilate, run as erosion of the complement: also ones)
        u32 in1 = 0xffffffffu;
        if (y >= 0 && y < h) in1 = ((dil0 ? ~raw : raw) & wmask) | ~wmask;
        u32 full1 = run_update<NS>(d1, in1, ky);
        const int c1 = y - lag1;
        if (c1 < r0 - (nops == 2 ? ay : 0)) return;
        u32 s1 = dil0 ? ~full1 : full1;  // stage-1 result for row c1
        u32 out;
        int r;
        if (nops == 1) {
            out = s1;
            r = c1;
        } else {
            // stage 2 runs the opposite operation on the stage-1 rows; outside the image
            // its input is its own identity
            const bool dil1 = !dil0;
            u32 in2 = 0xffffffffu;
            if (c1 >= 0 && c1 < h) in2 = ((dil1 ? ~s1 : s1) & wmask) | ~wmask;
            u32 full2 = run_update<NS>(d2, in2, ky);
            out = dil1 ? ~full2 : full2;
            r = c1 - lag2;
        }
        if (r < r0 || r >= r1) return;
        out &= wmask;
        if (bits_out && live) bits_out[(size_t)b * bsw + (size_t)r * bpw + j] = out;  // the result as a bit plane too
        const u32 wa = (u32)__shfl((int)out, lane >> 1), wb = (u32)__shfl((int)out, 32 + (lane >> 1));
        u8* orow = dst.row(b, r);
        if (xa < w) expand16_lut(lut, (wa >> sh) & 0xffffu, vmask, orow + xa);
        if (xb < w) expand16_lut(lut, (wb >> sh) & 0xffffu, vmask, orow + xb);
    };
    // four rows per trip: their loads are all in flight before the first is consumed
    for (int y = y_first; y <= y_last; y += 4) {
        u32 raw[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int yy = y + k;
            raw[k] = (live && yy >= 0 && yy < h && yy <= y_last) ? col[(size_t)yy * bpw] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (y + k <= y_last) step(y + k, raw[k]);
    }
}


// ---- two-valued images: bytes in, bytes out, any rectangle up to 33 x 31, one pass, no bit plane in memory -------
// The layout of k_bitmorph_v (a wave = 64 consecutive 32-pixel words of a strip of rows, walking down) fed from the
// bytes: lane L reads 16-byte piece L of each of the wave's two KiB of a row (every load instruction covers a
// whole KiB), turns its 32 bytes into 32 "non-zero" bits, and two lane moves hand every lane the word of its own
// 32 pixels.  A stage erodes along the row on the words of the lane and its two neighbours (doubling, as in
// k_bitmorph_h3; dilation = erosion of the complement, outside the image = ones) and along the column with the
// bit-sliced run counter of k_bitmorph_v; an open / close runs two stages, the second on the first one's words.
// Lanes 0 .. nops-1 and 64-nops .. 63 are halo.  The output value is the non-zero value of the image, picked up
// from the first non-zero dword the wave meets (an output pixel can only be set when an input pixel of its window
// was, and that one the wave has read).
// CHECK: the image is not known to be two-valued.  Every byte is compared with {0, value}; the wave that meets a
// third value sets flags[image] and stops, the others look at the flag every four rows; the caller runs the
// gray-level kernel for the flagged images afterwards.
__device__ __forceinline__ u32 lane_up1(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xf, 0xf, true); }    // from lane + 1
__device__ __forceinline__ u32 lane_down1(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true); }  // from lane - 1
__device__ __forceinline__ u32 bm_pack16(uint4 v)
{
    return pb_bytes_to_bits(pb_nonzero_bytes(v.x)) | (pb_bytes_to_bits(pb_nonzero_bytes(v.y)) << 4) |
           (pb_bytes_to_bits(pb_nonzero_bytes(v.z)) << 8) | (pb_bytes_to_bits(pb_nonzero_bytes(v.w)) << 12);
}
// erosion along the row: out[p] = AND of in[p - ax .. p - ax + kx - 1] for the word x between L and R
__device__ __forceinline__ u32 bm_erode_row(u32 L, u32 x, u32 R, int kx, int ax, int px)
{
    if (kx == 1) return x;
    Win3 v;
    v.a0 = L; v.a1 = x; v.a2 = R;
#pragma unroll
    for (int mm = 1; mm <= 32; mm <<= 1)  // px = 64 (kx = 64, 65) needs the whole-word step too
        if (mm < px) v = comb3(v, look(v, mm, 0xffffffffu), false);
    return look(v, -ax, 0xffffffffu).a1 & look(v, kx - px - ax, 0xffffffffu).a1;
}

template <int NS, bool CHECK>
__global__ __launch_bounds__(256) void k_binmorph2d(Plane src, Plane dst, int w, int h, int strips_y, int strip_rows, int useful,
                                                    int first_is_dilate, int nops, int kx, int ax, int px, int ky, int ay,
                                                    int32_t* __restrict__ flags)
{
    __shared__ uint2 lut[256];
    build_expand_lut(lut);
    const int lane = threadIdx.x;
    const int sidx = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (sidx >= strips_y) return;
    const int j0 = (int)blockIdx.x * useful - nops;  // word of lane 0
    const int j = j0 + lane;
    const int rem = w - 32 * j;
    const u32 wmask = (j < 0 || rem <= 0) ? 0u : (rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u));
    const int xa = 32 * j0 + 16 * lane, xb = xa + 1024;  // first pixel of the lane's two pieces
    const bool la = xa >= 0 && xa < w, lb = xb >= 0 && xb < w;
    const int na = w - xa, nb = w - xb;  // pixels of the piece inside the image when it is the last one
    const int below = ky - 1 - ay;
    const int r0 = sidx * strip_rows, r1 = min(r0 + strip_rows, h);
    const bool dil0 = first_is_dilate != 0;
    const u8* const sbase = src.p + (size_t)b * src.stride;
    // which halves of which lanes make up this lane's word, and which words this lane stores
    const int src_lane = (2 * lane) & 63;
    const bool hi_half = lane >= 32;
    const int sh = (lane & 1) * 16;
    const int wl_a = lane >> 1, wl_b = 32 + (lane >> 1);  // lanes whose words hold pieces L and 64 + L
    const bool st_a = wl_a >= nops && wl_a < 64 - nops && xa < w, st_b = wl_b >= nops && wl_b < 64 - nops && xb < w;
    u32 d1[NS], d2[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        d1[i] = ((ky >> i) & 1) ? 0xffffffffu : 0u;  // "no row seen yet"
        d2[i] = d1[i];
    }
    u32 vmask = 0;  // the image's non-zero value in all four bytes, 0 = not met yet (wave-uniform)
    const int lag1 = below, lag2 = nops == 2 ? below : 0;
    const int y_first = r0 - nops * ay, y_last = r1 - 1 + lag1 + lag2;
    bool stop = false;  // (wave-uniform)
    auto bytes_of = [&](uint4 v, int n) -> uint4 {  // the piece with the bytes beyond the image width cleared
        if (n >= 16) return v;
        u32 d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int m = n - 4 * q;
            d[q] = m >= 4 ? d[q] : (m <= 0 ? 0u : (d[q] & ((1u << (8 * m)) - 1u)));
        }
        return make_uint4(d[0], d[1], d[2], d[3]);
    };
    // looks at one row's pieces: picks up the image's value, and (CHECK) true when a byte is neither 0 nor the value
    auto examine = [&](uint4& pa, uint4& pb) -> bool {
        pa = bytes_of(pa, na);
        pb = bytes_of(pb, nb);
        const u32 any = pa.x | pa.y | pa.z | pa.w | pb.x | pb.y | pb.z | pb.w;
        if (vmask == 0) {
            const unsigned long long nz = __ballot(any != 0);
            if (nz) {
                u32 f = any | (any >> 16);
                f = (f | (f >> 8)) & 0xffu;
                vmask = (u32)__builtin_amdgcn_readlane((int)f, (int)__builtin_ctzll(nz)) * 0x01010101u;
            }
        }
        if (!CHECK) return false;
        // every byte must be 0 or the value: d == (0xff in its non-zero bytes) & vmask
        u32 bad = 0;
        const u32 dd[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const u32 nzb = pb_nonzero_bytes(dd[q]);
            bad |= dd[q] ^ (((nzb << 8) - nzb) & vmask);
        }
        return __any(bad != 0);
    };
    if (CHECK) {
        // Four rows spread over the strip first: an image with more than two values is usually found out here, before
        // any of the real work (a thresholded plane pays four extra rows per strip).
        uint4 qa[4], qb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int yy = min(r0 + (2 * k + 1) * strip_rows / 8, h - 1);
            qa[k] = qb[k] = make_uint4(0, 0, 0, 0);
            if (la) qa[k] = *(const uint4*)(sbase + (size_t)yy * src.pitch + xa);
            if (lb) qb[k] = *(const uint4*)(sbase + (size_t)yy * src.pitch + xb);
        }
        bool found = false;
#pragma unroll
        for (int k = 0; k < 4; k++) found = found || examine(qa[k], qb[k]);
        if (found) {
            if (lane == 0) atomicOr(&flags[b], 1);
            return;
        }
        if (__hip_atomic_load(&flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
    auto step = [&](int y, uint4 pa, uint4 pb) {  // y is the same in all lanes
        u32 in1 = 0xffffffffu;
        if (y >= 0 && y < h) {
            if (examine(pa, pb)) {
                if (lane == 0) atomicOr(&flags[b], 1);
                stop = true;
                return;
            }
            const u32 hw = bm_pack16(pa) | (bm_pack16(pb) << 16);
            const u32 v0 = (u32)__shfl((int)hw, src_lane), v1 = (u32)__shfl((int)hw, src_lane + 1);
            const u32 raw = hi_half ? ((v0 >> 16) | (v1 & 0xffff0000u)) : ((v0 & 0xffffu) | (v1 << 16));
            const u32 x = ((dil0 ? ~raw : raw) & wmask) | ~wmask;
            in1 = bm_erode_row(lane_down1(x), x, lane_up1(x), kx, ax, px);
        }
        const u32 full1 = ky == 1 ? in1 : run_update<NS>(d1, in1, ky);
        const int c1 = y - lag1;
        if (c1 < r0 - (nops == 2 ? ay : 0)) return;
        const u32 s1 = dil0 ? ~full1 : full1;  // stage-1 result of row c1
        u32 out;
        int r;
        if (nops == 1) {
            out = s1;
            r = c1;
        } else {
            const bool dil1 = !dil0;
            u32 in2 = 0xffffffffu;
            if (c1 >= 0 && c1 < h) {
                const u32 x = ((dil1 ? ~s1 : s1) & wmask) | ~wmask;
                in2 = bm_erode_row(lane_down1(x), x, lane_up1(x), kx, ax, px);
            }
            const u32 full2 = ky == 1 ? in2 : run_update<NS>(d2, in2, ky);
            out = dil1 ? ~full2 : full2;
            r = c1 - lag2;
        }
        if (r < r0 || r >= r1) return;
        out &= wmask;
        const u32 wa = (u32)__shfl((int)out, wl_a), wb = (u32)__shfl((int)out, wl_b);
        u8* orow = dst.row(b, r);
        if (st_a) expand16_lut(lut, (wa >> sh) & 0xffffu, vmask, orow + xa);
        if (st_b) expand16_lut(lut, (wb >> sh) & 0xffffu, vmask, orow + xb);
    };
    // four rows per trip: their loads are all in flight before the first is consumed
    for (int y = y_first; y <= y_last && !stop; y += 4) {
        uint4 pa[4], pb[4];
        int flag = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int yy = y + k;
            const bool in_rows = yy >= 0 && yy < h && yy <= y_last;
            pa[k] = pb[k] = make_uint4(0, 0, 0, 0);
            if (la && in_rows) pa[k] = *(const uint4*)(sbase + (size_t)yy * src.pitch + xa);
            if (lb && in_rows) pb[k] = *(const uint4*)(sbase + (size_t)yy * src.pitch + xb);
        }
        if (CHECK) flag = __hip_atomic_load(&flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (y + k <= y_last && !stop) step(y + k, pa[k], pb[k]);
        if (CHECK && flag) stop = true;
    }
}

// ------------------------------------------------------------------ general 8-bit

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// per-byte min / max of two packed dwords through 16-bit packed ops
__device__ __forceinline__ u32 minmax4(u32 a, u32 b, bool dil)
{
    u32 ae = a & 0x00ff00ffu, ao = (a >> 8) & 0x00ff00ffu;
    u32 be = b & 0x00ff00ffu, bo = (b >> 8) & 0x00ff00ffu;
    us2 xe = __builtin_bit_cast(us2, ae), ye = __builtin_bit_cast(us2, be);
    us2 xo = __builtin_bit_cast(us2, ao), yo = __builtin_bit_cast(us2, bo);
    us2 re = dil ? __builtin_elementwise_max(xe, ye) : __builtin_elementwise_min(xe, ye);
    us2 ro = dil ? __builtin_elementwise_max(xo, yo) : __builtin_elementwise_min(xo, yo);
    return __builtin_bit_cast(u32, re) | (__builtin_bit_cast(u32, ro) << 8);
}

// Row pass: a workgroup owns 1024 consecutive pixels of one row; the segment and
// its halo sit in LDS as bytes; each lane produces 4 adjacent outputs.
__global__ __launch_bounds__(256) void k_morph_row(Plane src, Plane dst, int w, int h, int segs, int k, int a,
                                                   int dil, const int32_t* __restrict__ gate)
{
    extern __shared__ __align__(16) u8 srow[];
    int blk = blockIdx.x;
    const int sx = blk % segs;
    blk /= segs;
    const int y = blk % h, b = blk / h;
    if (gate && gate[b] == 0) return;  // (gate: only the images flagged by the two-valued kernel)
    const int x0 = sx * 1024;
    const int L = (a + 3) & ~3, R = (k - 1 - a + 3) & ~3;
    const int n = 1024 + L + R;  // bytes in LDS, multiple of 4
    const u8 ident = dil ? 0 : 255;
    const u8* srcrow = src.row(b, y);
    for (int i = threadIdx.x; i < n / 4; i += 256) {
        int gx = x0 - L + 4 * i;
        u32 v;
        if (gx >= 0 && gx + 4 <= w) {
            v = *(const u32*)(srcrow + gx);
        } else {
            v = 0;
            for (int j = 0; j < 4; j++) {
                int x = gx + j;
                u32 p = (x >= 0 && x < w) ? srcrow[x] : ident;
                v |= p << (8 * j);
            }
        }
        ((u32*)srow)[i] = v;
    }
    __syncthreads();
    const int ox = x0 + 4 * threadIdx.x;
    if (ox >= w) return;
    const int base = L + 4 * threadIdx.x - a;
    int o0 = ident, o1 = ident, o2 = ident, o3 = ident;
    for (int i = 0; i < k + 3; i++) {
        int v = srow[base + i];
        if (dil) {
            if (i < k) o0 = max(o0, v);
            if (i >= 1 && i < k + 1) o1 = max(o1, v);
            if (i >= 2 && i < k + 2) o2 = max(o2, v);
            if (i >= 3) o3 = max(o3, v);
        } else {
            if (i < k) o0 = min(o0, v);
            if (i >= 1 && i < k + 1) o1 = min(o1, v);
            if (i >= 2 && i < k + 2) o2 = min(o2, v);
            if (i >= 3) o3 = min(o3, v);
        }
    }
    *(u32*)(dst.row(b, y) + ox) = (u32)o0 | ((u32)o1 << 8) | ((u32)o2 << 16) | ((u32)o3 << 24);
}

// Column pass: one lane = 4 adjacent columns (one dword) x 4 output rows.
__global__ __launch_bounds__(256) void k_morph_col(Plane src, Plane dst, int w, int h, int dwords, int rowgroups, int k,
                                                   int a, int dil, long long total, const int32_t* __restrict__ gate)
{
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += step) {
        int dx = (int)(c % dwords);
        long long t = c / dwords;
        int rg = (int)(t % rowgroups), b = (int)(t / rowgroups);
        if (gate && gate[b] == 0) continue;
        const int y0 = rg * 4;
        const u32 ident = dil ? 0u : 0xffffffffu;
        u32 o[4] = {ident, ident, ident, ident};
        for (int i = 0; i < k + 3; i++) {
            int yy = y0 - a + i;
            if (yy < 0 || yy >= h) continue;
            u32 v = *(const u32*)(src.row(b, yy) + 4 * dx);
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (i >= p && i < k + p) o[p] = minmax4(o[p], v, dil != 0);
        }
#pragma unroll
        for (int p = 0; p < 4; p++)
            if (y0 + p < h) *(u32*)(dst.row(b, y0 + p) + 4 * dx) = o[p];
    }
}


// ---- general 8-bit, O(1) per pixel in the kernel length ---------------------------------------
//
// Rows: a wave holds 1024 consecutive columns of TWO adjacent image rows in registers, 16 columns
// per lane, one dword per column (low half = upper row, high half = lower row), so that one packed
// 16-bit min handles both rows.  A k-long window minimum is built by doubling: M_2s[x] =
// min(M_s[x], M_s[x +- s]) for s = 1, 2, 4, .. up to the largest power of two p <= k, then one more
// min with a copy displaced by k - p.  Every displacement is a power of two (k - p is applied bit by
// bit), so a displacement below 16 is a renaming of the lane's registers plus one lane-to-lane move
// for the columns that cross, and a displacement of 16 q is a move from lane +- q.  The direction of
// each round is chosen so that the backward reach adds up to the anchor: no final realignment.
// A maximum is the minimum of the complemented bytes.  Both stages of an open / close run on the
// registers; columns outside the image are set to the identity before each stage.
typedef unsigned short mus2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pk_min16(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(mus2, a), __builtin_bit_cast(mus2, b)));
}

constexpr int RD_COLS = 16;            // columns per lane
constexpr int RD_WAVE = 64 * RD_COLS;  // columns per wave

// o[i] = m at column (own column i) + DIR * 2^J; columns beyond the wave's ends hold junk
template <int J, int DIR>
__device__ __forceinline__ void rd_shift(const u32 (&m)[RD_COLS], u32 (&o)[RD_COLS])
{
    if (J < 4) {
        constexpr int r = 1 << (J < 4 ? J : 0);
#pragma unroll
        for (int i = 0; i < RD_COLS; i++) {
            if (DIR > 0) o[i] = i + r < RD_COLS ? m[(i + r) % RD_COLS] : lane_up1(m[(i + r) % RD_COLS]);
            else o[i] = i - r >= 0 ? m[(i - r + RD_COLS) % RD_COLS] : lane_down1(m[(i - r + RD_COLS) % RD_COLS]);
        }
    } else {
        constexpr int q = 1 << (J >= 4 ? J - 4 : 0);
#pragma unroll
        for (int i = 0; i < RD_COLS; i++) {
            if (q == 1) o[i] = DIR > 0 ? lane_up1(m[i]) : lane_down1(m[i]);
            else o[i] = DIR > 0 ? (u32)__shfl_down((int)m[i], q) : (u32)__shfl_up((int)m[i], q);
        }
    }
}

template <int J>
__device__ __forceinline__ void rd_shift_dir(const u32 (&m)[RD_COLS], u32 (&o)[RD_COLS], bool backward)
{
    if (backward) rd_shift<J, -1>(m, o);
    else rd_shift<J, 1>(m, o);
}

// in place: m[x] = min over columns [x - a, x - a + k - 1] (16-bit halves independently); k, a wave-uniform
__device__ __forceinline__ void rd_window_min(u32 (&m)[RD_COLS], int k, int a)
{
    int p = 1;
    while (p * 2 <= k) p *= 2;
    const int t = k - p;
    const bool final_back = a > p - 1;
    const int abits = final_back ? a - t : a;
    u32 o[RD_COLS];
#define RD_ROUND(J)                                               \
    if ((1 << J) < p) {                                           \
        rd_shift_dir<J>(m, o, ((abits >> J) & 1) != 0);           \
        _Pragma("unroll") for (int i = 0; i < RD_COLS; i++) m[i] = pk_min16(m[i], o[i]); \
    }
    RD_ROUND(0) RD_ROUND(1) RD_ROUND(2) RD_ROUND(3) RD_ROUND(4) RD_ROUND(5) RD_ROUND(6) RD_ROUND(7) RD_ROUND(8)
#undef RD_ROUND
    if (t > 0) {
        u32 c[RD_COLS];
#pragma unroll
        for (int i = 0; i < RD_COLS; i++) c[i] = m[i];
#define RD_MOVE(J)                                                \
    if ((t >> J) & 1) {                                           \
        rd_shift_dir<J>(c, o, final_back);                        \
        _Pragma("unroll") for (int i = 0; i < RD_COLS; i++) c[i] = o[i]; \
    }
        RD_MOVE(0) RD_MOVE(1) RD_MOVE(2) RD_MOVE(3) RD_MOVE(4) RD_MOVE(5) RD_MOVE(6) RD_MOVE(7) RD_MOVE(8)
#undef RD_MOVE
#pragma unroll
        for (int i = 0; i < RD_COLS; i++) m[i] = pk_min16(m[i], c[i]);
    }
}

// One wave = one pair of rows x one strip of `useful` output columns (+ halo).  nops = 1 (erode or
// dilate) or 2 (open / close, second stage of the other kind).  halo_l is a multiple of 16.
__global__ __launch_bounds__(256) void k_morph_row_dbl(Plane src, Plane dst, int w, int h, int strips, int pairs, int useful,
                                                       int halo_l, int nops, int first_dil, int k, int a, long long nwaves,
                                                       const int32_t* __restrict__ gate)
{
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave_id >= nwaves) return;
    const int lane = threadIdx.x & 63;
    const u32 wid = (u32)wave_id;
    const u32 t1 = wid / (u32)strips;
    const int sx = (int)(wid - t1 * (u32)strips);
    const int b = (int)(t1 / (u32)pairs), rp = (int)(t1 - (u32)b * (u32)pairs);
    if (gate && gate[b] == 0) return;
    const int y = 2 * rp;
    const int x0 = sx * useful - halo_l + RD_COLS * lane;  // first column of this lane (a multiple of 16)
    const bool in_row = x0 >= 0 && x0 < w;
    uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
    if (in_row) {
        ra = *(const uint4*)(src.row(b, y) + x0);
        if (y + 1 < h) rb = *(const uint4*)(src.row(b, y + 1) + x0);
    }
    u32 m[RD_COLS];
    {
        const u32 A[4] = {ra.x, ra.y, ra.z, ra.w}, B[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            m[4 * j + 0] = __builtin_amdgcn_perm(B[j], A[j], 0x0c040c00u);
            m[4 * j + 1] = __builtin_amdgcn_perm(B[j], A[j], 0x0c050c01u);
            m[4 * j + 2] = __builtin_amdgcn_perm(B[j], A[j], 0x0c060c02u);
            m[4 * j + 3] = __builtin_amdgcn_perm(B[j], A[j], 0x0c070c03u);
        }
    }
    // columns of this lane inside the image (bit i = column x0 + i)
    u32 inside = 0;
    if (x0 + RD_COLS > 0 && x0 < w) {
        const int lo = x0 < 0 ? -x0 : 0, hi = w - x0 < RD_COLS ? w - x0 : RD_COLS;
        inside = ((hi >= 32 ? 0u : (1u << hi)) - 1u) & ~((1u << lo) - 1u);
    }
    const bool border_wave = sx * useful - halo_l < 0 || sx * useful - halo_l + RD_WAVE > w;  // wave-uniform
    for (int op = 0; op < nops; op++) {
        const bool dil = (first_dil != 0) != (op == 1);
        if (dil) {
#pragma unroll
            for (int i = 0; i < RD_COLS; i++) m[i] ^= 0x00ff00ffu;
        }
        if (border_wave) {
#pragma unroll
            for (int i = 0; i < RD_COLS; i++) m[i] = (inside >> i) & 1u ? m[i] : 0x00ff00ffu;
        }
        rd_window_min(m, k, a);
        if (dil) {
#pragma unroll
            for (int i = 0; i < RD_COLS; i++) m[i] ^= 0x00ff00ffu;
        }
    }
    const int ux = sx * useful;
    if (in_row && x0 >= ux && x0 < ux + useful) {
        u32 oa[4], ob[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            oa[j] = __builtin_amdgcn_perm(m[4 * j + 1], m[4 * j], 0x0c0c0400u) | __builtin_amdgcn_perm(m[4 * j + 3], m[4 * j + 2], 0x04000c0cu);
            ob[j] = __builtin_amdgcn_perm(m[4 * j + 1], m[4 * j], 0x0c0c0602u) | __builtin_amdgcn_perm(m[4 * j + 3], m[4 * j + 2], 0x06020c0cu);
        }
        *(uint4*)(dst.row(b, y) + x0) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
        if (y + 1 < h) *(uint4*)(dst.row(b, y + 1) + x0) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
    }
}


// ---- small 2-D kernels (each side at most 5 after folding the iterations), one pass, any gray levels --------
// The layout of k_morph_row_dbl (a wave holds 1024 columns of a pair of rows, 16 columns per lane, the two rows in
// the halves of a dword) walking down the image: the row direction is done per pair by doubling, the column
// direction combines the last three pairs -- a window of at most five rows around a pair lies inside them --
// with one byte permutation (which replicates a row into the halves that take part, 0xffff elsewhere) and one
// packed min per source row.  An open / close runs as two such stages one pair apart.  Two-valued images need
// no bit plane, no "is it binary" pass and no workspace on this path: 2 B/px of traffic.
struct Small2DOp {
    u32 sel[6];  // per source row (pair p-1 low/high, p low/high, p+1 low/high) used by BOTH rows of the pair: selector, else 0
    int te, to;  // the source row only the even (te = 0..2) / only the odd (to = 3..5) output row uses
    u32 sel_eo;  // selector that puts row te into the low and row to into the high half
};

// out = per-half minimum over the source rows of pairs A (p-1), B (p), C (p+1) selected by S
__device__ __forceinline__ void s2d_vertical(const u32 (&A)[RD_COLS], const u32 (&B)[RD_COLS], const u32 (&C)[RD_COLS],
                                             const Small2DOp& S, u32 (&out)[RD_COLS])
{
#pragma unroll
    for (int i = 0; i < RD_COLS; i++) out[i] = 0xffffffffu;
#pragma unroll
    for (int t = 0; t < 6; t++) {
        const u32 sel = S.sel[t];
        if (sel) {  // (wave-uniform)
#pragma unroll
            for (int i = 0; i < RD_COLS; i++) {
                const u32 x = t < 2 ? A[i] : (t < 4 ? B[i] : C[i]);
                out[i] = pk_min16(out[i], __builtin_amdgcn_perm(x, x, sel));
            }
        }
    }
    // the first row of the even window and the last row of the odd window: one permutation for the two
#pragma unroll
    for (int te = 0; te < 3; te++)
#pragma unroll
        for (int to = 3; to < 6; to++)
            if (S.te == te && S.to == to) {  // (wave-uniform)
#pragma unroll
                for (int i = 0; i < RD_COLS; i++) {
                    const u32 xe = te < 2 ? A[i] : B[i], xo = to < 4 ? B[i] : C[i];
                    out[i] = pk_min16(out[i], __builtin_amdgcn_perm(xo, xe, S.sel_eo));
                }
            }
}

// selectors of one stage: output row r of pair p takes source rows r - ay .. r - ay + ky - 1
static Small2DOp s2d_selectors(int ky, int ay)
{
    Small2DOp S;
    for (int t = 0; t < 6; t++) {
        const int d = t - 2;  // source row 2p + d
        const bool even = d >= -ay && d <= ky - 1 - ay, odd = d >= 1 - ay && d <= ky - ay;
        const u32 half = (t & 1) ? 0x0302u : 0x0100u;
        S.sel[t] = (even && odd) ? ((half << 16) | half) : 0u;
    }
    S.te = 2 - ay;       // row 2p - ay: in the even window only
    S.to = ky - ay + 2;  // row 2p + ky - ay: in the odd window only
    S.sel_eo = (((S.to & 1) ? 0x0706u : 0x0504u) << 16) | ((S.te & 1) ? 0x0302u : 0x0100u);
    return S;
}

template <int NOPS>
__global__ __launch_bounds__(256) void k_morph_small2d(Plane src, Plane dst, int w, int h, int strips, int bands, int useful,
                                                       int first_dil, int kx, int ax, Small2DOp S, long long nwaves,
                                                       const int32_t* __restrict__ gate)
{
    constexpr int ITER = 36;                        // pairs a wave walks through (a multiple of 3)
    constexpr int BP = NOPS == 1 ? ITER - 2 : ITER - 4;  // pairs of output rows per band
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave_id >= nwaves) return;
    const int lane = threadIdx.x & 63;
    const u32 wid = (u32)wave_id;
    const u32 t1 = wid / (u32)strips;
    const int sx = (int)(wid - t1 * (u32)strips);
    const int b = (int)(t1 / (u32)bands), band = (int)(t1 - (u32)b * (u32)bands);
    if (gate && gate[b] == 0) return;  // only the images the two-valued kernel gave up on
    const int p0 = band * BP;  // first output pair
    const int x0 = sx * useful - RD_COLS + RD_COLS * lane;  // one lane of halo on each side
    const bool in_row = x0 >= 0 && x0 < w;
    u32 inside = 0;
    if (x0 + RD_COLS > 0 && x0 < w) {
        const int lo = x0 < 0 ? -x0 : 0, hi = w - x0 < RD_COLS ? w - x0 : RD_COLS;
        inside = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
    }
    const u8* const sbase = src.p + (size_t)b * src.stride + x0;
    u8* const dbase = dst.p + (size_t)b * dst.stride + x0;
    auto load_pair = [&](int p, uint4& ra, uint4& rb) {
        ra = make_uint4(0, 0, 0, 0);
        rb = make_uint4(0, 0, 0, 0);
        const int y = 2 * p;
        if (in_row && y >= 0 && y < h) ra = *(const uint4*)(sbase + (size_t)y * src.pitch);
        if (in_row && y + 1 >= 0 && y + 1 < h) rb = *(const uint4*)(sbase + (size_t)(y + 1) * src.pitch);
    };
    // values of a stage's input outside the image -> identity; `dil`: the stage works on complemented bytes
    auto mask_outside = [&](u32 (&m)[RD_COLS], int p) {
        const u32 keep = ((2 * p >= 0 && 2 * p < h) ? 0x0000ffffu : 0u) | ((2 * p + 1 >= 0 && 2 * p + 1 < h) ? 0xffff0000u : 0u);
#pragma unroll
        for (int i = 0; i < RD_COLS; i++) m[i] = (inside >> i) & 1u ? ((m[i] & keep) | (0x00ff00ffu & ~keep)) : 0x00ff00ffu;
    };
    const bool dil1 = first_dil != 0, dil2 = !dil1;
    u32 H1[3][RD_COLS], H2[NOPS == 2 ? 3 : 1][RD_COLS];
    uint4 na, nb;
    const int j0 = p0 - (NOPS == 1 ? 1 : 2);  // first pair loaded
    load_pair(j0, na, nb);
#pragma unroll 3
    for (int it = 0; it < ITER; it++) {
        const int j = j0 + it;
        const uint4 ra = na, rb = nb;
        load_pair(j + 1, na, nb);
        u32 m[RD_COLS];
        {
            const u32 A[4] = {ra.x, ra.y, ra.z, ra.w}, B[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                m[4 * q + 0] = __builtin_amdgcn_perm(B[q], A[q], 0x0c040c00u);
                m[4 * q + 1] = __builtin_amdgcn_perm(B[q], A[q], 0x0c050c01u);
                m[4 * q + 2] = __builtin_amdgcn_perm(B[q], A[q], 0x0c060c02u);
                m[4 * q + 3] = __builtin_amdgcn_perm(B[q], A[q], 0x0c070c03u);
            }
        }
        if (dil1) {
#pragma unroll
            for (int i = 0; i < RD_COLS; i++) m[i] ^= 0x00ff00ffu;
        }
        mask_outside(m, j);
        if (kx > 1) rd_window_min(m, kx, ax);
#pragma unroll
        for (int i = 0; i < RD_COLS; i++) H1[it % 3][i] = m[i];
        if (it < 2) continue;
        // stage 1 of pair j - 1 from the row passes of pairs j - 2, j - 1, j
        u32 e[RD_COLS];
        s2d_vertical(H1[(it + 1) % 3], H1[(it + 2) % 3], H1[it % 3], S, e);
        int po = j - 1;  // pair held by e
        if (NOPS == 2) {
            // second stage: the other kind of extremum, on stage 1's result
#pragma unroll
            for (int i = 0; i < RD_COLS; i++) e[i] ^= 0x00ff00ffu;
            mask_outside(e, j - 1);
            if (kx > 1) rd_window_min(e, kx, ax);
#pragma unroll
            for (int i = 0; i < RD_COLS; i++) H2[(it + 2) % 3][i] = e[i];
            if (it < 4) continue;
            s2d_vertical(H2[it % 3], H2[(it + 1) % 3], H2[(it + 2) % 3], S, e);
            po = j - 2;
        }
        if ((NOPS == 2 ? dil2 : dil1)) {
#pragma unroll
            for (int i = 0; i < RD_COLS; i++) e[i] ^= 0x00ff00ffu;
        }
        const int y = 2 * po;
        if (in_row && lane >= 1 && lane <= 62 && x0 < sx * useful + useful && po < p0 + BP && y < h) {
            u32 oa[4], ob[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                oa[q] = __builtin_amdgcn_perm(e[4 * q + 1], e[4 * q], 0x0c0c0400u) | __builtin_amdgcn_perm(e[4 * q + 3], e[4 * q + 2], 0x04000c0cu);
                ob[q] = __builtin_amdgcn_perm(e[4 * q + 1], e[4 * q], 0x0c0c0602u) | __builtin_amdgcn_perm(e[4 * q + 3], e[4 * q + 2], 0x06020c0cu);
            }
            *(uint4*)(dbase + (size_t)y * dst.pitch) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
            if (y + 1 < h) *(uint4*)(dbase + (size_t)(y + 1) * dst.pitch) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
        }
    }
}

// Columns: van Herk / Gil-Werman.  The rows are cut into blocks of k; within a block P[i] is the
// minimum from the block's first row to row i and S[i] the minimum from row i to its last row.  A
// window of k rows that starts at row lo of block j is S_j[lo] and P_{j+1}[lo + k - 1]: three
// min operations per pixel whatever k is.  One wave walks down a strip 256 columns wide, lane = 4
// adjacent columns (one dword); a block's rows are kept in LDS (k dwords per lane) while the prefix
// runs in a register, at the end of the block the stored rows are turned into S in place.  Rows
// outside the image count as the identity.  Bands of rows give the parallelism; a band starts at the
// block boundary at or above its first window.
// acc = per-byte min (max) of acc and v: four byte-selecting 16-bit instructions, each replacing one byte of acc
// (a partial write of a register needs one wait state before the register is read again)
template <bool DIL>
__device__ __forceinline__ void bytes_minmax_acc(u32& acc, u32 v)
{
    if (DIL)
        asm("v_max_u16_sdwa %0, %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\ts_nop 0\n\t"
            "v_max_u16_sdwa %0, %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1\n\ts_nop 0\n\t"
            "v_max_u16_sdwa %0, %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_2\n\ts_nop 0\n\t"
            "v_max_u16_sdwa %0, %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3\n\ts_nop 0"
            : "+v"(acc) : "v"(v));
    else
        asm("v_min_u16_sdwa %0, %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\ts_nop 0\n\t"
            "v_min_u16_sdwa %0, %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1\n\ts_nop 0\n\t"
            "v_min_u16_sdwa %0, %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_2\n\ts_nop 0\n\t"
            "v_min_u16_sdwa %0, %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3\n\ts_nop 0"
            : "+v"(acc) : "v"(v));
}

template <bool DIL>
__global__ __launch_bounds__(64) void k_morph_col_vhgw(Plane src, Plane dst, int w, int h, int dwords, int strips, int bands,
                                                       int band_rows, int k, int a, const int32_t* __restrict__ gate)
{
    extern __shared__ u32 ring[];  // [k][64]
    const int lane = threadIdx.x;
    int blk = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);  // neighbouring strips (they share cache lines) on one XCD
    const int sx = blk % strips;
    blk /= strips;
    const int band = blk % bands, b = blk / bands;
    if (gate && gate[b] == 0) return;
    const int dx = sx * 64 + lane;
    const bool live = dx < dwords;
    const int Y0 = band * band_rows, Y1 = min(Y0 + band_rows, h);
    const u32 ident = DIL ? 0u : 0xffffffffu;
    // first block: the one that holds row Y0 - a (floor division for negative rows)
    const int lo0 = Y0 - a;
    int r = (lo0 >= 0 ? lo0 / k : -((-lo0 + k - 1) / k)) * k;
    // Every lane loads every row (from the nearest dword and row inside the image; what lies outside is replaced by
    // the identity when the row is consumed): with loads that may or may not be issued the compiler cannot count
    // how many younger ones may stay in flight, and waits for all of them.
    const u8* const sbase = src.p + (size_t)b * src.stride + (size_t)min(dx, dwords - 1) * 4;
    u8* const dbase = dst.p + (size_t)b * dst.stride + (size_t)dx * 4;
    auto load_row = [&](int yy) -> u32 { return *(const u32*)(sbase + (size_t)min(max(yy, 0), h - 1) * src.pitch); };
    // Rows are taken sixteen at a time whatever the block length is, from two sets of eight registers: while one set
    // is consumed the other is loaded (eight rows ahead).  No value moves to another register while its load is
    // under way - a move makes the wave wait for every load it has issued.  The end of a block is a wave-uniform
    // branch behind any of the rows.
    constexpr int PF = 8;
    u32 qa[PF], qb[PF];
#pragma unroll
    for (int j = 0; j < PF; j++) qa[j] = load_row(r + j);
    const int y_last = Y1 + k - 2 - a;  // last source row of the band's last window
    bool have_prev = false;
    int i = 0;  // position of the next row inside its block
    u32 pre = ident;
    auto row = [&](const u32 loaded, const int y) {
        const u32 v = y >= 0 && y < h ? loaded : ident;
        bytes_minmax_acc<DIL>(pre, v);
        // the window that ends at row y: output row yo; its first row is row i + 1 of the previous block
        const int yo = y - (k - 1) + a;
        u32 out = pre;
        if (i < k - 1) {
            const u32 sprev = have_prev ? ring[(i + 1) * 64 + lane] : ident;
            bytes_minmax_acc<DIL>(out, sprev);
        }
        ring[i * 64 + lane] = v;
        // (rows past y_last in the last group write nothing: their output rows lie below the band)
        if (live && yo >= Y0 && yo < Y1 && (have_prev || i == k - 1)) *(u32*)(dbase + (size_t)yo * dst.pitch) = out;
        if (++i == k) {
            // suffix minima of the block, in place
            u32 suf = ring[(k - 1) * 64 + lane];
            for (int e = k - 2; e >= 0; e--) {
                bytes_minmax_acc<DIL>(suf, ring[e * 64 + lane]);
                ring[e * 64 + lane] = suf;
            }
            have_prev = true;
            i = 0;
            pre = ident;
        }
    };
    for (int yy = r; yy <= y_last; yy += 2 * PF) {
#pragma unroll
        for (int j = 0; j < PF; j++) {
            row(qa[j], yy + j);
            qb[j] = load_row(yy + PF + j);
        }
#pragma unroll
        for (int j = 0; j < PF; j++) {
            row(qb[j], yy + PF + j);
            qa[j] = load_row(yy + 2 * PF + j);
        }
    }
}

__global__ __launch_bounds__(256) void k_copy_plane(Plane src, Plane dst, int cpr, int h, long long total)
{
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += step) {
        int cx = (int)(c % cpr);
        long long t = c / cpr;
        int y = (int)(t % h), b = (int)(t / h);
        *(uint4*)(dst.row(b, y) + (size_t)cx * 16) = *(const uint4*)(src.row(b, y) + (size_t)cx * 16);
    }
}

int pow2_floor(int v)
{
    int p = 1;
    while (p * 2 <= v) p *= 2;
    return p;
}

}  // namespace

int launch_pack_bits(Plane src, u32* bits, size_t bpw, size_t bsw, int w, int h, int batch, int32_t* vmax,
                     hipStream_t st);

int launch_pack_bits(Plane src, u32* bits, size_t bpw, size_t bsw, int w, int h, int batch, int32_t* vmax,
                     hipStream_t st)
{
    int cpr = cdiv(w, 16);
    cpr += cpr & 1;
    if (vmax) IPA_HIP(hipMemsetAsync(vmax, 0, sizeof(int32_t) * (size_t)batch, st));  // (NULL: the bits only)
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        Plane sb = src;
        sb.p += (size_t)b0 * src.stride;
        hipLaunchKernelGGL(k_pack_bits, rows_grid(cpr, h, nb), rows_block(), 0, st, sb, bits + (size_t)b0 * bsw, bpw, bsw, w, h,
                           cpr, vmax ? vmax + b0 : nullptr);
        IPA_LAUNCH_CHECK();
    }
    return IPA_OK;
}

// Bit-plane morphology of `nops` stages (1: erode or dilate, 2: open or close)
// with folded kernel kx x ky and folded anchor (ax, ay).
int launch_bitmorph(const u32* bits, size_t bpw, size_t bsw, Plane dst, int w, int h, int batch, int first_is_dilate,
                    int nops, int kx, int ky, int ax, int ay, int value, const int32_t* value_dev, hipStream_t st,
                    u32* bits_out)
{
    static int fast = -1;  // IPA_BITMORPH_FAST=0 forces the LDS tile kernel
    if (fast < 0) {
        const char* e = getenv("IPA_BITMORPH_FAST");
        fast = e ? atoi(e) : 1;
    }
    static const size_t morph_pad = lds_pad_kib("IPA_PAD_MORPH");
    if (fast && ky == 1 && kx > 1 && nops * ax <= 32 && nops * (kx - 1 - ax) <= 32) {
        const int words = cdiv(w, 32);
        for (int b0 = 0; b0 < batch; b0 += 65535) {
            const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
            Plane db = dst;
            db.p += (size_t)b0 * dst.stride;
#define IPA_H3(N, K, A)                                                                                          \
    hipLaunchKernelGGL((k_bitmorph_h3<N, K, A>), rows_grid(words, h, nb), rows_block(), morph_pad, st, bits + (size_t)b0 * bsw, \
                       bpw, bsw, db, w, h, first_is_dilate, nops, kx, ax, pow2_floor(kx), (u32)value,                   \
                       value_dev ? value_dev + b0 : nullptr, bits_out ? bits_out + (size_t)b0 * bsw : nullptr)
            // the shapes of the parsers (15-pixel lines, 1 or 2 iterations) and of 3x3 open / close
            if (nops == 2 && kx == 29 && ax == 14) IPA_H3(2, 29, 14);
            else if (nops == 2 && kx == 15 && ax == 7) IPA_H3(2, 15, 7);
            else if (nops == 2 && kx == 3 && ax == 1) IPA_H3(2, 3, 1);
            else if (nops == 2 && kx == 5 && ax == 2) IPA_H3(2, 5, 2);
            else if (nops == 1 && kx == 3 && ax == 1) IPA_H3(1, 3, 1);
            else IPA_H3(0, 0, 0);
#undef IPA_H3
            IPA_LAUNCH_CHECK();
        }
        return IPA_OK;
    }
    if (fast && kx == 1 && ky > 1 && ky < 256) {
        const int words = cdiv(w, 32);
        int strip_rows = 128;
        // keep at least ~128k lanes (2048 waves) in flight for single images (one 16384 x 16384 plane, 201 rows of
        // kernel: 0.207 ms with strips of 128 rows, 0.177 ms with 64 although a strip then reads 400 rows of run-in)
        while (strip_rows > 16 && (long long)words * cdiv(h, strip_rows) * batch <= 65536) strip_rows /= 2;
        const int strips = cdiv(h, strip_rows);
        for (int b0 = 0; b0 < batch; b0 += 65535) {
            const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
            Plane db = dst;
            db.p += (size_t)b0 * dst.stride;
            const dim3 grid((unsigned)cdiv(words, 64), (unsigned)cdiv(strips, 4), (unsigned)nb), block(64, 4);
            if (ky < 32)
                hipLaunchKernelGGL(k_bitmorph_v<5>, grid, block, morph_pad, st, bits + (size_t)b0 * bsw, bpw, bsw, db, w, h, words,
                                   strips, strip_rows, first_is_dilate, nops, ky, ay, (u32)value,
                                   value_dev ? value_dev + b0 : nullptr, bits_out ? bits_out + (size_t)b0 * bsw : nullptr);
            else
                hipLaunchKernelGGL(k_bitmorph_v<8>, grid, block, 0, st, bits + (size_t)b0 * bsw, bpw, bsw, db, w, h, words,
                                   strips, strip_rows, first_is_dilate, nops, ky, ay, (u32)value,
                                   value_dev ? value_dev + b0 : nullptr, bits_out ? bits_out + (size_t)b0 * bsw : nullptr);
            IPA_LAUNCH_CHECK();
        }
        return IPA_OK;
    }
    BitMorphArgs A;
    A.bits = bits; A.bpw = bpw; A.bsw = bsw; A.dst = dst; A.w = w; A.h = h;
    A.first_is_dilate = first_is_dilate; A.nops = nops;
    A.kx = kx; A.ky = ky; A.ax = ax; A.ay = ay;
    A.px = pow2_floor(kx); A.py = pow2_floor(ky);
    A.value = value; A.value_dev = value_dev;
    // reach of the whole op sequence, per side
    const int left = nops * ax, right = nops * (kx - 1 - ax);
    A.hlw = kx > 1 ? cdiv(left, 32) : 0;
    A.hrw = kx > 1 ? cdiv(right, 32) + 1 : 0;
    A.ht = nops * ay;
    A.hb = nops * (ky - 1 - ay);
    const int row_words = cdiv(w, 32);
    const int budget = 8192;  // words per LDS copy (32 KiB); two copies
    int tww, tr;
    if (ky == 1) {
        // horizontal work: wide tiles, a few rows.  Small tiles (12 KiB per copy, five or six workgroups per CU): a
        // workgroup spends most of its time writing its rows back as bytes, and more of them in flight keep the
        // stores going (open 201 x 1 of one 16384 x 16384 bit plane: 0.32 ms with 32 KiB per copy, 0.175 ms with 12,
        // 0.19 ms with 4)
        const int budget_h = 3072;
        tww = row_words < 256 ? row_words : 256;
        tr = budget_h / (tww + A.hlw + A.hrw);
        if (tr > 64) tr = 64;
    } else {
        tww = row_words < 32 ? row_words : 32;
        if (kx > 1 && tww < 64 && row_words >= 64) tww = 64;
        tr = budget / (tww + A.hlw + A.hrw) - A.ht - A.hb;
    }
    while (tr < 8 && tww > 1) {
        tww = (tww + 1) / 2;
        tr = budget / (tww + A.hlw + A.hrw) - A.ht - A.hb;
    }
    if (tr > h) tr = h;
    if (tr < 1) {
        ipa_set_error("morphology kernel %dx%d (iterations folded) is too large for the bit-plane tile", kx, ky);
        return IPA_ERR_ARG;
    }
    A.tww = tww; A.tr = tr;
    A.tiles_x = cdiv(row_words, tww);
    A.tiles_y = cdiv(h, tr);
    size_t lds = (size_t)2 * (tww + A.hlw + A.hrw) * (tr + A.ht + A.hb) * sizeof(u32);
    long long blocks = (long long)A.tiles_x * A.tiles_y * batch;
    if (blocks > 0x7fffffffLL) {
        ipa_set_error("morphology: too many tiles");
        return IPA_ERR_ARG;
    }
    hipLaunchKernelGGL(k_bitmorph, dim3((unsigned)blocks), dim3(256), lds, st, A);
    IPA_LAUNCH_CHECK();
    // (the tile kernel writes bytes only: its result is packed again when the caller wants the bit plane)
    if (bits_out) return launch_pack_bits(dst, bits_out, bpw, bsw, w, h, batch, nullptr, st);
    return IPA_OK;
}

// Row pass in registers (k_morph_row_dbl): nops stages with the same kernel length kx and anchor ax.
// Returns 1 when the shape is outside what the kernel covers (the caller then uses k_morph_row).
static int launch_morph_row_dbl(Plane src, Plane dst, int w, int h, int batch, int nops, int first_dil, int kx, int ax,
                                hipStream_t st, const int32_t* gate = nullptr)
{
    const int halo_l = (int)align_up((size_t)(nops * ax), 16), halo_r = nops * (kx - 1 - ax);
    const int useful = (RD_WAVE - halo_l - halo_r) & ~15;
    if (useful < 256) return 1;  // long kernels: most of a wave would be halo
    const int strips = cdiv(w, useful), pairs = cdiv(h, 2);
    const long long nwaves = (long long)strips * pairs * batch;
    if (nwaves >= 0x7fffffffLL) return 1;
    hipLaunchKernelGGL(k_morph_row_dbl, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, src, dst, w, h, strips, pairs, useful,
                       halo_l, nops, first_dil, kx, ax, nwaves, gate);
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}

// Small 2-D kernels in one pass (k_morph_small2d).  Returns 1 when the shape is outside what the kernel covers.
static bool small2d_covers(int nops, int kx, int ky, int ax, int ay)
{
    if (ky < 2 || ky > 5 || kx > 7 || ay > 2 || ky - ay > 3) return false;
    return nops * ax <= RD_COLS && nops * (kx - 1 - ax) <= RD_COLS;
}
static int launch_morph_small2d(Plane src, Plane dst, int w, int h, int batch, int nops, int first_dil, int kx, int ky, int ax,
                                int ay, const int32_t* gate, hipStream_t st)
{
    if (!small2d_covers(nops, kx, ky, ax, ay)) return 1;
    const int useful = RD_WAVE - 2 * RD_COLS;
    const int bp = nops == 1 ? 34 : 32;
    const int strips = cdiv(w, useful), bands = cdiv(cdiv(h, 2), bp);
    const long long nwaves = (long long)strips * bands * batch;
    if (nwaves >= 0x7fffffffLL) return 1;
    const Small2DOp S = s2d_selectors(ky, ay);
    const dim3 grid((unsigned)((nwaves + 3) / 4));
    if (nops == 1)
        hipLaunchKernelGGL(k_morph_small2d<1>, grid, dim3(256), 0, st, src, dst, w, h, strips, bands, useful, first_dil, kx, ax, S, nwaves, gate);
    else
        hipLaunchKernelGGL(k_morph_small2d<2>, grid, dim3(256), 0, st, src, dst, w, h, strips, bands, useful, first_dil, kx, ax, S, nwaves, gate);
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}


// Two-valued images in one pass over the bytes (k_binmorph2d).  flags != nullptr: the images are not known to be
// two-valued; flags[b] (zeroed here) is set for those that are not, and their output is to be redone.
// Returns 1 when the shape is outside what the kernel covers.
static int launch_binmorph2d(Plane src, Plane dst, int w, int h, int batch, int nops, int first_dil, int kx, int ky, int ax,
                             int ay, int32_t* flags, hipStream_t st)
{
    if (ax > 32 || kx - 1 - ax > 32 || ky > 31) return 1;
    static int on = -1;  // IPA_BINMORPH=0: the bit-plane kernels / the gray-level kernels as before (A/B)
    if (on < 0) {
        const char* e = getenv("IPA_BINMORPH");
        on = e ? atoi(e) : 1;
    }
    if (!on) return 1;
    int px = 1;
    while (px * 2 <= kx) px *= 2;
    const int useful = 64 - 2 * nops, words = cdiv(w, 32);
    const int strips_x = cdiv(words, useful);
    // strips of 128 rows; shorter ones (every strip re-reads nops * (ky - 1) rows) while the launch has few waves
    int strip_rows = 128;
    while (strip_rows > 32 && (long long)strips_x * cdiv(h, strip_rows) * batch < 4096) strip_rows >>= 1;
    const int strips_y = cdiv(h, strip_rows);
    if (flags) IPA_HIP(hipMemsetAsync(flags, 0, sizeof(int32_t) * (size_t)batch, st));
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        Plane sb = src, db = dst;
        sb.p += (size_t)b0 * src.stride;
        db.p += (size_t)b0 * dst.stride;
        const dim3 grid((unsigned)strips_x, (unsigned)cdiv(strips_y, 4), (unsigned)nb);
        int32_t* fb = flags ? flags + b0 : nullptr;
#define IPA_BM(NS, CHK)                                                                                                     \
    hipLaunchKernelGGL((k_binmorph2d<NS, CHK>), grid, rows_block(), 0, st, sb, db, w, h, strips_y, strip_rows, useful, first_dil, \
                       nops, kx, ax, px, ky, ay, fb)
        if (ky < 8) {
            if (flags) IPA_BM(3, true);
            else IPA_BM(3, false);
        } else {
            if (flags) IPA_BM(5, true);
            else IPA_BM(5, false);
        }
#undef IPA_BM
        IPA_LAUNCH_CHECK();
    }
    return IPA_OK;
}

static bool g_vhgw_attr_set = false;

// One general 8-bit erode/dilate pass pair (rows then columns) with folded sizes.
int launch_morph_generic(Plane src, Plane dst, Plane tmp, int w, int h, int batch, int is_dilate, int kx, int ky,
                         int ax, int ay, hipStream_t st, const int32_t* gate)
{
    Plane cur = src;
    if (kx > 1) {
        Plane out = ky > 1 ? tmp : dst;
        int rc = launch_morph_row_dbl(cur, out, w, h, batch, 1, is_dilate, kx, ax, st, gate);
        if (rc < 0) return rc;
        if (rc == 1) {
            int segs = cdiv(w, 1024);
            int L = (ax + 3) & ~3, R = (kx - 1 - ax + 3) & ~3;
            size_t lds = (size_t)1024 + L + R;
            if (lds > 64 * 1024) {
                ipa_set_error("morphology: kernel width %d too large", kx);
                return IPA_ERR_ARG;
            }
            long long blocks = (long long)segs * h * batch;
            hipLaunchKernelGGL(k_morph_row, dim3((unsigned)blocks), dim3(256), lds, st, cur, out, w, h, segs, kx, ax,
                               is_dilate, gate);
            IPA_LAUNCH_CHECK();
        }
        cur = out;
    }
    if (ky > 1) {
        const int dwords = cdiv(w, 4);
        const size_t lds = (size_t)ky * 64 * sizeof(u32);
        if (ky >= 6 && lds <= 150 * 1024) {
            // van Herk / Gil-Werman down the columns; bands of rows (at least four kernel lengths each, so that
            // the rows a band reads above its first output stay a small part) give the launch its parallelism
            const int strips = cdiv(dwords, 64);
            int band_rows = h;
            static int vhgw_blocks = -1;  // IPA_VHGW_BLOCKS: workgroups a launch should reach before bands stop shrinking
            if (vhgw_blocks < 0) {
                const char* e = getenv("IPA_VHGW_BLOCKS");
                vhgw_blocks = e ? atoi(e) : 4096;
            }
            while (band_rows / 2 >= 4 * ky && (long long)strips * cdiv(h, band_rows) * batch < vhgw_blocks) band_rows = (band_rows + 1) / 2;
            const int bands = cdiv(h, band_rows);
            const long long blocks = (long long)strips * bands * batch;
            if (blocks > 0x7fffffffLL) {
                ipa_set_error("morphology: too many strips");
                return IPA_ERR_ARG;
            }
            if (lds > 48 * 1024 && !g_vhgw_attr_set) {
                IPA_HIP(hipFuncSetAttribute((const void*)k_morph_col_vhgw<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                IPA_HIP(hipFuncSetAttribute((const void*)k_morph_col_vhgw<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                g_vhgw_attr_set = true;
            }
            if (is_dilate)
                hipLaunchKernelGGL(k_morph_col_vhgw<true>, dim3((unsigned)blocks), dim3(64), lds, st, cur, dst, w, h, dwords, strips, bands,
                                   band_rows, ky, ay, gate);
            else
                hipLaunchKernelGGL(k_morph_col_vhgw<false>, dim3((unsigned)blocks), dim3(64), lds, st, cur, dst, w, h, dwords, strips, bands,
                                   band_rows, ky, ay, gate);
            IPA_LAUNCH_CHECK();
        } else {
            int rowgroups = cdiv(h, 4);
            long long total = (long long)dwords * rowgroups * batch;
            hipLaunchKernelGGL(k_morph_col, dim3(stream_grid(total, 256)), dim3(256), 0, st, cur, dst, w, h, dwords,
                               rowgroups, ky, ay, is_dilate, total, gate);
            IPA_LAUNCH_CHECK();
        }
        cur = dst;
    }
    if (cur.p == src.p) {
        int cpr = cdiv(w, 16);
        long long total = (long long)cpr * h * batch;
        hipLaunchKernelGGL(k_copy_plane, dim3(stream_grid(total, 256)), dim3(256), 0, st, src, dst, cpr, h, total);
        IPA_LAUNCH_CHECK();
    }
    return IPA_OK;
}

// workspace: bit plane + per-image value + two 8-bit temporaries (general path)
static size_t morph_tmp_pitch(int w) { return align_up((size_t)w, 256); }

IPA_EXPORT size_t ipa_morph_workspace_bytes(int width, int height, int batch)
{
    if (width <= 0 || height <= 0 || batch <= 0) return 0;
    Arena a(nullptr, 0);
    a.take(bits_pitch_words(width) * 4 * (size_t)height * batch);
    a.take(sizeof(int32_t) * (size_t)batch);
    a.take(morph_tmp_pitch(width) * (size_t)height * batch);
    a.take(morph_tmp_pitch(width) * (size_t)height * batch);
    return a.off;
}

IPA_EXPORT int ipa_binary_check_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, int width, int height,
                                   int batch, int32_t* flags, void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    if (batch == 0) return IPA_OK;
    IPA_REQUIRE(flags != nullptr, IPA_ERR_ARG, "binary_check: flags is NULL");
    IPA_HIP(hipMemsetAsync(flags, 0, sizeof(int32_t) * (size_t)batch, (hipStream_t)stream));
    if (width == 0 || height == 0) return IPA_OK;
    Plane s = mkplane(src, src_pitch, src_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch), IPA_ERR_ALIGN,
                "binary_check: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    const int cpr = cdiv(width, 16);
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        Plane sb = s;
        sb.p += (size_t)b0 * s.stride;
        hipLaunchKernelGGL(k_binary_check, rows_grid(cpr, height, nb), rows_block(), 0, (hipStream_t)stream, sb, width, height,
                           cpr, flags + b0);
        IPA_LAUNCH_CHECK();
    }
    return IPA_OK;
}

IPA_EXPORT int ipa_morph_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst, size_t dst_pitch,
                            size_t dst_stride, int width, int height, int batch, int op, int kw, int kh, int anchor_x,
                            int anchor_y, int iterations, int assume_binary, void* ws, size_t ws_bytes, void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    IPA_REQUIRE(op >= 0 && op <= 3, IPA_ERR_ARG, "morphology: op must be ERODE, DILATE, OPEN or CLOSE");
    IPA_REQUIRE(kw >= 1 && kh >= 1, IPA_ERR_ARG, "morphology: empty structuring element");
    IPA_REQUIRE(iterations >= 0, IPA_ERR_ARG, "morphology: negative iterations");
    if (anchor_x < 0) anchor_x = kw / 2;
    if (anchor_y < 0) anchor_y = kh / 2;
    IPA_REQUIRE(anchor_x < kw && anchor_y < kh, IPA_ERR_ARG, "morphology: anchor outside the kernel");
    if (width == 0 || height == 0 || batch == 0) return IPA_OK;
    hipStream_t st = (hipStream_t)stream;
    Plane s = mkplane(src, src_pitch, src_stride), d = mkplane(dst, dst_pitch, dst_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch) && plane_ok(d, (size_t)width, batch), IPA_ERR_ALIGN,
                "morphology: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    int n = iterations;
    if (n == 0 || (kw == 1 && kh == 1)) {
        int cpr = cdiv(width, 16);
        long long total = (long long)cpr * height * batch;
        hipLaunchKernelGGL(k_copy_plane, dim3(stream_grid(total, 256)), dim3(256), 0, st, s, d, cpr, height, total);
        IPA_LAUNCH_CHECK();
        return IPA_OK;
    }
    const int KX = kw + (n - 1) * (kw - 1), KY = kh + (n - 1) * (kh - 1);
    const int AX = anchor_x * n, AY = anchor_y * n;
    IPA_REQUIRE(ws_bytes >= ipa_morph_workspace_bytes(width, height, batch) && ws, IPA_ERR_WORKSPACE,
                "morphology: workspace too small");
    Arena ar(ws, ws_bytes);
    size_t bpw = bits_pitch_words(width);
    size_t bsw = bpw * (size_t)height;
    u32* bits = (u32*)ar.take(bsw * 4 * (size_t)batch);
    int32_t* vmax = (int32_t*)ar.take(sizeof(int32_t) * (size_t)batch);
    size_t tp = morph_tmp_pitch(width);
    Plane t1 = mkplane(ar.take(tp * (size_t)height * batch), tp, tp * (size_t)height);
    Plane t2 = mkplane(ar.take(tp * (size_t)height * batch), tp, tp * (size_t)height);
    const int nops = (op == IPA_MORPH_OPEN || op == IPA_MORPH_CLOSE) ? 2 : 1;
    const int first_dil = (op == IPA_MORPH_DILATE || op == IPA_MORPH_CLOSE) ? 1 : 0;
    // Two-valued images (every call of the reference): one pass over the bytes through bit words in registers.
    // assume_binary: by the caller's word.  Otherwise the same kernel checks what it reads, gives up on an image
    // as soon as it meets a third value and flags it; the gray-level kernels below then run for the flagged images
    // only (they read the flags).  A thresholded plane costs one pass, a gray image the gray-level passes and the
    // few rows the first kernel read.
    const int32_t* gate = nullptr;
    {
        const int rcb = launch_binmorph2d(s, d, width, height, batch, nops, first_dil, KX, KY, AX, AY, assume_binary ? nullptr : vmax, st);
        if (rcb < 0 || (rcb == 0 && assume_binary)) return rcb;
        if (rcb == 0) gate = vmax;
    }
    {   // small 2-D kernels: one pass over the bytes
        const int rc2 = launch_morph_small2d(s, d, width, height, batch, nops, first_dil, KX, KY, AX, AY, gate, st);
        if (rc2 <= 0) return rc2;
    }
    if (assume_binary) {  // (rectangles beyond what the one-pass kernel covers)
        int rc = launch_pack_bits(s, bits, bpw, bsw, width, height, batch, vmax, st);
        if (rc) return rc;
        return launch_bitmorph(bits, bpw, bsw, d, width, height, batch, first_dil, nops, KX, KY, AX, AY, 0, vmax, st);
    }
    if (nops == 1) return launch_morph_generic(s, d, t1, width, height, batch, first_dil, KX, KY, AX, AY, st, gate);
    if (KY == 1) {  // open / close along the rows only: both stages in one pass over the registers
        const int rc1 = launch_morph_row_dbl(s, d, width, height, batch, 2, first_dil, KX, AX, st, gate);
        if (rc1 <= 0) return rc1;
    }
    int rc = launch_morph_generic(s, t2, t1, width, height, batch, first_dil, KX, KY, AX, AY, st, gate);
    if (rc) return rc;
    return launch_morph_generic(t2, d, t1, width, height, batch, !first_dil, KX, KY, AX, AY, st, gate);
}
