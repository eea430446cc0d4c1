// Device code of the BGR -> gray / threshold / bit-plane conversion that more than one kernel runs: the
// streaming kernel of color.hip and the spare waves of the Hough vote kernel (hough.hip), which convert the
// next part of a batch while the other waves vote.
#pragma once
#include "common.hpp"

// Sixteen pixels (48 bytes in d[0..11]) -> their gray values in g[0..3].  The luma is computed scaled by two,
// 2 * 3735 = 29 * 256 + 46, 2 * 19235 = 150 * 256 + 70, 2 * 9798 = 76 * 256 + 140 (bias 32768), so that the gray
// value is byte 2 of the 24-bit result and four results are packed by three byte permutes; a pixel that starts
// at byte 1 of a dword is read through coefficients moved up by one byte instead of being shifted down first.
__device__ __forceinline__ u32 luma2(u32 px, bool up)
{
    const u32 hi = __builtin_amdgcn_udot4(px, up ? 0x4c961d00u : 0x004c961du, 0u, false);
    const u32 lo = __builtin_amdgcn_udot4(px, up ? 0x8c462e00u : 0x008c462eu, 32768u, false);
    return (hi << 8) + lo;
}
__device__ __forceinline__ void gray16(const u32 (&d)[13], u32 (&g)[4])
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        // pixels 4q .. 4q+3 live in dwords 3q .. 3q+2 at byte offsets 0, 3, 6, 9
        const u32 t0 = luma2(d[3 * q], false);
        const u32 t1 = luma2(__builtin_amdgcn_alignbyte(d[3 * q + 1], d[3 * q], 3), false);
        const u32 t2 = luma2(__builtin_amdgcn_alignbyte(d[3 * q + 2], d[3 * q + 1], 2), false);
        const u32 t3 = luma2(d[3 * q + 2], true);
        const u32 lo = __builtin_amdgcn_perm(t1, t0, 0x0c0c0602u), hi = __builtin_amdgcn_perm(t3, t2, 0x06020c0cu);
        g[q] = lo | hi;
    }
}
// Threshold of four gray bytes at once: "above" <=> gray + k >= 256 with k = 255 - t in 0 .. 255.  k7 = (k & 0x7f) in
// every byte, kb = 0x80808080 when k >= 128 else 0: the carry into bit 7 comes from the sum of the low seven bits,
// the carry out of the byte is the majority of that carry and the two top bits.  Returns 1 in every byte above.
__device__ __forceinline__ u32 above4(u32 g, u32 k7, u32 kb)
{
    const u32 L = (g & 0x7f7f7f7fu) + k7;
    const u32 R = (L & (g | kb)) | (g & kb);
    return (R >> 7) & 0x01010101u;
}
// thresholded bytes and the 16 "above" bits of the lane's 16 pixels
__device__ __forceinline__ u32 thresh16(const u32 (&g)[4], u32 (&t)[4], u32 k7, u32 kb, u32 a4, u32 b4)
{
    u32 above = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const u32 b1 = above4(g[q], k7, kb);
        const u32 m = (b1 << 8) - b1;  // 0xff in the bytes above
        t[q] = (m & a4) | (~m & b4);
        above |= __builtin_amdgcn_udot4(b1, 0x08040201u, 0u, false) << (4 * q);
    }
    return above;
}
// (k7, kb) for a threshold in -1 .. 256 and the values above / below, swapped when every pixel is above (t < 0)
struct ThreshK {
    u32 k7, kb, va, vb;
};
static inline ThreshK thresh_consts(int ithresh, u32 v_above, u32 v_below)
{
    ThreshK r;
    int k = 255 - (ithresh > 255 ? 255 : ithresh);
    r.va = v_above;
    r.vb = v_below;
    if (ithresh < 0) {  // every pixel is above: "never above" with the two values swapped
        r.va = v_below;
        r.vb = v_above;
        k = 0;
    }
    r.k7 = (u32)(k & 0x7f) * 0x01010101u;
    r.kb = k >= 128 ? 0x80808080u : 0u;
    return r;
}

// One launch (or side job) of the flat conversion: planes of the first image, geometry and threshold constants.
// cpr = 16-pixel chunks per row, rounded up to an even number so that two neighbouring lanes always hold the two
// halves of one 32-bit word of the bit plane; row = chunk / cpr by multiplication: chunk < 2^28,
// div_m = floor(2^div_k / cpr) + 1, div_k = 28 + bits(cpr).
struct GrayArgs {
    Plane bgr, gray, bin;
    u32* bits;
    size_t bpw, bsw;
    int w, h, cpr;
    u32 nchunks, div_m;
    int div_k, ithresh;  // ithresh in 0 .. 255 (thresh_consts), only read by lanes that are partly inside the image
    u32 va, vb, k7, kb;
};

// false when the image has too many chunks for the 32-bit row arithmetic
static inline bool gray_args_make(GrayArgs* G, Plane bgr, Plane gray, Plane bin, u32* bits, size_t bpw, size_t bsw, int w, int h,
                                  int ithresh, u32 va, u32 vb, u32 k7, u32 kb)
{
    int cpr = cdiv(w, 16);
    cpr += cpr & 1;
    if (cpr <= 0 || h <= 0 || (unsigned long long)cpr * h >= (1ull << 28)) return false;
    G->bgr = bgr; G->gray = gray; G->bin = bin; G->bits = bits; G->bpw = bpw; G->bsw = bsw;
    G->w = w; G->h = h; G->cpr = cpr;
    G->nchunks = (u32)cpr * (u32)h;
    int lb = 0;
    while ((1u << lb) <= (u32)cpr) lb++;  // cpr < 2^lb
    G->div_k = 28 + lb;
    G->div_m = (u32)((1ull << G->div_k) / (u32)cpr) + 1u;
    G->ithresh = ithresh; G->va = va; G->vb = vb; G->k7 = k7; G->kb = kb;
    return true;
}

// A conversion handed to another kernel's spare waves: `total_units` units of 64 consecutive chunks each.
struct GrayJob {
    GrayArgs G;
    u32 units_per_image;  // ceil(chunks per image / 64): one unit = one chunk for every lane of a wave
    u32 total_units;      // units_per_image * images of the job
};

// Chunk `ci` (16 pixels, 48 input bytes) of image `b`: consecutive lanes take consecutive chunks of the image in
// row-major order, so a wave covers 1024 consecutive pixels (wrapping over row ends) and stores one whole 128-byte
// line of a bit-plane row (a lane stores its 16 bits as two bytes).
template <bool THRESH, bool PACK>
__device__ __forceinline__ void gray_flat_chunk(const GrayArgs& G, int b, u32 ci)
{
    const u32 y = (u32)(((unsigned long long)ci * G.div_m) >> G.div_k);
    const int cx = (int)(ci - y * (u32)G.cpr);
    const bool live = cx * 16 < G.w;
    const bool full = cx * 16 + 16 <= G.w;
    const u8* const srow = G.bgr.p + (size_t)b * G.bgr.stride + (size_t)y * G.bgr.pitch + (size_t)cx * 48u;
    u8* const gd = G.gray.p + (size_t)b * G.gray.stride + (size_t)y * G.gray.pitch + (size_t)cx * 16u;
    u8* const td = THRESH ? G.bin.p + (size_t)b * G.bin.stride + (size_t)y * G.bin.pitch + (size_t)cx * 16u : nullptr;
    const u32 a4 = G.va * 0x01010101u, b4 = G.vb * 0x01010101u;
    const u32 nz_a = G.va ? 0xffffu : 0u, nz_b = G.vb ? 0xffffu : 0u;
    u32 mybits = 0;
    if (live && full) {
        const uint4* s4 = (const uint4*)srow;
        const uint4 i0 = s4[0], i1 = s4[1], i2 = s4[2];
        const u32 d[13] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w, 0u};
        u32 g[4];
        gray16(d, g);
        *(uint4*)gd = make_uint4(g[0], g[1], g[2], g[3]);
        if (THRESH) {
            u32 t[4];
            const u32 above = thresh16(g, t, G.k7, G.kb, a4, b4);  // bit k: pixel k is above the threshold
            *(uint4*)td = make_uint4(t[0], t[1], t[2], t[3]);
            if (PACK) mybits = (above & nz_a) | (~above & nz_b);
        }
    } else if (live) {
        const int n = G.w - cx * 16;
        for (int k = 0; k < n; k++) {
            const u32 v = (srow[3 * k] * 3735u + srow[3 * k + 1] * 19235u + srow[3 * k + 2] * 9798u + 16384u) >> 15;
            gd[k] = (u8)v;
            if (THRESH) {
                const bool above = (int)v > G.ithresh;
                td[k] = (u8)(above ? G.va : G.vb);
                if (PACK) mybits |= (u32)((above ? G.va : G.vb) != 0u) << k;
            }
        }
    }
    // (cpr is even: the chunk behind the last one of an odd row writes the zero half of the last word)
    if (PACK) *(unsigned short*)((u8*)(G.bits + (size_t)b * G.bsw + (size_t)y * G.bpw) + (size_t)cx * 2u) = (unsigned short)mybits;
}
