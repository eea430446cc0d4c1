// Streaming per-pixel kernels: BGR->gray (optionally fused with the binary
// threshold and a 1-bit-per-pixel copy of it), channel shuffles, bitwise ops,
// fixed / Otsu thresholds and the 256-bin histogram.
//
// Every kernel moves 16 pixels per lane with 16-byte accesses; a wave64 touches
// 1 KiB (gray) or 3 KiB (BGR) of contiguous memory per row segment.
#include "common.hpp"
#include "gray_core.hpp"

#include <float.h>

namespace {

__device__ __forceinline__ u32 byte_of(const u32* d, int k) { return (d[k >> 2] >> (8 * (k & 3))) & 0xffu; }

// OpenCV 4.x 8-bit BGR2GRAY: Q15 coefficients, round to nearest.
__device__ __forceinline__ u32 luma(u32 b, u32 g, u32 r) { return (b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15; }

// ---------------------------------------------------------------------------
// BGR -> gray [, thresholded plane][, bit plane].  One lane = one 16-pixel chunk
// (48 input bytes).  `cpr` = chunks per row, rounded up to an even number so
// that lanes 2k / 2k+1 of a wave always hold the two halves of one 32-bit word
// of the bit plane.
// Q15 luma of the pixel in the low three bytes of `px` (B, G, R; the top byte is ignored) with
// two 4-way byte dot products: 3735 = 14*256 + 151, 19235 = 75*256 + 35, 9798 = 38*256 + 70.
__device__ __forceinline__ u32 luma_dot(u32 px)
{
    const u32 hi = __builtin_amdgcn_udot4(px, 0x00264b0eu, 0u, false);
    const u32 lo = __builtin_amdgcn_udot4(px, 0x00462397u, 16384u, false);
    return ((hi << 8) + lo) >> 15;
}

template <bool THRESH, bool PACK>
__global__ __launch_bounds__(256) void k_bgr2gray(Plane bgr, Plane gray, Plane bin, u32* __restrict__ bits,
                                                  size_t bpw, size_t bsw, int w, int h, int cpr, int ithresh,
                                                  u32 v_above, u32 v_below, u32 k7, u32 kb, ClearList clr)
{
    if (clr.n[0] | clr.n[1] | clr.n[2]) {  // (wave-uniform) the first workgroup zeroes the scratch of the stages that follow
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
            const u32 t = threadIdx.y * 64 + threadIdx.x;
            for (int a = 0; a < 3; a++)
                for (u32 i = t; i < clr.n[a]; i += 256) clr.p[a][i] = 0;
        }
    }
    const int cx = blockIdx.x * 64 + threadIdx.x;
    const int b = blockIdx.z;
    // a wave's lanes share threadIdx.y: row numbers and row addresses are wave-uniform (scalar)
    const int y0 = ((int)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y)) * ROWS_PER_LANE;
    const bool live = cx < cpr && cx * 16 < w;
    const bool full = cx * 16 + 16 <= w;
    const u8* const simg = bgr.p + (size_t)b * bgr.stride;
    u8* const gimg = gray.p + (size_t)b * gray.stride;
    u8* const timg = THRESH ? bin.p + (size_t)b * bin.stride : nullptr;
    const u32 soff = (u32)cx * 48u, doff = (u32)cx * 16u;
    // all loads of the lane's rows are issued before the first is used
    uint4 in[ROWS_PER_LANE][3];
#pragma unroll
    for (int i = 0; i < ROWS_PER_LANE; i++) {
        const int y = y0 + i;
        if (live && full && y < h) {
            const uint4* s4 = (const uint4*)(simg + (size_t)y * bgr.pitch + soff);
            in[i][0] = s4[0]; in[i][1] = s4[1]; in[i][2] = s4[2];
        }
    }
    // thresholded bytes / bit-plane bits as functions of "above the threshold"
    const u32 a4 = v_above * 0x01010101u, b4 = v_below * 0x01010101u;
    const u32 nz_a = v_above ? 0xffffu : 0u, nz_b = v_below ? 0xffffu : 0u;
    u32 rowbits[ROWS_PER_LANE] = {0, 0, 0, 0};  // the lane's 16 bit-plane bits of each row
#pragma unroll
    for (int i = 0; i < ROWS_PER_LANE; i++) {
        const int y = y0 + i;
        if (y >= h) break;  // uniform per wave
        u32 mybits = 0;
        if (live) {
            u8* gd = gimg + (size_t)y * gray.pitch + doff;
            if (full) {
                const u32 d[13] = {in[i][0].x, in[i][0].y, in[i][0].z, in[i][0].w, in[i][1].x, in[i][1].y, in[i][1].z,
                                   in[i][1].w, in[i][2].x, in[i][2].y, in[i][2].z, in[i][2].w, 0u};
                u32 g[4];
                gray16(d, g);
                *(uint4*)gd = make_uint4(g[0], g[1], g[2], g[3]);
                if (THRESH) {
                    u32 t[4];
                    const u32 above = thresh16(g, t, k7, kb, a4, b4);  // bit k: pixel k is above the threshold
                    *(uint4*)(timg + (size_t)y * bin.pitch + doff) = make_uint4(t[0], t[1], t[2], t[3]);
                    if (PACK) mybits = (above & nz_a) | (~above & nz_b);
                }
            } else {
                const u8* s = simg + (size_t)y * bgr.pitch + soff;
                const int n = w - cx * 16;
                u8* td = THRESH ? timg + (size_t)y * bin.pitch + doff : nullptr;
                for (int k = 0; k < n; k++) {
                    u32 v = luma(s[3 * k], s[3 * k + 1], s[3 * k + 2]);
                    gd[k] = (u8)v;
                    if (THRESH) {
                        bool above = (int)v > ithresh;
                        td[k] = (u8)(above ? v_above : v_below);
                        if (PACK) mybits |= (u32)((above ? v_above : v_below) != 0u) << k;
                    }
                }
            }
        }
        if (PACK) rowbits[i] = mybits;
    }
    if (PACK) {
        // The wave's bits of its four rows are 4 x 128 bytes of the bit plane.  A 4 x 4 transposition
        // of 16-bit items inside every quad of lanes (two DPP exchanges, no LDS: this kernel runs
        // beside the vote kernel, whose atomics keep the LDS queue full) leaves lane 4q + r with the
        // 64 bits of row r, chunk columns 4q .. 4q+3, so that ONE store instruction writes everything:
        // every row gets 16 lanes x 8 contiguous bytes.
        static_assert(ROWS_PER_LANE == 4, "quad transposition");
        const int lane = threadIdx.x;
        const u32 p01 = rowbits[0] | (rowbits[1] << 16), p23 = rowbits[2] | (rowbits[3] << 16);
        const u32 o01 = (u32)__builtin_amdgcn_mov_dpp((int)p01, 0xB1, 0xf, 0xf, true);  // lane ^ 1
        const u32 o23 = (u32)__builtin_amdgcn_mov_dpp((int)p23, 0xB1, 0xf, 0xf, true);
        const u32 sel = (lane & 1) ? 0x03020706u : 0x05040100u;  // odd lanes keep the odd rows
        const u32 q01 = __builtin_amdgcn_perm(o01, p01, sel), q23 = __builtin_amdgcn_perm(o23, p23, sel);
        const bool upper = (lane & 2) != 0;  // lanes 2, 3 of a quad end up with rows 2, 3
        const u32 send = upper ? q01 : q23;
        const u32 recv = (u32)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);  // lane ^ 2
        const uint2 v = make_uint2(upper ? recv : q01, upper ? q23 : recv);
        const int r = lane & 3, c0 = (lane >> 2) * 4;
        const int yy = y0 + r, cc = blockIdx.x * 64 + c0;
        if (yy < h && cc < cpr) *(uint2*)((u8*)(bits + (size_t)b * bsw + (size_t)yy * bpw) + (size_t)cc * 2) = v;
    }
}

// The same conversion with ONE 16-pixel chunk per lane and consecutive lanes on consecutive chunks of the image in
// row-major order (gray_flat_chunk in gray_core.hpp): measured 7 % faster than the 4-row shape above on this traffic
// pattern (scripts/micro/gray_shapes.hip: 5.6 against 5.2 TB/s).
template <bool THRESH, bool PACK>
__global__ __launch_bounds__(256) void k_bgr2gray_flat(GrayArgs G, ClearList clr)
{
    if (clr.n[0] | clr.n[1] | clr.n[2]) {  // (wave-uniform) the first workgroup zeroes the scratch of the stages that follow
        if (blockIdx.x == 0 && blockIdx.z == 0) {
            for (int a = 0; a < 3; a++)
                for (u32 i = threadIdx.x; i < clr.n[a]; i += 256) clr.p[a][i] = 0;
        }
    }
    const u32 ci = blockIdx.x * 256u + threadIdx.x;
    if (ci >= G.nchunks) return;
    gray_flat_chunk<THRESH, PACK>(G, (int)blockIdx.z, ci);
}

// gray -> BGR (replicate) : 16 px per lane, 16 B in, 48 B out.
__global__ __launch_bounds__(256) void k_gray2bgr(Plane gray, Plane bgr, int w, int h, int cpr, long long total)
{
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += step) {
        int cx = (int)(c % cpr);
        long long t = c / cpr;
        int y = (int)(t % h), b = (int)(t / h);
        const u8* s = gray.row(b, y) + (size_t)cx * 16;
        u8* o = bgr.row(b, y) + (size_t)cx * 48;
        if (cx * 16 + 16 <= w) {
            uint4 a = *(const uint4*)s;
            u32 d[4] = {a.x, a.y, a.z, a.w};
            u32 r[12];
#pragma unroll
            for (int k = 0; k < 12; k++) r[k] = 0;
#pragma unroll
            for (int k = 0; k < 48; k++) r[k >> 2] |= byte_of(d, k / 3) << (8 * (k & 3));
            uint4* o4 = (uint4*)o;
            o4[0] = make_uint4(r[0], r[1], r[2], r[3]);
            o4[1] = make_uint4(r[4], r[5], r[6], r[7]);
            o4[2] = make_uint4(r[8], r[9], r[10], r[11]);
        } else {
            int n = w - cx * 16;
            for (int i = 0; i < n; i++) o[3 * i] = o[3 * i + 1] = o[3 * i + 2] = s[i];
        }
    }
}

// swap channels 0 and 2 of an interleaved 3-channel image.
__global__ __launch_bounds__(256) void k_swap_rb(Plane src, Plane dst, int w, int h, int cpr, long long total)
{
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += step) {
        int cx = (int)(c % cpr);
        long long t = c / cpr;
        int y = (int)(t % h), b = (int)(t / h);
        const u8* s = src.row(b, y) + (size_t)cx * 48;
        u8* o = dst.row(b, y) + (size_t)cx * 48;
        if (cx * 16 + 16 <= w) {
            const uint4* s4 = (const uint4*)s;
            uint4 a0 = s4[0], a1 = s4[1], a2 = s4[2];
            u32 d[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
            u32 r[12];
#pragma unroll
            for (int k = 0; k < 12; k++) r[k] = 0;
#pragma unroll
            for (int k = 0; k < 48; k++) r[k >> 2] |= byte_of(d, k - (k % 3) + (2 - k % 3)) << (8 * (k & 3));
            uint4* o4 = (uint4*)o;
            o4[0] = make_uint4(r[0], r[1], r[2], r[3]);
            o4[1] = make_uint4(r[4], r[5], r[6], r[7]);
            o4[2] = make_uint4(r[8], r[9], r[10], r[11]);
        } else {
            int n = w - cx * 16;
            for (int i = 0; i < n; i++) {
                u8 c0 = s[3 * i], c1 = s[3 * i + 1], c2 = s[3 * i + 2];
                o[3 * i] = c2; o[3 * i + 1] = c1; o[3 * i + 2] = c0;
            }
        }
    }
}

// Element-wise byte kernels over `rb` bytes per row (rb = width * channels).
// MODE 0: dst = src > t ? hi : lo (t read from `tdev[b]` when non-null)
// MODE 1: dst = ~src        MODE 2: dst = src | src2
template <int MODE>
__global__ __launch_bounds__(256) void k_bytewise(Plane src, Plane src2, Plane dst, int rb, int h, int cpr,
                                                  long long total, int ithresh, const int32_t* __restrict__ tdev,
                                                  u32 v_above, u32 v_below)
{
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += step) {
        int cx = (int)(c % cpr);
        long long t = c / cpr;
        int y = (int)(t % h), b = (int)(t / h);
        // reading / writing the whole 16-byte chunk is safe: pitch is a multiple of 16
        uint4 a = *(const uint4*)(src.row(b, y) + (size_t)cx * 16);
        u32 d[4] = {a.x, a.y, a.z, a.w}, r[4];
        if (MODE == 0) {
            int th = tdev ? tdev[b] : ithresh;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 o = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int v = (d[k] >> (8 * j)) & 0xff;
                    o |= (v > th ? v_above : v_below) << (8 * j);
                }
                r[k] = o;
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = ~d[k];
        } else {
            uint4 e = *(const uint4*)(src2.row(b, y) + (size_t)cx * 16);
            r[0] = d[0] | e.x; r[1] = d[1] | e.y; r[2] = d[2] | e.z; r[3] = d[3] | e.w;
        }
        *(uint4*)(dst.row(b, y) + (size_t)cx * 16) = make_uint4(r[0], r[1], r[2], r[3]);
    }
}

// 256-bin histogram.  A wave whose 1024 pixels all hold one value (a white page is ~97 % one value) is detected
// with a wave vote and costs one LDS atomic.  Otherwise every lane adds its 16 bytes one by one into one of sixteen
// copies of the histogram that the workgroup keeps in LDS, copy = lane mod 16, stored interleaved (word = 16 value +
// copy): lanes that hold the same value -- on a noisy scan most of a wave does -- then hit sixteen different words
// in sixteen different banks instead of one address sixty-four times (atomics of one instruction that share an
// address are served one after the other), and neighbouring values fall into the other half of the banks.
constexpr int HIST_COPIES = 16;
// div_k != 0: the image has fewer than 2^28 chunks and row = chunk / cpr is a multiplication (div_m = floor(2^div_k / cpr) + 1)
template <int NT>
__global__ __launch_bounds__(NT) void k_hist(Plane src, int w, int h, int cpr, long long chunks_per_img,
                                             u32* __restrict__ hist, u32 div_m, int div_k)
{
    __shared__ u32 lh[256 * HIST_COPIES];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 256 * HIST_COPIES; i += NT) lh[i] = 0;
    __syncthreads();
    char* const mine = (char*)lh + (threadIdx.x & (HIST_COPIES - 1)) * 4;  // this lane's copy
    const long long step = (long long)gridDim.x * blockDim.x;
    const u8* const img = src.p + (size_t)b * src.stride;
    // the chunk of the next trip is loaded before the current one is counted
    auto fetch = [&](long long c, int& n) -> uint4 {
        n = 0;
        if (c >= chunks_per_img) return make_uint4(0, 0, 0, 0);
        int cx, y;
        if (div_k) {
            y = (int)(((unsigned long long)(u32)c * div_m) >> div_k);
            cx = (int)((u32)c - (u32)y * (u32)cpr);
        } else {
            cx = (int)(c % cpr);
            y = (int)(c / cpr);
        }
        n = min(16, w - cx * 16);
        return *(const uint4*)(img + (size_t)y * src.pitch + (size_t)cx * 16);
    };
    // the chunks of the next two trips are under way while the current one is counted
    int n_next, n_next2;
    uint4 a_next = fetch((long long)blockIdx.x * blockDim.x + threadIdx.x, n_next);
    uint4 a_next2 = fetch((long long)blockIdx.x * blockDim.x + step + threadIdx.x, n_next2);
    for (long long base = (long long)blockIdx.x * blockDim.x; base < chunks_per_img; base += step) {
        const uint4 a = a_next;
        const int n = n_next;
        a_next = a_next2;
        n_next = n_next2;
        a_next2 = fetch(base + 2 * step + threadIdx.x, n_next2);
        const u32 d[4] = {a.x, a.y, a.z, a.w};
        const u32 first = d[0] & 0xffu;
        const u32 rep = first * 0x01010101u;
        const bool same = n == 16 && d[0] == rep && d[1] == rep && d[2] == rep && d[3] == rep;
        const u32 lead = __shfl(first, 0);
        if (__all(same && first == lead)) {
            if ((threadIdx.x & 63) == 0) atomicAdd(&lh[lead * HIST_COPIES], 1024u);
        } else if (n == 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const u32 v = (d[i >> 2] >> (8 * (i & 3))) & 0xffu;
                atomicAdd((u32*)(mine + v * (HIST_COPIES * 4)), 1u);
            }
        } else {
            for (int i = 0; i < n; i++) {
                const u32 v = (d[i >> 2] >> (8 * (i & 3))) & 0xffu;
                atomicAdd((u32*)(mine + v * (HIST_COPIES * 4)), 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        u32 s = 0;
#pragma unroll
        for (int k = 0; k < HIST_COPIES; k++) s += lh[threadIdx.x * HIST_COPIES + k];
        if (s) atomicAdd(&hist[(size_t)b * 256 + threadIdx.x], s);
    }
}

// Otsu's threshold from a 256-bin histogram, with the results of OpenCV's sequential scan in double precision.  One
// workgroup of 256 threads per image.  The mean is a sum of integers below 2^53 and therefore exact in any order
// (wave reduction).  What is sequential in the scan is split by what it depends on: the running class weight q1 is a
// chain of additions (one lane, first); whether a level is skipped, the product k * p_k and the reciprocal of q1
// refined by two Newton steps depend on q1 alone and are made side by side for all levels; the class mean mu1 is then
// a chain of five dependent operations per level (multiply, add, quotient estimate, two fused corrections) -- the
// same operations, in the same order and with the same roundings, as the multiply, add and division of the scan.
// The between-class variance of all levels is evaluated side by side with the expression of the scan, and the
// first level that reaches the largest variance wins, as the scan's strict comparison does.
__global__ __launch_bounds__(256) void k_otsu_scan(const u32* __restrict__ hist, long long npix, int32_t* __restrict__ out)
{
    __shared__ double s_p[256], s_q1[256], s_mu1[256], s_red[256], s_rcp[256], s_kp[256];
    __shared__ u32 s_valid[8];
    const int b = blockIdx.x, i = threadIdx.x;
    const double scale = 1.0 / (double)npix;
    const u32 hi_ = hist[(size_t)b * 256 + i];
    s_p[i] = hi_ * scale;
    s_red[i] = i * (double)hi_;
    if (i < 8) s_valid[i] = 0;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (i < o) s_red[i] += s_red[i + o];
        __syncthreads();
    }
    const double mu = s_red[0] * scale;
    // (1) the running class weight: 256 dependent additions, one lane
    if (i == 0) {
        double q1 = 0;
#pragma unroll 8
        for (int k = 0; k < 256; k++) {
            q1 += s_p[k];
            s_q1[k] = q1;
        }
    }
    __syncthreads();
    // (2) side by side for all levels: is the level skipped, the product k * p_k, and the part of the division by q1
    // that does not depend on the numerator.  The division is the sequence the compiler emits for a double
    // quotient (reciprocal estimate, two Newton steps, quotient, one correction); its scaling and fix-up steps
    // are the identity here because q1 lies in [2^-23, 1) and the numerator in [0, 255].
    const double q1_i = s_q1[i], q2_i = 1. - q1_i;
    const bool ok_i = !(fmin(q1_i, q2_i) < FLT_EPSILON || fmax(q1_i, q2_i) > 1. - FLT_EPSILON);
    {
        const unsigned long long okm = __ballot(ok_i);
        if ((i & 63) == 0) {
            s_valid[(i >> 6) * 2] = (u32)okm;
            s_valid[(i >> 6) * 2 + 1] = (u32)(okm >> 32);
        }
        double r = __builtin_amdgcn_rcp(q1_i);
        r = __builtin_fma(r, __builtin_fma(-q1_i, r, 1.0), r);
        r = __builtin_fma(r, __builtin_fma(-q1_i, r, 1.0), r);
        s_rcp[i] = r;
        s_kp[i] = i * s_p[i];
    }
    __syncthreads();
    // (3) the class mean: the only chain that needs the previous level's result -- five dependent operations a level.
    // q1 never falls, so the levels the scan does not skip form one interval [lo, hi]; before it mu1 is still 0
    // (0 times anything finite), behind it nothing is looked at any more.
    if (i == 0) {
        int lo = 256, hi = -1;
        for (int wd = 0; wd < 8; wd++) {
            const u32 vb = s_valid[wd];
            if (vb) {
                if (lo == 256) lo = wd * 32 + __ffs(vb) - 1;
                hi = wd * 32 + 31 - __clz(vb);
            }
        }
        double mu1 = 0, q1_prev = lo > 0 && lo < 256 ? s_q1[lo - 1] : 0.0;
#pragma unroll 4
        for (int k = lo; k <= hi; k++) {
            const double q1 = s_q1[k], r = s_rcp[k], kp = s_kp[k];
            const double num = mu1 * q1_prev + kp;
            const double q0 = num * r;
            mu1 = __builtin_fma(__builtin_fma(-q1, q0, num), r, q0);
            q1_prev = q1;
            s_mu1[k] = mu1;
        }
    }
    __syncthreads();
    double sigma = -1.0;  // (levels the scan skips never win: the scan starts from a largest variance of 0)
    if ((s_valid[i >> 5] >> (i & 31)) & 1u) {
        const double q1 = s_q1[i], mu1 = s_mu1[i];
        const double q2 = 1. - q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double d = mu1 - mu2;
        sigma = q1 * q2 * d * d;
    }
    __syncthreads();
    s_red[i] = sigma;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (i < o) s_red[i] = fmax(s_red[i], s_red[i + o]);
        __syncthreads();
    }
    const double best = s_red[0];
    __syncthreads();
    // first level with the largest variance (none when no variance is above 0: level 0, as the scan's initial value)
    s_red[i] = (best > 0 && sigma == best) ? (double)i : 256.0;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (i < o) s_red[i] = fmin(s_red[i], s_red[i + o]);
        __syncthreads();
    }
    if (i == 0) out[b] = s_red[0] < 256.0 ? (int)s_red[0] : 0;
}

int sat_u8_round(double v)
{
    long r = lrint(v);
    return (int)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

}  // namespace

int launch_bgr2gray(Plane bgr, Plane gray, Plane bin, u32* bits, size_t bpw, size_t bsw, int w, int h, int batch,
                    int ithresh, int maxval, int inv, hipStream_t st, ClearList clr)
{
    int cpr = cdiv(w, 16);
    cpr += cpr & 1;
    if (cpr == 0 || h == 0 || batch == 0) return IPA_OK;
    // normalised so that "above" means gray > ithresh with ithresh in 0 .. 255 (a negative threshold, every pixel
    // above, becomes "never above" with the two values swapped)
    const ThreshK tk = thresh_consts(ithresh, inv ? 0u : (u32)maxval, inv ? (u32)maxval : 0u);
    const u32 va = tk.va, vb = tk.vb, k7 = tk.k7, kb = tk.kb;
    if (ithresh < 0 || ithresh > 255) ithresh = 255;
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        Plane sb = bgr, gb = gray, tb = bin;
        sb.p += (size_t)b0 * bgr.stride;
        gb.p += (size_t)b0 * gray.stride;
        if (tb.p) tb.p += (size_t)b0 * bin.stride;
        u32* bb = bits ? bits + (size_t)b0 * bsw : nullptr;
        const dim3 grid = rows_grid(cpr, h, nb), block = rows_block();
        static int flat = -1;  // IPA_GRAY_FLAT=0: the 4-row shape for every launch (A/B)
        if (flat < 0) {
            const char* e = getenv("IPA_GRAY_FLAT");
            flat = e ? atoi(e) : 1;
        }
        GrayArgs G;
        if (flat && gray_args_make(&G, sb, gb, tb, bb, bpw, bsw, w, h, ithresh, va, vb, k7, kb)) {
            const dim3 fgrid((G.nchunks + 255u) / 256u, 1, (unsigned)nb);
            if (bin.p && bits) hipLaunchKernelGGL((k_bgr2gray_flat<true, true>), fgrid, dim3(256), 0, st, G, clr);
            else if (bin.p) hipLaunchKernelGGL((k_bgr2gray_flat<true, false>), fgrid, dim3(256), 0, st, G, clr);
            else hipLaunchKernelGGL((k_bgr2gray_flat<false, false>), fgrid, dim3(256), 0, st, G, clr);
            IPA_LAUNCH_CHECK();
            clr.n[0] = clr.n[1] = clr.n[2] = 0;  // (only the first launch clears)
            continue;
        }
        // the bit plane holds "pixel is non-zero in the thresholded plane"
        if (bin.p && bits)
            hipLaunchKernelGGL((k_bgr2gray<true, true>), grid, block, 0, st, sb, gb, tb, bb, bpw, bsw, w, h, cpr, ithresh, va, vb, k7, kb, clr);
        else if (bin.p)
            hipLaunchKernelGGL((k_bgr2gray<true, false>), grid, block, 0, st, sb, gb, tb, bb, bpw, bsw, w, h, cpr, ithresh, va, vb, k7, kb, clr);
        else
            hipLaunchKernelGGL((k_bgr2gray<false, false>), grid, block, 0, st, sb, gb, tb, bb, bpw, bsw, w, h, cpr, ithresh, va, vb, k7, kb, clr);
        IPA_LAUNCH_CHECK();
        clr.n[0] = clr.n[1] = clr.n[2] = 0;  // (only the first launch clears)
    }
    return IPA_OK;
}

bool gray_job_make(GrayJob* J, Plane bgr, Plane gray, Plane bin, u32* bits, size_t bpw, size_t bsw, int w, int h, int batch,
                   int ithresh, int maxval, int inv)
{
    if (batch <= 0 || !bin.p || !bits) return false;
    const ThreshK tk = thresh_consts(ithresh, inv ? 0u : (u32)maxval, inv ? (u32)maxval : 0u);
    if (ithresh < 0 || ithresh > 255) ithresh = 255;
    if (!gray_args_make(&J->G, bgr, gray, bin, bits, bpw, bsw, w, h, ithresh, tk.va, tk.vb, tk.k7, tk.kb)) return false;
    J->units_per_image = (J->G.nchunks + 63u) / 64u;
    const unsigned long long total = (unsigned long long)J->units_per_image * (unsigned)batch;
    if (total >= 0xffffffffull) return false;
    J->total_units = (u32)total;
    return true;
}

#define CHECK_DIMS()                                                                          \
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size"); \
    if (width == 0 || height == 0 || batch == 0) return IPA_OK

IPA_EXPORT int ipa_bgr2gray_u8(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride, uint8_t* gray,
                               size_t gray_pitch, size_t gray_stride, int width, int height, int batch, void* stream)
{
    CHECK_DIMS();
    Plane s = mkplane(bgr, bgr_pitch, bgr_stride), d = mkplane(gray, gray_pitch, gray_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width * 3, batch) && plane_ok(d, (size_t)width, batch), IPA_ERR_ALIGN,
                "bgr2gray: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    return launch_bgr2gray(s, d, mkplane(nullptr, 0, 0), nullptr, 0, 0, width, height, batch, 0, 0, 0,
                           (hipStream_t)stream);
}

namespace {
// numpy's slice bounds for a[i:j] on an axis of length n
__device__ __forceinline__ int slice_bound(int i, int n)
{
    if (i < 0) i += n;
    return i < 0 ? 0 : (i > n ? n : i);
}

// image[y1:y2, x1:x2] = colour for a list of rectangles (one colour: the order of overlapping
// rectangles does not matter).  blockIdx.x = rectangle, blockIdx.y = stripe of its rows.
__global__ __launch_bounds__(256) void k_fill_rects(Plane img, int w, int h, int batch, const int32_t* __restrict__ rects,
                                                    u32 c0, u32 c1, u32 c2)
{
    const int32_t* rc = rects + (size_t)blockIdx.x * 5;
    const int b = rc[0];
    if (b < 0 || b >= batch) return;
    const int x1 = slice_bound(rc[1], w), y1 = slice_bound(rc[2], h), x2 = slice_bound(rc[3], w), y2 = slice_bound(rc[4], h);
    if (x1 >= x2 || y1 >= y2) return;
    const int nbytes = (x2 - x1) * 3;
    const u32 col[3] = {c0, c1, c2};
    for (int y = y1 + (int)blockIdx.y; y < y2; y += (int)gridDim.y) {
        u8* row = img.row(b, y) + (size_t)x1 * 3;
        for (int i = threadIdx.x; i < nbytes; i += 256) row[i] = (u8)col[i % 3];
    }
}
}  // namespace

IPA_EXPORT int ipa_fill_rects_c3_u8(uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride, int width, int height, int batch,
                                    const int32_t* rects, int count, int c0, int c1, int c2, void* stream)
{
    CHECK_DIMS();
    IPA_REQUIRE(count >= 0, IPA_ERR_ARG, "fill_rects: negative count");
    if (count == 0) return IPA_OK;
    IPA_REQUIRE(rects != nullptr, IPA_ERR_ARG, "fill_rects: rects is NULL");
    IPA_REQUIRE((unsigned)c0 < 256u && (unsigned)c1 < 256u && (unsigned)c2 < 256u, IPA_ERR_ARG, "fill_rects: colour out of range");
    Plane s = mkplane(bgr, bgr_pitch, bgr_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width * 3, batch), IPA_ERR_ALIGN,
                "fill_rects: pointer/pitch must be 16-byte aligned and pitch >= row bytes");
    for (int r0 = 0; r0 < count; r0 += 65535 * 32) {
        const int n = count - r0 < 65535 * 32 ? count - r0 : 65535 * 32;
        hipLaunchKernelGGL(k_fill_rects, dim3((unsigned)n, 16), dim3(256), 0, (hipStream_t)stream, s, width, height, batch,
                           rects + (size_t)r0 * 5, (u32)c0, (u32)c1, (u32)c2);
        IPA_LAUNCH_CHECK();
    }
    return IPA_OK;
}

IPA_EXPORT int ipa_gray_threshold_u8(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride, uint8_t* gray,
                                     uint8_t* thresh, size_t plane_pitch, size_t plane_stride, int width, int height,
                                     int batch, double thresh_val, double maxval, int type, void* stream)
{
    CHECK_DIMS();
    IPA_REQUIRE(type == IPA_THRESH_BINARY || type == IPA_THRESH_BINARY_INV, IPA_ERR_ARG, "gray_threshold: unknown threshold type");
    Plane s = mkplane(bgr, bgr_pitch, bgr_stride), g = mkplane(gray, plane_pitch, plane_stride),
          t = mkplane(thresh, plane_pitch, plane_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width * 3, batch) && plane_ok(g, (size_t)width, batch) && plane_ok(t, (size_t)width, batch),
                IPA_ERR_ALIGN, "gray_threshold: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    // cv::threshold on 8-bit data: floor(thresh), maxval rounded and saturated
    double ft = floor(thresh_val);
    const int it = ft < -1 ? -1 : (ft > 256 ? 256 : (int)ft);
    long mv = lrint(maxval);
    const int imax = mv < 0 ? 0 : (mv > 255 ? 255 : (int)mv);
    return launch_bgr2gray(s, g, t, nullptr, 0, 0, width, height, batch, it, imax, type == IPA_THRESH_BINARY_INV,
                           (hipStream_t)stream);
}

IPA_EXPORT int ipa_gray2bgr_u8(const uint8_t* gray, size_t gray_pitch, size_t gray_stride, uint8_t* bgr,
                               size_t bgr_pitch, size_t bgr_stride, int width, int height, int batch, void* stream)
{
    CHECK_DIMS();
    Plane s = mkplane(gray, gray_pitch, gray_stride), d = mkplane(bgr, bgr_pitch, bgr_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch) && plane_ok(d, (size_t)width * 3, batch), IPA_ERR_ALIGN,
                "gray2bgr: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    int cpr = cdiv(width, 16);
    long long total = (long long)cpr * height * batch;
    hipLaunchKernelGGL(k_gray2bgr, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, s, d, width,
                       height, cpr, total);
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}

IPA_EXPORT int ipa_swap_rb_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst, size_t dst_pitch,
                              size_t dst_stride, int width, int height, int batch, void* stream)
{
    CHECK_DIMS();
    Plane s = mkplane(src, src_pitch, src_stride), d = mkplane(dst, dst_pitch, dst_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width * 3, batch) && plane_ok(d, (size_t)width * 3, batch), IPA_ERR_ALIGN,
                "swap_rb: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    int cpr = cdiv(width, 16);
    long long total = (long long)cpr * height * batch;
    hipLaunchKernelGGL(k_swap_rb, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, s, d, width,
                       height, cpr, total);
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}

IPA_EXPORT int ipa_bitwise_not_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst,
                                  size_t dst_pitch, size_t dst_stride, int row_bytes, int height, int batch,
                                  void* stream)
{
    int width = row_bytes;
    CHECK_DIMS();
    Plane s = mkplane(src, src_pitch, src_stride), d = mkplane(dst, dst_pitch, dst_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)row_bytes, batch) && plane_ok(d, (size_t)row_bytes, batch), IPA_ERR_ALIGN,
                "bitwise_not: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    int cpr = cdiv(row_bytes, 16);
    long long total = (long long)cpr * height * batch;
    hipLaunchKernelGGL((k_bytewise<1>), dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, s, s, d,
                       row_bytes, height, cpr, total, 0, (const int32_t*)nullptr, 0u, 0u);
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}

IPA_EXPORT int ipa_bitwise_or_u8(const uint8_t* a, size_t a_pitch, size_t a_stride, const uint8_t* b, size_t b_pitch,
                                 size_t b_stride, uint8_t* dst, size_t dst_pitch, size_t dst_stride, int row_bytes,
                                 int height, int batch, void* stream)
{
    int width = row_bytes;
    CHECK_DIMS();
    Plane s = mkplane(a, a_pitch, a_stride), s2 = mkplane(b, b_pitch, b_stride), d = mkplane(dst, dst_pitch, dst_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)row_bytes, batch) && plane_ok(s2, (size_t)row_bytes, batch) &&
                    plane_ok(d, (size_t)row_bytes, batch),
                IPA_ERR_ALIGN, "bitwise_or: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    int cpr = cdiv(row_bytes, 16);
    long long total = (long long)cpr * height * batch;
    hipLaunchKernelGGL((k_bytewise<2>), dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, s, s2, d,
                       row_bytes, height, cpr, total, 0, (const int32_t*)nullptr, 0u, 0u);
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}

static int threshold_impl(Plane s, Plane d, int width, int height, int batch, int ithresh, const int32_t* tdev,
                          double maxval, int type, hipStream_t st)
{
    IPA_REQUIRE(type == IPA_THRESH_BINARY || type == IPA_THRESH_BINARY_INV, IPA_ERR_ARG,
                "threshold: only THRESH_BINARY / THRESH_BINARY_INV (optionally with THRESH_OTSU) are supported");
    int mv = sat_u8_round(maxval);
    u32 va = type == IPA_THRESH_BINARY ? (u32)mv : 0u, vb = type == IPA_THRESH_BINARY ? 0u : (u32)mv;
    int cpr = cdiv(width, 16);
    long long total = (long long)cpr * height * batch;
    hipLaunchKernelGGL((k_bytewise<0>), dim3(stream_grid(total, 256)), dim3(256), 0, st, s, s, d, width, height, cpr,
                       total, ithresh, tdev, va, vb);
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}

IPA_EXPORT int ipa_threshold_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst,
                                size_t dst_pitch, size_t dst_stride, int width, int height, int batch, double thresh,
                                double maxval, int type, void* stream)
{
    CHECK_DIMS();
    Plane s = mkplane(src, src_pitch, src_stride), d = mkplane(dst, dst_pitch, dst_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch) && plane_ok(d, (size_t)width, batch), IPA_ERR_ALIGN,
                "threshold: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    double f = floor(thresh);
    int ithresh = f < -1 ? -1 : (f > 256 ? 256 : (int)f);
    return threshold_impl(s, d, width, height, batch, ithresh, nullptr, maxval, type, (hipStream_t)stream);
}

static int histogram_impl(Plane s, int width, int height, int batch, u32* hist, hipStream_t st)
{
    IPA_HIP(hipMemsetAsync(hist, 0, (size_t)batch * 256 * sizeof(u32), st));
    int cpr = cdiv(width, 16);
    long long cpi = (long long)cpr * height;
    // 1024 threads per workgroup and at most 768 of them (three per CU): every workgroup ends with one global atomic
    // per non-empty bin, so the same number of waves in fewer, larger workgroups costs less (one 16384 x 16384
    // page: 1024 x 256 threads 85 us, 256 x 1024 82 us, 512 x 1024 71 us, 768 x 1024 70 us)
    static int nt_env = -1;  // IPA_HIST_THREADS: 256 or 1024 (default) threads per workgroup (A/B)
    if (nt_env < 0) {
        const char* e = getenv("IPA_HIST_THREADS");
        nt_env = e ? atoi(e) : 1024;
    }
    const int nt = nt_env == 256 ? 256 : 1024;
    unsigned gx = (unsigned)((cpi + nt * 8 - 1) / (nt * 8));
    static int cap_env = -1;  // IPA_HIST_CAP: workgroups per launch (A/B)
    if (cap_env < 0) {
        const char* e = getenv("IPA_HIST_CAP");
        cap_env = e ? atoi(e) : 0;
    }
    unsigned cap = (unsigned)((cap_env > 0 ? cap_env : (nt == 1024 ? 768 : 1024)) / (batch < 768 ? batch : 768));  // (few workgroups: every one ends with an atomic per non-empty bin on the image's 256 counters)
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        Plane sb = s;
        sb.p += (size_t)b0 * s.stride;
        u32 div_m = 0;
        int div_k = 0;
        if ((unsigned long long)cpi < (1ull << 28)) {
            int lb = 0;
            while ((1u << lb) <= (u32)cpr) lb++;  // cpr < 2^lb
            div_k = 28 + lb;
            div_m = (u32)((1ull << div_k) / (u32)cpr) + 1u;
        }
        if (nt == 1024)
            hipLaunchKernelGGL(k_hist<1024>, dim3(gx, nb), dim3(1024), 0, st, sb, width, height, cpr, cpi, hist + (size_t)b0 * 256, div_m, div_k);
        else
            hipLaunchKernelGGL(k_hist<256>, dim3(gx, nb), dim3(256), 0, st, sb, width, height, cpr, cpi, hist + (size_t)b0 * 256, div_m, div_k);
        IPA_LAUNCH_CHECK();
    }
    return IPA_OK;
}

IPA_EXPORT int ipa_histogram_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, int width, int height,
                                int batch, uint32_t* hist, void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    if (batch == 0) return IPA_OK;
    IPA_REQUIRE(hist != nullptr, IPA_ERR_ARG, "histogram: hist is NULL");
    if (width == 0 || height == 0) {
        IPA_HIP(hipMemsetAsync(hist, 0, (size_t)batch * 256 * sizeof(u32), (hipStream_t)stream));
        return IPA_OK;
    }
    Plane s = mkplane(src, src_pitch, src_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch), IPA_ERR_ALIGN,
                "histogram: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    return histogram_impl(s, width, height, batch, hist, (hipStream_t)stream);
}

IPA_EXPORT int ipa_threshold_otsu_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst,
                                     size_t dst_pitch, size_t dst_stride, int width, int height, int batch,
                                     double maxval, int type, uint32_t* hist, int32_t* otsu, void* stream)
{
    CHECK_DIMS();
    IPA_REQUIRE(hist && otsu, IPA_ERR_ARG, "threshold_otsu: hist/otsu buffers are NULL");
    Plane s = mkplane(src, src_pitch, src_stride), d = mkplane(dst, dst_pitch, dst_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch) && plane_ok(d, (size_t)width, batch), IPA_ERR_ALIGN,
                "threshold_otsu: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    hipStream_t st = (hipStream_t)stream;
    int rc = histogram_impl(s, width, height, batch, hist, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_otsu_scan, dim3((unsigned)batch), dim3(256), 0, st, hist, (long long)width * height, otsu);
    IPA_LAUNCH_CHECK();
    return threshold_impl(s, d, width, height, batch, 0, otsu, maxval, type & 7, st);
}
