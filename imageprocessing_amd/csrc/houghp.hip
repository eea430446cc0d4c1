// cv2.HoughLinesP (progressive probabilistic Hough transform) on the device: lineDetection.py:12-19,
// "New Version/comparison.py":14, newVersion3/detectHorizontalVerticalLines.py:64 and three more
// scripts of the reference.
//
// The algorithm is sequential by construction: the non-zero pixels are visited in a pseudo-random
// order (cv::RNG, multiply-with-carry), each visit reads and changes the accumulator and the pixel
// mask, and a found segment removes its pixels before the next visit.  One WORKGROUP therefore owns one
// image and runs the visits in order (k_hp_run_mw: its waves all run the same control flow and split the
// bulk work; k_hp_run: the one-wave form); inside a wave the 64 lanes share the work (pixels of a batch of
// draws, angles of a vote, steps of a line walk), and a batch of images runs as many workgroups side by side.
//
//   k_hp_rowcount / k_hp_scan / k_hp_fill   non-zero pixels as a list in raster order (the order
//                                           the random draws index into) from the 1-bit plane
//   k_hp_run                                 the visits; the bit plane is the mask
#include "common.hpp"

#include <math.h>

int launch_pack_bits(Plane src, u32* bits, size_t bpw, size_t bsw, int w, int h, int batch, int32_t* vmax, hipStream_t st);

namespace {

constexpr int HP_MAX_ANGLES = 448;
struct HpTrig {
    float c[HP_MAX_ANGLES];
    float s[HP_MAX_ANGLES];
};

struct HpArgs {
    u32* bits;  // [batch][h][bpw] mask, one bit per pixel
    size_t bpw, bsw;
    u32* rowcnt;   // [batch][h]
    u32* rowoff;   // [batch][h + 1]
    int32_t* total;  // [batch] non-zero pixels, -1: more than cap
    u32* nz;       // [batch][cap] (y << 16) | x
    u32 cap;
    int32_t* accum;  // [batch][numangle][numrho]
    size_t accum_stride;
    int w, h, numangle, numrho, threshold, line_length, line_gap;
    int32_t* lines;  // [batch][max_lines][4]
    int max_lines;
    int32_t* counts;  // [batch] segments found (may exceed max_lines), -1: list capacity exceeded
    long long* stats;  // debug: per-phase wall-clock ticks of image 0 (or NULL)
};

// one wave per row
__global__ __launch_bounds__(256) void k_hp_rowcount(HpArgs A)
{
    const int b = blockIdx.y;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= A.h) return;
    const int lane = threadIdx.x & 63;
    const int words = (A.w + 31) >> 5;
    const u32* row = A.bits + (size_t)b * A.bsw + (size_t)y * A.bpw;
    u32 n = 0;
    for (int j = lane; j < words; j += 64) n += __popc(row[j]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += (u32)__shfl_xor((int)n, o);
    if (lane == 0) A.rowcnt[(size_t)b * A.h + y] = n;
}

__global__ __launch_bounds__(1024) void k_hp_scan(HpArgs A)
{
    __shared__ u32 s_wave[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const u32* cnt = A.rowcnt + (size_t)b * A.h;
    u32* off = A.rowoff + (size_t)b * (A.h + 1);
    u32 base = 0;
    for (int y0 = 0; y0 < A.h; y0 += 1024) {
        const int y = y0 + tid;
        const u32 c = y < A.h ? cnt[y] : 0u;
        u32 total;
        const u32 incl = block_scan_1024(c, s_wave, total);
        if (y < A.h) off[y] = base + incl - c;
        base += total;
    }
    if (tid == 0) {
        off[A.h] = base;
        A.total[b] = base > A.cap ? -1 : (int32_t)base;
    }
}

// one wave per row: the row's non-zero pixels, left to right
__global__ __launch_bounds__(256) void k_hp_fill(HpArgs A)
{
    const int b = blockIdx.y;
    if (A.total[b] < 0) return;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= A.h) return;
    const int lane = threadIdx.x & 63;
    const int words = (A.w + 31) >> 5;
    const u32* row = A.bits + (size_t)b * A.bsw + (size_t)y * A.bpw;
    u32* out = A.nz + (size_t)b * A.cap;
    u32 k = A.rowoff[(size_t)b * (A.h + 1) + y];
    for (int j0 = 0; j0 < words; j0 += 64) {
        const int j = j0 + lane;
        u32 t = j < words ? row[j] : 0u;
        const u32 c = __popc(t);
        u32 scan = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 v = (u32)__shfl_up((int)scan, o);
            if (lane >= o) scan += v;
        }
        u32 pos = k + scan - c;
        while (t) {
            const int i = __ffs(t) - 1;
            t &= t - 1;
            out[pos++] = ((u32)y << 16) | (u32)(32 * j + i);
        }
        k += (u32)__shfl((int)scan, 63);
    }
}

__device__ __forceinline__ u32 ld_l2(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_l2(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int hp_rho(int x, int y, float c, float s)
{
    return __float2int_rn(__fadd_rn(__fmul_rn((float)x, c), __fmul_rn((float)y, s)));
}

// One wave per image.  Everything that steers control flow is wave-uniform.
//
// The visits are taken 64 at a time where that provably changes nothing:
//  * the order of the draws does not depend on the image (the list shrinks by one entry per draw
//    whatever the pixel turns out to be), so the 64 list operations of a batch are done by the 64
//    lanes at once unless two of them touch the same list position (then the batch runs one draw
//    at a time);
//  * pixels whose mask bit is already clear are skipped by all lanes at once;
//  * votes that trigger no segment commute, so the live pixels of a batch vote side by side
//    (lane = pixel); if any bin they touched ends at or above the threshold the votes are taken
//    back and the batch is replayed one visit at a time (lanes = angles / steps of the line walk).
__global__ __launch_bounds__(64) void k_hp_run(HpArgs A, HpTrig Tk)
{
    // the tables are indexed per lane: from LDS, not from the kernel-argument segment (host memory)
    __shared__ HpTrig T;
    for (int n = threadIdx.x; n < HP_MAX_ANGLES; n += 64) {
        T.c[n] = Tk.c[n];
        T.s[n] = Tk.s[n];
    }
    __syncthreads();
    const int b = blockIdx.x, lane = threadIdx.x;
    int count = A.total[b];
    if (count < 0) {
        if (lane == 0) A.counts[b] = -1;
        return;
    }
    u32* const bits = A.bits + (size_t)b * A.bsw;
    u32* const nz = A.nz + (size_t)b * A.cap;
    int32_t* const acc = A.accum + (size_t)b * A.accum_stride;
    int32_t* const lines = A.lines + (size_t)b * A.max_lines * 4;
    const int w = A.w, h = A.h, numangle = A.numangle, numrho = A.numrho, half = (numrho - 1) / 2;
    const int shift = 16;
    auto mask_at = [&](int i1, int j1) -> u32 { return (ld_l2(&bits[(size_t)i1 * A.bpw + (j1 >> 5)]) >> (j1 & 31)) & 1u; };
    int nlines = 0;

    // one visit of pixel (row i, column j), exactly as cv2 orders its steps
    auto visit = [&](const int i, const int j) {
        if (!mask_at(i, j)) return;  // already part of a segment

        // vote; the winner is the first angle whose bin holds the largest count
        int best_val = -1, best_n = 0;
        for (int n0 = lane; n0 < numangle; n0 += 256) {  // four atomics in flight per lane
            int val[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n = n0 + 64 * q;
                val[q] = -1;
                if (n < numangle) val[q] = atomicAdd(&acc[(size_t)n * numrho + hp_rho(j, i, T.c[n], T.s[n]) + half], 1) + 1;
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (val[q] > best_val) {
                    best_val = val[q];
                    best_n = n0 + 64 * q;
                }
        }
        unsigned long long key = ((unsigned long long)(u32)(best_val + 1) << 32) | (u32)(0x7fffffff - best_n);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o);
            key = other > key ? other : key;
        }
        const int max_val = (int)(key >> 32) - 1, max_n = 0x7fffffff - (int)(u32)key;
        if (max_val < A.threshold) return;

        // direction of the line and its 16.16 fixed-point step
        const float a = -T.s[max_n], bb = T.c[max_n];
        int x0 = j, y0 = i, dx0, dy0;
        bool xflag;
        if (fabsf(a) > fabsf(bb)) {
            xflag = true;
            dx0 = a > 0 ? 1 : -1;
            dy0 = __float2int_rn(__fdiv_rn(__fmul_rn(bb, 65536.0f), fabsf(a)));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = false;
            dy0 = bb > 0 ? 1 : -1;
            dx0 = __float2int_rn(__fdiv_rn(__fmul_rn(a, 65536.0f), fabsf(bb)));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        auto pos_at = [&](int k, int t, int& i1, int& j1) {
            const int x = x0 + (k ? -dx0 : dx0) * t, y = y0 + (k ? -dy0 : dy0) * t;
            j1 = xflag ? x : x >> shift;
            i1 = xflag ? y >> shift : y;
        };

        // first walk: how far the segment reaches in each direction (step index of its last pixel);
        // WCH chunks of 64 steps per round trip to the mask
        constexpr int WCH = 4;
        int last_step[2] = {0, 0};
        for (int k = 0; k < 2; k++) {
            int gap = 0;
            bool stop = false;
            for (int base = 0; !stop; base += 64 * WCH) {
                unsigned long long M[WCH], O[WCH];
                u32 word[WCH];
                int bitpos[WCH];
                bool inb[WCH];
#pragma unroll
                for (int q = 0; q < WCH; q++) {
                    int i1, j1;
                    pos_at(k, base + 64 * q + lane, i1, j1);
                    inb[q] = j1 >= 0 && j1 < w && i1 >= 0 && i1 < h;
                    bitpos[q] = j1 & 31;
                    word[q] = inb[q] ? ld_l2(&bits[(size_t)i1 * A.bpw + (j1 >> 5)]) : 0u;
                }
#pragma unroll
                for (int q = 0; q < WCH; q++) {
                    M[q] = __ballot(inb[q] && ((word[q] >> bitpos[q]) & 1u));
                    O[q] = __ballot(!inb[q]);
                }
#pragma unroll
                for (int q = 0; q < WCH; q++) {
                    if (stop) break;
                    if (O[q] == 0 && M[q] == ~0ULL) {
                        gap = 0;
                        last_step[k] = base + 64 * q + 63;
                        continue;
                    }
                    for (int t = 0; t < 64; t++) {
                        if ((O[q] >> t) & 1ULL) {
                            stop = true;
                            break;
                        }
                        if ((M[q] >> t) & 1ULL) {
                            gap = 0;
                            last_step[k] = base + 64 * q + t;
                        } else if (++gap > A.line_gap) {
                            stop = true;
                            break;
                        }
                    }
                }
            }
        }
        int ex[2], ey[2];
        pos_at(0, last_step[0], ey[0], ex[0]);
        pos_at(1, last_step[1], ey[1], ex[1]);
        const bool good = abs(ex[1] - ex[0]) >= A.line_length || abs(ey[1] - ey[0]) >= A.line_length;

        // second walk: the segment's pixels leave the mask and, for a kept segment, take their votes
        // back.  Step 0 (the drawn pixel itself) belongs to both directions; it is handled once.
        for (int k = 0; k < 2; k++) {
            for (int base = k; base <= last_step[k]; base += 64 * WCH) {
                unsigned long long M[WCH];
                u32 word[WCH];
                u32* wp[WCH];
                int bitpos[WCH];
                bool in_seg[WCH];
#pragma unroll
                for (int q = 0; q < WCH; q++) {
                    int i1, j1;
                    const int t = base + 64 * q + lane;
                    pos_at(k, t, i1, j1);
                    in_seg[q] = t <= last_step[k];
                    bitpos[q] = j1 & 31;
                    wp[q] = &bits[(size_t)i1 * A.bpw + (j1 >> 5)];
                    word[q] = in_seg[q] ? ld_l2(wp[q]) : 0u;
                }
#pragma unroll
                for (int q = 0; q < WCH; q++) {
                    const bool on = in_seg[q] && ((word[q] >> bitpos[q]) & 1u);
                    M[q] = __ballot(on);
                    if (on) atomicAnd(wp[q], ~(1u << bitpos[q]));
                }
                if (good) {
#pragma unroll
                    for (int q = 0; q < WCH; q++) {
                        unsigned long long m = M[q];
                        while (m) {
                            const int tt = base + 64 * q + (__ffsll((long long)m) - 1);
                            m &= m - 1;
                            int iu, ju;
                            pos_at(k, tt, iu, ju);
                            for (int n = lane; n < numangle; n += 64)
                                atomicAdd(&acc[(size_t)n * numrho + hp_rho(ju, iu, T.c[n], T.s[n]) + half], -1);
                        }
                    }
                }
            }
        }
        __threadfence_block();  // the next visit reads what this one changed
        if (good) {
            if (lane == 0 && nlines < A.max_lines) {
                lines[4 * nlines + 0] = ex[0];
                lines[4 * nlines + 1] = ey[0];
                lines[4 * nlines + 2] = ex[1];
                lines[4 * nlines + 3] = ey[1];
            }
            nlines++;
        }
    };

    const unsigned long long RNG_COEFF = 4164903690ULL;
    unsigned long long rng = 0xffffffffffffffffULL;
    long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tk = wall_clock64();
    auto lap = [&](int k) { const long long now = wall_clock64(); st[k] += now - tk; tk = now; };
    while (count > 0) {
        st[5]++;
        int K = count < 64 ? count : 64;
        // the next K draws: lane t keeps the t-th
        unsigned long long x = rng;
        u32 mine = 0;
        for (int t = 0; t < K; t++) {
            x = (unsigned long long)(u32)x * RNG_COEFF + (u32)(x >> 32);
            if (lane == t) mine = (u32)x;
        }
        const u32 idx = lane < K ? mine % (u32)(count - lane) : 0xffffffffu;
        // Two of the K list operations may touch one position (a drawn position drawn again, or a
        // drawn position that is the last entry of a later draw): the batch ends before the later of
        // the two.  Draw 0 never is the later one.
        u32 later = 64;  // per lane: the smallest later partner of a clash this lane is the earlier part of
        for (int u = 1; u < K; u++) {
            const u32 iu = (u32)__builtin_amdgcn_readlane((int)idx, u);
            if (lane < u && (idx == iu || idx == (u32)(count - 1 - u))) later = min(later, (u32)u);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) later = min(later, (u32)__shfl_xor((int)later, o));
        if ((int)later < K) {
            K = (int)later;
            st[6]++;
            // the generator restarts behind draw K - 1
            x = rng;
            for (int t = 0; t < K; t++) x = (unsigned long long)(u32)x * RNG_COEFF + (u32)(x >> 32);
        }
        const bool act = lane < K;
        lap(0);
        // draw without replacement: the last entry takes the drawn entry's place
        u32 p = 0;
        if (act) {
            p = ld_l2(&nz[idx]);
            const u32 last = ld_l2(&nz[count - 1 - lane]);
            st_l2(&nz[idx], last);
        }
        __threadfence_block();
        const int pi = (int)(p >> 16), pj = (int)(p & 0xffffu);
        lap(2);
        // Votes that trigger no segment commute: the live pixels from `start` on vote side by side
        // (lane = pixel).  A pixel none of whose bins ends at or above the threshold cannot have
        // triggered a segment in cv2's order either; the first pixel that might is visited on its own
        // (after the votes from it on have been taken back), and the rest of the batch starts over.
        auto vote_all = [&](bool on, int delta) {
            if (on)
                for (int n = 0; n < numangle; n++)
                    atomicAdd(&acc[(size_t)n * numrho + hp_rho(pj, pi, T.c[n], T.s[n]) + half], delta);
        };
        for (int start = 0; start < K;) {
            const bool live = act && lane >= start && mask_at(pi, pj);
            if (!__any(live)) break;
            vote_all(live, 1);
            __threadfence_block();
            int mx = 0;
            if (live) {
                int n = 0;
                for (; n + 30 <= numangle; n += 30) {  // thirty loads in flight per lane
                    int v[30];
#pragma unroll
                    for (int q = 0; q < 30; q++)
                        v[q] = (int)ld_l2((const u32*)&acc[(size_t)(n + q) * numrho + hp_rho(pj, pi, T.c[n + q], T.s[n + q]) + half]);
#pragma unroll
                    for (int q = 0; q < 30; q++) mx = v[q] > mx ? v[q] : mx;
                }
                for (; n < numangle; n++) {
                    const int v = (int)ld_l2((const u32*)&acc[(size_t)n * numrho + hp_rho(pj, pi, T.c[n], T.s[n]) + half]);
                    mx = v > mx ? v : mx;
                }
            }
            const unsigned long long hot = __ballot(live && mx >= A.threshold);
            lap(3);
            if (!hot) break;
            st[7]++;
            const int t1 = __ffsll((long long)hot) - 1;
            vote_all(live && lane >= t1, -1);
            __threadfence_block();
            const u32 pt = (u32)__builtin_amdgcn_readlane((int)p, t1);
            visit((int)(pt >> 16), (int)(pt & 0xffffu));
            start = t1 + 1;
            lap(4);
        }
        count -= K;
        rng = x;
    }
    if (A.stats && b == 0 && lane == 0)
        for (int k = 0; k < 8; k++) A.stats[k] = st[k];
    if (lane == 0) A.counts[b] = nlines;
}

// The same visits by a workgroup of NW waves per image.  Every wave runs the same (wave-uniform) control flow on the
// same data, so all of them take the same decisions; what is bulk work inside a batch or a visit is split between
// them: the angles of the speculative votes and of their maximum scan, the 64-step chunks of a line walk, the
// pixels of a found segment that take their votes back.  The list of pixels is touched by wave 0 only.  Waves meet
// at barriers that also wait for the memory operations issued before them.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_hp_run_mw(HpArgs A, HpTrig Tk)
{
    __shared__ HpTrig T;  // (indexed per lane: from LDS, not from the kernel-argument segment)
    for (int n = threadIdx.x; n < HP_MAX_ANGLES; n += 64 * NW) {
        T.c[n] = Tk.c[n];
        T.s[n] = Tk.s[n];
    }
    __syncthreads();
    __shared__ u32 s_p[64];
    __shared__ u32 s_later[NW];
    __shared__ int s_mx[NW][64];
    __shared__ unsigned long long s_key[NW];
    constexpr int WCH = 4;  // chunks of 64 walk steps per wave and round trip to the mask
    __shared__ unsigned long long s_M[2][NW * WCH], s_O[2][NW * WCH];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int count = A.total[b];
    if (count < 0) {
        if (threadIdx.x == 0) A.counts[b] = -1;
        return;
    }
    u32* const bits = A.bits + (size_t)b * A.bsw;
    u32* const nz = A.nz + (size_t)b * A.cap;
    int32_t* const acc = A.accum + (size_t)b * A.accum_stride;
    int32_t* const lines = A.lines + (size_t)b * A.max_lines * 4;
    const int w = A.w, h = A.h, numangle = A.numangle, numrho = A.numrho, half = (numrho - 1) / 2;
    const int shift = 16;
    auto mask_at = [&](int i1, int j1) -> u32 { return (ld_l2(&bits[(size_t)i1 * A.bpw + (j1 >> 5)]) >> (j1 & 31)) & 1u; };
    // everything written before (memory, LDS) is seen by every wave behind it
    auto sync = [&]() {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    };
    int nlines = 0;
    int par = 0;  // which copy of s_M / s_O the next walk round writes
    long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tk = wall_clock64();
    auto lap = [&](int q) { const long long now = wall_clock64(); st[q] += now - tk; tk = now; };

    auto visit = [&](const int i, const int j) {
        if (!mask_at(i, j)) return;  // already part of a segment
        // vote; the winner is the first angle whose bin holds the largest count
        int best_val = -1, best_n = 0;
        for (int n = wv * 64 + lane; n < numangle; n += 64 * NW) {
            const int val = atomicAdd(&acc[(size_t)n * numrho + hp_rho(j, i, T.c[n], T.s[n]) + half], 1) + 1;
            if (val > best_val) {
                best_val = val;
                best_n = n;
            }
        }
        unsigned long long key = ((unsigned long long)(u32)(best_val + 1) << 32) | (u32)(0x7fffffff - best_n);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o);
            key = other > key ? other : key;
        }
        if (lane == 0) s_key[wv] = key;
        sync();
        key = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) key = s_key[q] > key ? s_key[q] : key;
        const int max_val = (int)(key >> 32) - 1, max_n = 0x7fffffff - (int)(u32)key;
        lap(4);
        if (max_val < A.threshold) return;

        const float a = -T.s[max_n], bb = T.c[max_n];
        int x0 = j, y0 = i, dx0, dy0;
        bool xflag;
        if (fabsf(a) > fabsf(bb)) {
            xflag = true;
            dx0 = a > 0 ? 1 : -1;
            dy0 = __float2int_rn(__fdiv_rn(__fmul_rn(bb, 65536.0f), fabsf(a)));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = false;
            dy0 = bb > 0 ? 1 : -1;
            dx0 = __float2int_rn(__fdiv_rn(__fmul_rn(a, 65536.0f), fabsf(bb)));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        auto pos_at = [&](int k, int t, int& i1, int& j1) {
            const int x = x0 + (k ? -dx0 : dx0) * t, y = y0 + (k ? -dy0 : dy0) * t;
            j1 = xflag ? x : x >> shift;
            i1 = xflag ? y >> shift : y;
        };
        // first walk, both directions at once: a round looks at 64 * WCH * NW / 2 steps of each (the first half of
        // the waves one way, the second half the other, WCH chunks of 64 steps per wave); every wave then reads the
        // chunks in order
        constexpr int HW = NW / 2, CPR = HW * WCH;  // chunks per direction and round
        int last_step[2] = {0, 0};
        {
            int gap[2] = {0, 0};
            bool stop[2] = {false, false};
            const int kd = wv >= HW ? 1 : 0, cq = wv - kd * HW;
            for (int base = 0; !(stop[0] && stop[1]); base += 64 * CPR) {
                {
                    u32 word[WCH];
                    int bitpos[WCH];
                    bool inb[WCH];
#pragma unroll
                    for (int m = 0; m < WCH; m++) {
                        int i1, j1;
                        pos_at(kd, base + 64 * (cq + HW * m) + lane, i1, j1);
                        inb[m] = j1 >= 0 && j1 < w && i1 >= 0 && i1 < h;
                        bitpos[m] = j1 & 31;
                        word[m] = inb[m] ? ld_l2(&bits[(size_t)i1 * A.bpw + (j1 >> 5)]) : 0u;
                    }
#pragma unroll
                    for (int m = 0; m < WCH; m++) {
                        const unsigned long long M = __ballot(inb[m] && ((word[m] >> bitpos[m]) & 1u)), O = __ballot(!inb[m]);
                        if (lane == 0) {
                            s_M[par][kd * CPR + cq + HW * m] = M;
                            s_O[par][kd * CPR + cq + HW * m] = O;
                        }
                    }
                }
                sync();
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++) {
                    for (int q = 0; q < CPR && !stop[k2]; q++) {
                        const unsigned long long M = s_M[par][k2 * CPR + q], O = s_O[par][k2 * CPR + q];
                        if (O == 0 && M == ~0ULL) {
                            gap[k2] = 0;
                            last_step[k2] = base + 64 * q + 63;
                            continue;
                        }
                        for (int t = 0; t < 64; t++) {
                            if ((O >> t) & 1ULL) {
                                stop[k2] = true;
                                break;
                            }
                            if ((M >> t) & 1ULL) {
                                gap[k2] = 0;
                                last_step[k2] = base + 64 * q + t;
                            } else if (++gap[k2] > A.line_gap) {
                                stop[k2] = true;
                                break;
                            }
                        }
                    }
                }
                par ^= 1;
            }
        }
        lap(1);
        int ex[2], ey[2];
        pos_at(0, last_step[0], ey[0], ex[0]);
        pos_at(1, last_step[1], ey[1], ex[1]);
        const bool good = abs(ex[1] - ex[0]) >= A.line_length || abs(ey[1] - ey[0]) >= A.line_length;
        // second walk: the segment's pixels leave the mask and, for a kept segment, take their votes back; a wave
        // takes every NW-th chunk of 64 steps.  Step 0 belongs to both directions; it is handled once.
        for (int k = 0; k < 2; k++) {
            for (int base = k; base <= last_step[k]; base += 64 * NW) {
                int i1, j1;
                const int t = base + 64 * wv + lane;
                pos_at(k, t, i1, j1);
                const bool in_seg = t <= last_step[k];
                u32* const wp = &bits[(size_t)i1 * A.bpw + (j1 >> 5)];
                const u32 word = in_seg ? ld_l2(wp) : 0u;
                const bool on = in_seg && ((word >> (j1 & 31)) & 1u);
                unsigned long long m = __ballot(on);
                if (on) atomicAnd(wp, ~(1u << (j1 & 31)));
                if (good && m) {
                    // Lane = angle, the chunk's pixels one after the other: neighbouring pixels of a segment fall into
                    // the same bin of the angles near its normal, and a lane takes such a run back with one atomic
                    // (lane = pixel sends the pixels of all waves to the same few addresses in L2, where they queue
                    // up).  Bins may go below zero, as in cv2: pixels that have not voted yet leave as well.
                    for (int g = 0; g < numangle; g += 64) {
                        const int n = g + lane;
                        const bool av = n < numangle;
                        const float cn = av ? T.c[n] : 0.f, sn = av ? T.s[n] : 0.f;
                        int32_t* const rowp = acc + (size_t)(av ? n : 0) * numrho + half;
                        int cur = 0, cnt = 0;
                        unsigned long long mm = m;
                        while (mm) {
                            const int tt = base + 64 * wv + (__ffsll((long long)mm) - 1);
                            mm &= mm - 1;
                            int iu, ju;
                            pos_at(k, tt, iu, ju);
                            const int rr = hp_rho(ju, iu, cn, sn);
                            if (av && cnt != 0 && rr != cur) {
                                atomicAdd(&rowp[cur], -cnt);
                                cnt = 0;
                            }
                            cur = rr;
                            cnt++;
                        }
                        if (av) atomicAdd(&rowp[cur], -cnt);
                    }
                }
            }
        }
        sync();  // the next visit reads what this one changed
        lap(6);
        if (good) {
            if (threadIdx.x == 0 && nlines < A.max_lines) {
                lines[4 * nlines + 0] = ex[0];
                lines[4 * nlines + 1] = ey[0];
                lines[4 * nlines + 2] = ex[1];
                lines[4 * nlines + 3] = ey[1];
            }
            nlines++;
        }
    };

    const unsigned long long RNG_COEFF = 4164903690ULL;
    unsigned long long rng = 0xffffffffffffffffULL;
    while (count > 0) {
        st[5]++;
        int K = count < 64 ? count : 64;
        // the next K draws: lane t keeps the t-th (every wave computes the same)
        unsigned long long x = rng;
        u32 mine = 0;
        for (int t = 0; t < K; t++) {
            x = (unsigned long long)(u32)x * RNG_COEFF + (u32)(x >> 32);
            if (lane == t) mine = (u32)x;
        }
        const u32 idx = lane < K ? mine % (u32)(count - lane) : 0xffffffffu;
        u32 later = 64;
        for (int u = 1 + wv; u < K; u += NW) {  // (this wave's share of the later partners)
            const u32 iu = (u32)__builtin_amdgcn_readlane((int)idx, u);
            if (lane < u && (idx == iu || idx == (u32)(count - 1 - u))) later = min(later, (u32)u);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) later = min(later, (u32)__shfl_xor((int)later, o));
        if (lane == 0) s_later[wv] = later;
        sync();
#pragma unroll
        for (int q = 0; q < NW; q++) later = min(later, s_later[q]);
        if ((int)later < K) {
            K = (int)later;
            x = rng;
            for (int t = 0; t < K; t++) x = (unsigned long long)(u32)x * RNG_COEFF + (u32)(x >> 32);
        }
        const bool act = lane < K;
        lap(0);
        // draw without replacement (wave 0): the last entry takes the drawn entry's place
        if (wv == 0) {
            u32 p0 = 0;
            if (act) {
                p0 = ld_l2(&nz[idx]);
                const u32 last = ld_l2(&nz[count - 1 - lane]);
                st_l2(&nz[idx], last);
            }
            s_p[lane] = p0;
        }
        sync();
        const u32 p = s_p[lane];
        const int pi = (int)(p >> 16), pj = (int)(p & 0xffffu);
        lap(2);
        // this wave's share of the angles; returns the largest count one of its votes produced
        auto vote_all = [&](bool on, int delta) -> int {
            int mx = 0;
            if (on) {
                for (int n0 = wv; n0 < numangle; n0 += 8 * NW) {  // eight atomics in flight per lane
                    int v[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int n = n0 + q * NW;
                        v[q] = n < numangle ? atomicAdd(&acc[(size_t)n * numrho + hp_rho(pj, pi, T.c[n], T.s[n]) + half], delta) + delta : 0;
                    }
#pragma unroll
                    for (int q = 0; q < 8; q++) mx = v[q] > mx ? v[q] : mx;
                }
            }
            return mx;
        };
        auto unvote_all = [&](bool on) {
            if (on)
                for (int n = wv; n < numangle; n += NW)
                    atomicAdd(&acc[(size_t)n * numrho + hp_rho(pj, pi, T.c[n], T.s[n]) + half], -1);
        };
        for (int start = 0; start < K;) {
            const bool live = act && lane >= start && mask_at(pi, pj);
            if (!__any(live)) break;
            // The counts the votes return stand in for the final ones: the values one bin hands out are consecutive,
            // so if the votes of the lanes up to t bring a bin to the threshold in cv2's order, at least one of THOSE
            // lanes is handed a value at or above it, whatever the order the hardware serves them in: the first
            // flagged lane is never behind the first lane that triggers.
            int mx = vote_all(live, 1);
            s_mx[wv][lane] = mx;
            sync();
#pragma unroll
            for (int q = 0; q < NW; q++) mx = max(mx, s_mx[q][lane]);
            const unsigned long long hot = __ballot(live && mx >= A.threshold);
            lap(3);
            if (!hot) break;
            st[7]++;
            const int t1 = __ffsll((long long)hot) - 1;
            unvote_all(live && lane >= t1);
            sync();
            const u32 pt = (u32)__builtin_amdgcn_readlane((int)p, t1);
            visit((int)(pt >> 16), (int)(pt & 0xffffu));
            start = t1 + 1;
            lap(4);
        }
        count -= K;
        rng = x;
    }
    if (A.stats && b == 0 && threadIdx.x == 0)
        for (int q = 0; q < 8; q++) A.stats[q] = st[q];
    if (threadIdx.x == 0) A.counts[b] = nlines;
}

void hp_dims(int w, int h, double rho, double theta, int* numangle, int* numrho)
{
    const float frho = (float)rho, ftheta = (float)theta;
    int na = (int)floor((M_PI - 0.0) / ftheta) + 1;
    if (na > 1 && fabs(M_PI - (na - 1) * ftheta) < ftheta / 2) --na;
    *numangle = na;
    *numrho = (int)lrintf(((w + h) * 2 + 1) / frho);
}

u32 hp_cap(int w, int h, int max_points)
{
    const long long full = (long long)w * h;
    if (max_points > 0 && max_points < full) return (u32)max_points;
    return (u32)(full > 0x7fffffffLL ? 0x7fffffffLL : full);
}

struct HpLayout {
    size_t bits, vmax, rowcnt, rowoff, total, nz, accum, accum_stride;
};
size_t hp_layout(HpLayout& L, int w, int h, int batch, int numangle, int numrho, u32 cap, void* ws, size_t ws_bytes, u8** ptrs)
{
    Arena a(ws, ws_bytes);
    L.accum_stride = align_up((size_t)numangle * numrho, 64);
    const size_t sizes[7] = {bits_pitch_words(w) * 4 * (size_t)h * batch, sizeof(int32_t) * (size_t)batch,
                             sizeof(u32) * (size_t)h * batch,             sizeof(u32) * ((size_t)h + 1) * batch,
                             sizeof(int32_t) * (size_t)batch,             sizeof(u32) * (size_t)cap * batch,
                             sizeof(int32_t) * L.accum_stride * batch};
    for (int i = 0; i < 7; i++) {
        u8* p = (u8*)a.take(sizes[i]);
        if (ptrs) ptrs[i] = p;
    }
    return a.off + 256;
}

}  // namespace

IPA_EXPORT int ipa_hough_lines_p_dims(int width, int height, double rho, double theta, int* numangle, int* numrho)
{
    IPA_REQUIRE(width >= 0 && height >= 0, IPA_ERR_ARG, "negative image size");
    IPA_REQUIRE(rho > 0 && theta > 0, IPA_ERR_ARG, "HoughLinesP: rho and theta must be positive");
    IPA_REQUIRE(numangle && numrho, IPA_ERR_ARG, "HoughLinesP: NULL output");
    hp_dims(width, height, rho, theta, numangle, numrho);
    return IPA_OK;
}

IPA_EXPORT size_t ipa_hough_lines_p_workspace_bytes(int width, int height, int batch, double rho, double theta, int max_points)
{
    if (width <= 0 || height <= 0 || batch <= 0 || !(rho > 0) || !(theta > 0)) return 0;
    int na, nr;
    hp_dims(width, height, rho, theta, &na, &nr);
    if (na < 1 || nr < 1) return 0;
    HpLayout L;
    return hp_layout(L, width, height, batch, na, nr, hp_cap(width, height, max_points), nullptr, 0, nullptr);
}

IPA_EXPORT int ipa_hough_lines_p_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, int width, int height, int batch,
                                    double rho, double theta, int threshold, int min_line_length, int max_line_gap,
                                    int max_points, int32_t* lines, int max_lines, int32_t* counts, void* ws, size_t ws_bytes,
                                    void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0 && max_lines >= 0, IPA_ERR_ARG, "HoughLinesP: negative size");
    IPA_REQUIRE(rho > 0 && theta > 0, IPA_ERR_ARG, "HoughLinesP: rho and theta must be positive");
    if (batch == 0) return IPA_OK;
    IPA_REQUIRE(counts != nullptr && (lines != nullptr || max_lines == 0), IPA_ERR_ARG, "HoughLinesP: NULL output");
    hipStream_t st = (hipStream_t)stream;
    if (width == 0 || height == 0) {
        IPA_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)batch, st));
        return IPA_OK;
    }
    IPA_REQUIRE(width <= 65535 && height <= 65535, IPA_ERR_ARG, "HoughLinesP: image larger than 65535 pixels on a side");
    int na, nr;
    hp_dims(width, height, rho, theta, &na, &nr);
    IPA_REQUIRE(na >= 1 && na <= HP_MAX_ANGLES, IPA_ERR_ARG, "HoughLinesP: %d angles (theta too small; at most %d supported)", na,
                HP_MAX_ANGLES);
    IPA_REQUIRE(nr >= 1 && (long long)na * nr < 0x7fffffffLL, IPA_ERR_ARG, "HoughLinesP: accumulator too large");
    Plane s = mkplane(src, src_pitch, src_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch), IPA_ERR_ALIGN,
                "HoughLinesP: pointer/pitch must be 16-byte aligned and pitch >= row bytes");
    IPA_REQUIRE(ws && ws_bytes >= ipa_hough_lines_p_workspace_bytes(width, height, batch, rho, theta, max_points),
                IPA_ERR_WORKSPACE, "HoughLinesP: workspace too small");
    HpLayout L;
    u8* ptr[7];
    const u32 cap = hp_cap(width, height, max_points);
    hp_layout(L, width, height, batch, na, nr, cap, ws, ws_bytes, ptr);
    HpArgs A;
    A.bits = (u32*)ptr[0];
    A.bpw = bits_pitch_words(width);
    A.bsw = A.bpw * (size_t)height;
    int32_t* vmax = (int32_t*)ptr[1];
    A.rowcnt = (u32*)ptr[2];
    A.rowoff = (u32*)ptr[3];
    A.total = (int32_t*)ptr[4];
    A.nz = (u32*)ptr[5];
    A.cap = cap;
    A.accum = (int32_t*)ptr[6];
    A.accum_stride = L.accum_stride;
    A.w = width; A.h = height; A.numangle = na; A.numrho = nr;
    A.threshold = threshold; A.line_length = min_line_length; A.line_gap = max_line_gap;
    A.lines = lines; A.max_lines = max_lines; A.counts = counts;
    A.stats = nullptr;
    if (getenv("IPA_HP_STATS")) {
        static long long* dbg = nullptr;
        if (!dbg) IPA_HIP(hipHostMalloc((void**)&dbg, 64, 0));
        A.stats = dbg;
        static long long* prev = nullptr;
        // multi-wave kernel: [1] = first walks, [4] = votes of the visits, [6] = second walks (10 ns ticks);
        // single-wave kernel: [1] unused, [4] = whole visits, [6] = batches cut short by a clash
        if (prev) fprintf(stderr, "hp stats (10 ns ticks): draws %lld [1] %lld list %lld spec votes %lld [4] %lld | batches %lld [6] %lld replays %lld\n", prev[0], prev[1], prev[2], prev[3], prev[4], prev[5], prev[6], prev[7]);
        prev = dbg;
    }
    HpTrig T;
    {
        // cv::HoughLinesProbabilistic: trigtab[n] = (float)(cos((double)n * theta) / rho) with float rho, theta
        const float frho = (float)rho, ftheta = (float)theta;
        const float irho = 1 / frho;
        for (int n = 0; n < na; n++) {
            T.c[n] = (float)(cos((double)n * ftheta) * irho);
            T.s[n] = (float)(sin((double)n * ftheta) * irho);
        }
        for (int n = na; n < HP_MAX_ANGLES; n++) T.c[n] = T.s[n] = 0.f;
    }
    int rc = launch_pack_bits(s, A.bits, A.bpw, A.bsw, width, height, batch, vmax, st);
    if (rc) return rc;
    IPA_HIP(hipMemsetAsync(A.accum, 0, sizeof(int32_t) * L.accum_stride * (size_t)batch, st));
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        HpArgs B = A;
        B.bits += (size_t)b0 * A.bsw;
        B.rowcnt += (size_t)b0 * height;
        B.rowoff += (size_t)b0 * (height + 1);
        B.total += b0;
        B.nz += (size_t)b0 * cap;
        B.accum += (size_t)b0 * L.accum_stride;
        B.lines = lines ? lines + (size_t)b0 * max_lines * 4 : nullptr;
        B.counts += b0;
        const dim3 rows((unsigned)cdiv(height, 4), (unsigned)nb);
        hipLaunchKernelGGL(k_hp_rowcount, rows, dim3(256), 0, st, B);
        hipLaunchKernelGGL(k_hp_scan, dim3((unsigned)nb), dim3(1024), 0, st, B);
        hipLaunchKernelGGL(k_hp_fill, rows, dim3(256), 0, st, B);
        static int hp_waves = -1;  // IPA_HP_WAVES: waves per image (1 = the single-wave kernel, 16 = default)
        if (hp_waves < 0) {
            const char* e = getenv("IPA_HP_WAVES");
            hp_waves = e ? atoi(e) : 16;
        }
        if (hp_waves >= 16)
            hipLaunchKernelGGL(k_hp_run_mw<16>, dim3((unsigned)nb), dim3(1024), 0, st, B, T);
        else if (hp_waves >= 4)
            hipLaunchKernelGGL(k_hp_run_mw<4>, dim3((unsigned)nb), dim3(256), 0, st, B, T);
        else
            hipLaunchKernelGGL(k_hp_run, dim3((unsigned)nb), dim3(64), 0, st, B, T);
        IPA_LAUNCH_CHECK();
    }
    return IPA_OK;
}
