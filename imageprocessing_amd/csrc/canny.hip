// Canny edge detector (aperture 3, L1 gradient magnitude), OpenCV semantics.
//
//  k_canny_nms : one workgroup = one 128x32 tile.  Gray tile + 2-pixel halo in
//                LDS -> Sobel magnitude for tile + 1 ring -> non-maximum
//                suppression -> hysteresis INSIDE the tile -> edge bytes:
//                255 edge, 0 none, WEAK for candidates the tile could not
//                resolve alone.  Tiles that still hold WEAK pixels are appended
//                to a worklist.
//  k_hyst_iter : worklist pass.  A tile re-reads its edge bytes with a 1-pixel
//                ring (seeing what the neighbours have resolved), propagates to a
//                fixed point in LDS, writes back, and queues the neighbours when
//                its rim changed.  Repeated until a pass queues nothing.
//  k_hyst_final: remaining WEAK bytes -> 0.
//
// The result does not depend on processing order: a candidate survives iff its
// 8-connected component of candidates contains a pixel above the high threshold.
#include "common.hpp"

#include <mutex>

namespace {

constexpr int TW = 128, TH = 32;
constexpr int GW = TW + 8;   // gray row bytes in LDS (4-byte halo each side, 2 used)
constexpr int GH = TH + 4;
constexpr int MW = TW + 2, MH = TH + 2;
constexpr u8 WEAK = 1;

struct CannyWs {
    u32* list0;      // tiles left with WEAK pixels by k_canny_nms
    u32* listA;      // ping-pong worklists of the hysteresis passes
    u32* listB;
    u32* mark;       // per tile: last pass number that queued it
    u8* has_weak;    // per tile
    u32* counters;   // [0] = |list0|, [1] = |listA|, [2] = |listB|
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void k_canny_nms(Plane src, Plane dst, int w, int h, int tiles_x, int tiles_y,
                                                   int low, int high, CannyWs W)
{
    __shared__ __align__(16) u8 g[GH][GW];
    __shared__ u16 mag[MH][MW + 2];
    __shared__ u8 mp[MH][MW + 2];
    const int tid = threadIdx.x;
    int tile = blockIdx.x;
    const int tx = tile % tiles_x;
    int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y, b = t2 / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;

    // gray tile with replicated borders: LDS (r, c) <-> image (y0 - 2 + r, x0 - 4 + c)
    for (int e = tid; e < GH * (GW / 4); e += 256) {
        int r = e / (GW / 4), cd = e - r * (GW / 4);
        int gy = clampi(y0 - 2 + r, 0, h - 1);
        int gx = x0 - 4 + cd * 4;
        const u8* row = src.row(b, gy);
        u32 v;
        if (gx >= 0 && gx + 4 <= w) {
            v = *(const u32*)(row + gx);
        } else {
            v = 0;
            for (int j = 0; j < 4; j++) v |= (u32)row[clampi(gx + j, 0, w - 1)] << (8 * j);
        }
        *(u32*)&g[r][cd * 4] = v;
    }
    __syncthreads();

    // magnitude for tile + 1 ring; zero outside the image
    for (int e = tid; e < MH * MW; e += 256) {
        int r = e / MW, c = e - r * MW;  // image (y0 - 1 + r, x0 - 1 + c)
        int iy = y0 - 1 + r, ix = x0 - 1 + c;
        int m = 0;
        if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
            const u8* p0 = &g[r][c + 2];  // row above, column ix-1  (LDS row = iy - (y0-2) - 1 = r)
            const u8* p1 = &g[r + 1][c + 2];
            const u8* p2 = &g[r + 2][c + 2];
            int dx = (p0[2] - p0[0]) + 2 * (p1[2] - p1[0]) + (p2[2] - p2[0]);
            int dy = (p2[0] - p0[0]) + 2 * (p2[1] - p0[1]) + (p2[2] - p0[2]);
            m = abs(dx) + abs(dy);
        }
        mag[r][c] = (u16)m;
        mp[r][c] = 1;
    }
    __syncthreads();

    // non-maximum suppression: lane = 16 consecutive pixels of one tile row
    const int lr = tid >> 3, lc0 = (tid & 7) * 16;
    bool any_weak = false, any_strong = false;
    {
        const int iy = y0 + lr;
        for (int i = 0; i < 16; i++) {
            int lc = lc0 + i, ix = x0 + lc;
            u8 v = 1;
            if (iy < h && ix < w) {
                int m = mag[lr + 1][lc + 1];
                if (m > low) {
                    const u8* p0 = &g[lr + 1][lc + 3];
                    const u8* p1 = &g[lr + 2][lc + 3];
                    const u8* p2 = &g[lr + 3][lc + 3];
                    int dx = (p0[2] - p0[0]) + 2 * (p1[2] - p1[0]) + (p2[2] - p2[0]);
                    int dy = (p2[0] - p0[0]) + 2 * (p2[1] - p0[1]) + (p2[2] - p0[2]);
                    int ax = abs(dx), ay = abs(dy) << 15;
                    int tg22x = ax * 13573;
                    bool is_max;
                    if (ay < tg22x) {
                        is_max = m > mag[lr + 1][lc] && m >= mag[lr + 1][lc + 2];
                    } else {
                        int tg67x = tg22x + (ax << 16);
                        if (ay > tg67x) {
                            is_max = m > mag[lr][lc + 1] && m >= mag[lr + 2][lc + 1];
                        } else {
                            int s = ((dx ^ dy) < 0) ? -1 : 1;
                            is_max = m > mag[lr][lc + 1 - s] && m > mag[lr + 2][lc + 1 + s];
                        }
                    }
                    if (is_max) v = m > high ? 2 : 0;
                }
            }
            mp[lr + 1][lc + 1] = v;
            any_weak |= v == 0;
            any_strong |= v == 2;
        }
    }
    // hysteresis inside the tile (only if it holds both kinds)
    int both = __syncthreads_or(any_weak) && __syncthreads_or(any_strong);
    if (both) {
        for (;;) {
            bool changed = false;
            for (int i = 0; i < 16; i++) {
                int r = lr + 1, c = lc0 + i + 1;
                if (mp[r][c] == 0) {
                    bool s = mp[r - 1][c - 1] == 2 || mp[r - 1][c] == 2 || mp[r - 1][c + 1] == 2 || mp[r][c - 1] == 2 ||
                             mp[r][c + 1] == 2 || mp[r + 1][c - 1] == 2 || mp[r + 1][c] == 2 || mp[r + 1][c + 1] == 2;
                    if (s) {
                        mp[r][c] = 2;
                        changed = true;
                    }
                }
            }
            if (!__syncthreads_or(changed)) break;
        }
    }
    // write edge bytes
    bool left_weak = false;
    {
        const int iy = y0 + lr, ix0 = x0 + lc0;
        u32 o[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++) {
            u8 v = mp[lr + 1][lc0 + i + 1];
            u32 e = v == 2 ? 255u : (v == 0 ? (u32)WEAK : 0u);
            left_weak |= v == 0;
            o[i >> 2] |= e << (8 * (i & 3));
        }
        if (iy < h && ix0 < w) *(uint4*)(dst.row(b, iy) + ix0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    int lw = __syncthreads_or(left_weak);
    if (tid == 0) {
        W.has_weak[blockIdx.x] = lw ? 1 : 0;
        W.mark[blockIdx.x] = 0;
        if (lw) {
            u32 pos = atomicAdd(&W.counters[0], 1u);
            W.list0[pos] = blockIdx.x;
        }
    }
}

// One hysteresis pass over a worklist.  `pass` >= 1 numbers the pass (used to
// queue each tile at most once per pass).
__global__ __launch_bounds__(256) void k_hyst_iter(Plane dst, int w, int h, int tiles_x, int tiles_y,
                                                   const u32* __restrict__ in_list, const u32* __restrict__ in_count,
                                                   u32* __restrict__ out_list, u32* __restrict__ out_count, u32 pass,
                                                   CannyWs W)
{
    __shared__ u8 mp[MH][MW + 2];
    const int tid = threadIdx.x;
    const u32 count = *in_count;
    const int lr = tid >> 3, lc0 = (tid & 7) * 16;
    for (u32 li = blockIdx.x; li < count; li += gridDim.x) {
        const u32 tile = in_list[li];
        const int tx = tile % tiles_x;
        u32 t2 = tile / tiles_x;
        const int ty = t2 % tiles_y, b = t2 / tiles_y;
        const int x0 = tx * TW, y0 = ty * TH;
        __syncthreads();  // previous tile's LDS fully consumed
        for (int e = tid; e < MH * MW; e += 256) {
            int r = e / MW, c = e - r * MW;
            int iy = y0 - 1 + r, ix = x0 - 1 + c;
            u8 v = 0;
            if (iy >= 0 && iy < h && ix >= 0 && ix < w) v = dst.row(b, iy)[ix];
            mp[r][c] = v;
        }
        __syncthreads();
        bool mine_changed = false, rim_changed = false;
        for (;;) {
            bool changed = false;
            for (int i = 0; i < 16; i++) {
                int r = lr + 1, c = lc0 + i + 1;
                if (mp[r][c] == WEAK) {
                    bool s = mp[r - 1][c - 1] == 255 || mp[r - 1][c] == 255 || mp[r - 1][c + 1] == 255 ||
                             mp[r][c - 1] == 255 || mp[r][c + 1] == 255 || mp[r + 1][c - 1] == 255 ||
                             mp[r + 1][c] == 255 || mp[r + 1][c + 1] == 255;
                    if (s) {
                        mp[r][c] = 255;
                        changed = true;
                        if (lr == 0 || lr == TH - 1 || c == 1 || c == TW) rim_changed = true;
                    }
                }
            }
            mine_changed |= changed;
            if (!__syncthreads_or(changed)) break;
        }
        if (mine_changed) {
            const int iy = y0 + lr, ix0 = x0 + lc0;
            if (iy < h && ix0 < w) {
                u32 o[4] = {0, 0, 0, 0};
                for (int i = 0; i < 16; i++) o[i >> 2] |= (u32)mp[lr + 1][lc0 + i + 1] << (8 * (i & 3));
                *(uint4*)(dst.row(b, iy) + ix0) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        int rim = __syncthreads_or(rim_changed);
        if (rim && tid < 9 && tid != 4) {
            int nx = tx + (tid % 3) - 1, ny = ty + (tid / 3) - 1;
            if (nx >= 0 && nx < tiles_x && ny >= 0 && ny < tiles_y) {
                u32 nt = ((u32)b * tiles_y + ny) * tiles_x + nx;
                if (W.has_weak[nt]) {
                    u32 old = atomicExch(&W.mark[nt], pass);
                    if (old != pass) {
                        u32 pos = atomicAdd(out_count, 1u);
                        out_list[pos] = nt;
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_hyst_final(Plane dst, int w, int h, int tiles_x, int tiles_y,
                                                    const u32* __restrict__ list, const u32* __restrict__ count)
{
    const u32 n = *count;
    const int lr = threadIdx.x >> 3, lc0 = (threadIdx.x & 7) * 16;
    for (u32 li = blockIdx.x; li < n; li += gridDim.x) {
        const u32 tile = list[li];
        const int tx = tile % tiles_x;
        u32 t2 = tile / tiles_x;
        const int ty = t2 % tiles_y, b = t2 / tiles_y;
        const int iy = ty * TH + lr, ix0 = tx * TW + lc0;
        if (iy >= h || ix0 >= w) continue;
        uint4* p = (uint4*)(dst.row(b, iy) + ix0);
        uint4 v = *p;
        u32 d[4] = {v.x, v.y, v.z, v.w};
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // bytes are 0, WEAK(1) or 255: keep a byte only if its top bit is set
            u32 keep = ((d[k] >> 7) & 0x01010101u) * 0xffu;
            any |= (d[k] & ~keep) != 0;
            d[k] &= keep;
        }
        if (any) *p = make_uint4(d[0], d[1], d[2], d[3]);
    }
}

std::mutex g_pin_mu;
u32* g_pinned = nullptr;  // small pinned host buffer for convergence read-backs

}  // namespace

size_t canny_ws_bytes(int w, int h, int batch)
{
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    size_t ntiles = (size_t)cdiv(w, TW) * cdiv(h, TH) * batch;
    Arena a(nullptr, 0);
    a.take(ntiles * 4);
    a.take(ntiles * 4);
    a.take(ntiles * 4);
    a.take(ntiles * 4);
    a.take(ntiles);
    a.take(64);
    return a.off + 256;
}

// Enqueues the whole detector and iterates the hysteresis to convergence.
// Synchronises `st` once per convergence check.
int run_canny(Plane src, Plane dst, int w, int h, int batch, int low, int high, void* ws, size_t ws_bytes,
              hipStream_t st)
{
    const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
    const long long ntiles_ll = (long long)tiles_x * tiles_y * batch;
    IPA_REQUIRE(ntiles_ll < 0x7fffffffLL, IPA_ERR_ARG, "Canny: too many tiles");
    IPA_REQUIRE(ws && ws_bytes >= canny_ws_bytes(w, h, batch), IPA_ERR_WORKSPACE, "Canny: workspace too small");
    const size_t ntiles = (size_t)ntiles_ll;
    Arena ar(ws, ws_bytes);
    CannyWs W;
    W.list0 = (u32*)ar.take(ntiles * 4);
    W.listA = (u32*)ar.take(ntiles * 4);
    W.listB = (u32*)ar.take(ntiles * 4);
    W.mark = (u32*)ar.take(ntiles * 4);
    W.has_weak = (u8*)ar.take(ntiles);
    W.counters = (u32*)ar.take(64);
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (!g_pinned) IPA_HIP(hipHostMalloc((void**)&g_pinned, 256, hipHostMallocDefault));
    }
    IPA_HIP(hipMemsetAsync(W.counters, 0, 64, st));
    prof_begin(PROF_CANNY_NMS, st);
    hipLaunchKernelGGL(k_canny_nms, dim3((unsigned)ntiles), dim3(256), 0, st, src, dst, w, h, tiles_x, tiles_y, low,
                       high, W);
    prof_end(PROF_CANNY_NMS, st);
    IPA_LAUNCH_CHECK();
    ProfScope ps(PROF_HYSTERESIS, st);
    const unsigned grid = (unsigned)(ntiles < 2048 ? ntiles : 2048);
    // pass p reads list (p==1: list0, else ping-pong) and writes the other ping-pong list
    u32 pass = 1;
    const u32* in_list = W.list0;
    const u32* in_cnt = W.counters + 0;
    int out_sel = 0;  // 0 -> listA/counters[1], 1 -> listB/counters[2]
    for (int guard = 0; guard < 100000; guard++) {
        u32* out_list = out_sel ? W.listB : W.listA;
        u32* out_cnt = W.counters + 1 + out_sel;
        IPA_HIP(hipMemsetAsync(out_cnt, 0, 4, st));
        hipLaunchKernelGGL(k_hyst_iter, dim3(grid), dim3(256), 0, st, dst, w, h, tiles_x, tiles_y, in_list, in_cnt,
                           out_list, out_cnt, pass, W);
        IPA_LAUNCH_CHECK();
        in_list = out_list;
        in_cnt = out_cnt;
        out_sel ^= 1;
        pass++;
        // read the size of the list just produced after the 2nd pass and then every 2 passes
        if (pass >= 3 && (pass & 1)) {
            u32 pending;
            {
                std::lock_guard<std::mutex> lk(g_pin_mu);
                IPA_HIP(hipMemcpyAsync(g_pinned, in_cnt, 4, hipMemcpyDeviceToHost, st));
                IPA_HIP(hipStreamSynchronize(st));
                pending = g_pinned[0];
            }
            if (pending == 0) break;
        }
    }
    hipLaunchKernelGGL(k_hyst_final, dim3(grid), dim3(256), 0, st, dst, w, h, tiles_x, tiles_y,
                       (const u32*)W.list0, (const u32*)(W.counters + 0));
    IPA_LAUNCH_CHECK();
    return IPA_OK;
}

IPA_EXPORT size_t ipa_canny_workspace_bytes(int width, int height, int batch)
{
    return canny_ws_bytes(width, height, batch);
}

IPA_EXPORT int ipa_canny_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst, size_t dst_pitch,
                            size_t dst_stride, int width, int height, int batch, double threshold1, double threshold2,
                            void* ws, size_t ws_bytes, void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    if (width == 0 || height == 0 || batch == 0) return IPA_OK;
    Plane s = mkplane(src, src_pitch, src_stride), d = mkplane(dst, dst_pitch, dst_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch) && plane_ok(d, (size_t)width, batch), IPA_ERR_ALIGN,
                "Canny: pointers/pitches must be 16-byte aligned and pitch >= row bytes");
    if (threshold1 > threshold2) {
        double t = threshold1;
        threshold1 = threshold2;
        threshold2 = t;
    }
    double fl = floor(threshold1), fh = floor(threshold2);
    int low = fl < -1 ? -1 : (fl > 1e6 ? 1000000 : (int)fl);
    int high = fh < -1 ? -1 : (fh > 1e6 ? 1000000 : (int)fh);
    return run_canny(s, d, width, height, batch, low, high, ws, ws_bytes, (hipStream_t)stream);
}
