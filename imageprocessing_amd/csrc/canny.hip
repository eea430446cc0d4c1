// Canny edge detector (aperture 3, L1 gradient magnitude), OpenCV semantics.
//
//  k_canny_nms : one workgroup = one 128x32 tile.  Gray tile + 2-pixel halo in
//                LDS -> Sobel magnitude for tile + 1 ring -> non-maximum
//                suppression -> edge bytes: 255 above the high threshold, WEAK
//                for candidates between the thresholds, 0 otherwise.  Tiles that
//                hold WEAK pixels are flagged.
//  k_hyst_dfs  : hysteresis.  One workgroup owns one image (or one region of a
//                huge image), scans the flagged tiles for WEAK pixels that touch
//                an edge pixel, promotes them with an atomic OR and walks the
//                chain depth-first from there (private LDS stack per lane).
//                Every promotion is an atomic read-modify-write and values only
//                ever go WEAK -> 255, so walks may cross into other workgroups'
//                regions and stale plain reads are harmless.
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
constexpr int MH = TH + 2;               // gradient rows: tile + 1 ring
constexpr int DW = 136;                  // dwords per row of the gradient array (16-byte aligned tile start)
constexpr u8 WEAK = 1;

struct CannyWs {
    u8* has_weak;   // per tile: the tile holds WEAK pixels
    u32* again;     // set when a depth-first walk had to drop work in a multi-region image
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int grad_dx(u32 d) { return (int)(d << 16) >> 16; }
__device__ __forceinline__ int grad_dy(u32 d) { return (int)d >> 16; }
__device__ __forceinline__ int grad_mag(u32 d) { return abs(grad_dx(d)) + abs(grad_dy(d)); }

// LDS layout (one 128x32 tile):
//   g[36][136]  gray bytes, image (y0-2+r, x0-4+c), borders replicated
//   D[34][136]  Sobel gradients packed (dx & 0xffff) | (dy << 16); entry (r, cc) is image
//               (y0-1+r, x0-4+cc); zero outside the image so the magnitude there is 0
//   M[32][128]  output codes 0 / WEAK / 255
//   Q[4096]     tile-local coordinates of pixels whose magnitude exceeds the low threshold
// Phases: load gray -> Sobel (4 pixels per lane) -> threshold scan (16 pixels per lane,
// candidates queued) -> direction test + neighbour compare for queued pixels only ->
// store codes.  Only ~3 % of the pixels of a diagram are candidates, so the branchy
// non-maximum test runs on a compacted list with all lanes busy.
__global__ __launch_bounds__(256) void k_canny_nms(Plane src, Plane dst, int w, int h, int tiles_x, int tiles_y,
                                                   int low, int high, CannyWs W)
{
    __shared__ __align__(16) u8 g[GH][GW];
    __shared__ __align__(16) u32 D[MH][DW];
    __shared__ __align__(16) u8 M[TH][TW];
    __shared__ u16 Q[TH * TW];
    __shared__ u32 s_qn;
    __shared__ int s_weak;
    const int tid = threadIdx.x;
    int tile = blockIdx.x;
    const int tx = tile % tiles_x;
    int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y, b = t2 / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;

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
    ((uint4*)&M[0][0])[tid] = make_uint4(0, 0, 0, 0);
    if (tid == 0) {
        s_qn = 0;
        s_weak = 0;
    }
    __syncthreads();

    // Sobel, vertical pass first: vs = top + 2 mid + bot, vd = bot - top per column, then
    // dx = vs[x+1] - vs[x-1], dy = vd[x-1] + 2 vd[x] + vd[x+1].  One lane = 4 pixels:
    // gradient columns cc = 4q + 3 .. 4q + 6 need gray bytes 4q + 2 .. 4q + 7.
    for (int e = tid; e < MH * 33; e += 256) {
        const int r = e / 33, q = e - r * 33;
        const u32* g0 = (const u32*)&g[r][4 * q];
        const u32* g1 = (const u32*)&g[r + 1][4 * q];
        const u32* g2 = (const u32*)&g[r + 2][4 * q];
        const uint2 t = make_uint2(g0[0], g0[1]);
        const uint2 m = make_uint2(g1[0], g1[1]);
        const uint2 bt = make_uint2(g2[0], g2[1]);
        int vs[6], vd[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int k = i + 2;  // byte index within the 8 loaded bytes
            const u32 tw = k < 4 ? t.x : t.y, mw = k < 4 ? m.x : m.y, bw = k < 4 ? bt.x : bt.y;
            const int tb = (tw >> (8 * (k & 3))) & 0xff, mb = (mw >> (8 * (k & 3))) & 0xff,
                      bb = (bw >> (8 * (k & 3))) & 0xff;
            vs[i] = tb + 2 * mb + bb;
            vd[i] = bb - tb;
        }
        const int iy = y0 - 1 + r;
        const bool row_in = iy >= 0 && iy < h;
        u32 o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int dx = vs[j + 2] - vs[j];
            const int dy = vd[j] + 2 * vd[j + 1] + vd[j + 2];
            const int ix = x0 - 4 + 4 * q + 3 + j;
            const bool in = row_in && ix >= 0 && ix < w;
            o[j] = in ? (((u32)dx & 0xffffu) | ((u32)dy << 16)) : 0u;
        }
        // cc = 4q + 3 + j: not 16-byte aligned, four dword stores
        u32* drow = &D[r][4 * q + 3];
        drow[0] = o[0]; drow[1] = o[1]; drow[2] = o[2]; drow[3] = o[3];
    }
    __syncthreads();

    // threshold scan: lane = 16 consecutive pixels of one tile row
    const int lr = tid >> 3, lc0 = (tid & 7) * 16;
    {
        const int iy = y0 + lr;
        u32 cand = 0;
        if (iy < h) {
            const uint4* dp = (const uint4*)&D[lr + 1][lc0 + 4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 v = dp[k];
                const u32 d4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (grad_mag(d4[j]) > low && x0 + lc0 + 4 * k + j < w) cand |= 1u << (4 * k + j);
            }
        }
        if (cand) {
            u32 pos = atomicAdd(&s_qn, (u32)__popc(cand));
            while (cand) {
                const int i = __ffs(cand) - 1;
                cand &= cand - 1;
                Q[pos++] = (u16)((lr << 7) | (lc0 + i));
            }
        }
    }
    __syncthreads();

    // non-maximum suppression on the queued pixels
    const u32 qn = s_qn;
    bool weak = false;
    for (u32 i = tid; i < qn; i += 256) {
        const int e = Q[i];
        const int r = e >> 7, c = e & 127;
        const u32* ctr = &D[r + 1][c + 4];
        const u32 d = ctr[0];
        const int dx = grad_dx(d), dy = grad_dy(d);
        const int m = abs(dx) + abs(dy);
        const int ax = abs(dx), ay = abs(dy) << 15;
        const int tg22x = ax * 13573;
        bool is_max;
        if (ay < tg22x) {
            is_max = m > grad_mag(ctr[-1]) && m >= grad_mag(ctr[1]);
        } else {
            const int tg67x = tg22x + (ax << 16);
            if (ay > tg67x) {
                is_max = m > grad_mag(ctr[-DW]) && m >= grad_mag(ctr[DW]);
            } else {
                const int s = ((dx ^ dy) < 0) ? -1 : 1;
                is_max = m > grad_mag(ctr[-DW - s]) && m > grad_mag(ctr[DW + s]);
            }
        }
        if (is_max) {
            const bool strong = m > high;
            M[r][c] = strong ? (u8)255 : WEAK;
            weak |= !strong;
        }
    }
    if (weak) s_weak = 1;
    __syncthreads();

    {
        const int iy = y0 + lr, ix0 = x0 + lc0;
        if (iy < h && ix0 < w) *(uint4*)(dst.row(b, iy) + ix0) = ((const uint4*)&M[0][0])[tid];
    }
    if (tid == 0) W.has_weak[blockIdx.x] = s_weak ? 1 : 0;
}

constexpr int DFS_THREADS = 512;
constexpr int DFS_DEPTH = 16;
constexpr int DFS_QCAP = 2048;
constexpr int DFS_SCAN_STEPS = 4;
constexpr int DFS_MAX_TILES = 4096;  // tiles per region

// L2-served load (bypasses the per-CU L1) of the aligned word holding byte (x) of a row
__device__ __forceinline__ u32 load_word_l2(const u8* row, int x)
{
    return __hip_atomic_load((const u32*)(row + (x & ~3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32 byte_at(const u8* row, int x) { return (load_word_l2(row, x) >> (8 * (x & 3))) & 0xffu; }

// WEAK -> 255 with an atomic OR; true iff this lane made the change
__device__ __forceinline__ bool promote(u8* row, int x)
{
    const u32 sh = 8u * (u32)(x & 3);
    u32 old = atomicOr((u32*)(row + (x & ~3)), 0xffu << sh);
    return ((old >> sh) & 0xffu) == WEAK;
}

// The 3x3 neighbourhood of (x, y) as three 24-bit values (bytes x-1, x, x+1 of rows y-1,
// y, y+1; 0 outside the image).  All word loads are issued before the first use.
__device__ __forceinline__ void load_3x3(const Plane& dst, int b, int x, int y, int w, int h, u32 out[3])
{
    const int xl = x - 1, xr = x + 1;
    u32 lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int ny = y + k - 1;
        lo[k] = hi[k] = 0;
        if (ny >= 0 && ny < h) {
            const u8* row = dst.row(b, ny);
            // lo = word of x-1, hi = word of min(x+1, w-1): between them they always hold x
            const int xc = xr < w ? xr : x;
            if (xl >= 0) lo[k] = load_word_l2(row, xl);
            hi[k] = ((xc & ~3) == (xl & ~3) && xl >= 0) ? lo[k] : load_word_l2(row, xc);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        u32 l = xl >= 0 ? (lo[k] >> (8 * (xl & 3))) & 0xffu : 0u;
        u32 r = xr < w ? (hi[k] >> (8 * (xr & 3))) & 0xffu : 0u;
        // centre byte: in the word of x-1 or of x+1
        u32 cw = ((x & ~3) == (xl & ~3) && xl >= 0) ? lo[k] : hi[k];
        u32 c = (cw >> (8 * (x & 3))) & 0xffu;
        out[k] = l | (c << 8) | (r << 16);
    }
}

// Promote (x, y) if it is WEAK and walk the chain of WEAK pixels reachable from it.
// Returns true if some promoted pixel could not be pushed (its neighbours are then
// found by the next scan).
__device__ __forceinline__ bool dfs_from(const Plane& dst, int b, int x, int y, int w, int h, u32* stack)
{
    bool dropped = false;
    if (!promote(dst.row(b, y), x)) return false;
    int sp = 0;
    stack[sp++] = ((u32)y << 16) | (u32)x;
    while (sp) {
        const u32 p = stack[--sp];
        const int py = (int)(p >> 16), px = (int)(p & 0xffffu);
        u32 nb[3];
        load_3x3(dst, b, px, py, w, h, nb);
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (k == 1 && j == 1) continue;
                if (((nb[k] >> (8 * j)) & 0xffu) != WEAK) continue;
                const int ny = py + k - 1, nx = px + j - 1;
                if (promote(dst.row(b, ny), nx)) {
                    if (sp < DFS_DEPTH) stack[sp++] = ((u32)ny << 16) | (u32)nx;
                    else dropped = true;
                }
            }
        }
    }
    return dropped;
}

// One workgroup = one image (or one region of a huge image).  Rounds of: scan a batch
// of flagged tiles for WEAK pixels (queued in LDS), then every lane takes queued
// pixels, tests whether an edge pixel touches them and if so walks the chain.
__global__ __launch_bounds__(DFS_THREADS) void k_hyst_dfs(Plane dst, int w, int h, int tiles_x, int tiles_y, int rt_x,
                                                          int rt_y, int regions_x, int regions_y, CannyWs W)
{
    __shared__ u32 s_stack[DFS_THREADS * DFS_DEPTH];
    __shared__ u32 s_tiles[DFS_MAX_TILES];
    __shared__ u32 s_q[DFS_QCAP];
    __shared__ u32 s_ntiles, s_qn;
    __shared__ int s_overflow;
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    const int rx = blk % regions_x;
    blk /= regions_x;
    const int ry = blk % regions_y, b = blk / regions_y;
    if (tid == 0) {
        s_ntiles = 0;
        s_qn = 0;
        s_overflow = 0;
    }
    __syncthreads();
    for (int t = tid; t < rt_x * rt_y; t += DFS_THREADS) {
        int tx = rx * rt_x + t % rt_x, ty = ry * rt_y + t / rt_x;
        if (tx < tiles_x && ty < tiles_y && W.has_weak[((size_t)b * tiles_y + ty) * tiles_x + tx]) {
            u32 pos = atomicAdd(&s_ntiles, 1u);
            s_tiles[pos] = ((u32)ty << 16) | (u32)tx;
        }
    }
    __syncthreads();
    const u32 ntiles = s_ntiles;
    if (ntiles == 0) return;
    u32* stack = s_stack + tid * DFS_DEPTH;
    // scan geometry: a tile is 256 chunks of 16 pixels; 512 lanes cover 2 tiles per step,
    // DFS_SCAN_STEPS steps (independent loads) per batch
    const int half = tid >> 8, l = tid & 255;
    const int lr = l >> 3, lc0 = (l & 7) * 16;
    constexpr int TILES_PER_BATCH = 2 * DFS_SCAN_STEPS;
    for (int round = 0; round < (1 << 20); round++) {
        bool dropped = false;
        for (u32 t0 = 0; t0 < ntiles; t0 += TILES_PER_BATCH) {
            // ---- scan
            u32 d[DFS_SCAN_STEPS][4];
            int ys[DFS_SCAN_STEPS], xs[DFS_SCAN_STEPS];
#pragma unroll
            for (int s = 0; s < DFS_SCAN_STEPS; s++) {
                const u32 ti = t0 + 2 * s + half;
                ys[s] = -1;
                xs[s] = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) d[s][k] = 0;
                if (ti < ntiles) {
                    const u32 tile = s_tiles[ti];
                    const int iy = (int)(tile >> 16) * TH + lr, ix0 = (int)(tile & 0xffffu) * TW + lc0;
                    if (iy < h && ix0 < w) {
                        const u8* row = dst.row(b, iy);
                        ys[s] = iy;
                        xs[s] = ix0;
#pragma unroll
                        for (int k = 0; k < 4; k++) d[s][k] = load_word_l2(row, ix0 + 4 * k);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < DFS_SCAN_STEPS; s++) {
                if (ys[s] < 0) continue;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    u32 v = d[s][k];
                    if (v == 0u || v == 0xffffffffu) continue;
                    for (int j = 0; j < 4; j++) {
                        if (((v >> (8 * j)) & 0xffu) != WEAK) continue;
                        const int x = xs[s] + 4 * k + j;
                        if (x >= w) continue;
                        u32 pos = atomicAdd(&s_qn, 1u);
                        if (pos < DFS_QCAP) s_q[pos] = ((u32)ys[s] << 16) | (u32)x;
                        else {
                            // queue full (a noise field): handle it here
                            u32 nb[3];
                            load_3x3(dst, b, x, ys[s], w, h, nb);
                            if ((((nb[0] | nb[2]) & 0x808080u) | (nb[1] & 0x800080u)) != 0u)
                                dropped |= dfs_from(dst, b, x, ys[s], w, h, stack);
                        }
                    }
                }
            }
            __syncthreads();
            // ---- resolve the queued pixels
            const u32 qn = min(s_qn, (u32)DFS_QCAP);
            for (u32 i = tid; i < qn; i += DFS_THREADS) {
                const u32 p = s_q[i];
                const int y = (int)(p >> 16), x = (int)(p & 0xffffu);
                u32 nb[3];
                load_3x3(dst, b, x, y, w, h, nb);
                // bytes are 0, WEAK (1) or 255: an edge pixel touches iff some neighbour has its top bit set
                if ((((nb[0] | nb[2]) & 0x808080u) | (nb[1] & 0x800080u)) != 0u)
                    dropped |= dfs_from(dst, b, x, y, w, h, stack);
            }
            __syncthreads();
            if (tid == 0) s_qn = 0;
            __syncthreads();
        }
        if (dropped) s_overflow = 1;
        __syncthreads();
        const int again = s_overflow;
        __syncthreads();
        if (!again) break;
        if (tid == 0) {
            s_overflow = 0;
            // a dropped pixel may border another region: ask for one more launch
            if (regions_x * regions_y > 1) atomicExch(W.again, 1u);
        }
        __syncthreads();
    }
}

// Flagged tiles: bytes that are still WEAK become 0 (bytes are 0, WEAK or 255).
__global__ __launch_bounds__(256) void k_hyst_final(Plane dst, int w, int h, int tiles_x, int tiles_y, u32 ntiles,
                                                    const u8* __restrict__ has_weak)
{
    const int lr = threadIdx.x >> 3, lc0 = (threadIdx.x & 7) * 16;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (!has_weak[tile]) continue;
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
            // keep a byte only if its top bit is set
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
    return align_up(ntiles, 256) + 512;
}

// Enqueues the whole detector.  Images of up to DFS_MAX_TILES tiles (e.g. 4K) are
// resolved by one workgroup each with no host interaction; larger images are cut
// into regions and the stream is synchronised to learn whether a further
// hysteresis launch is needed.
int run_canny(Plane src, Plane dst, int w, int h, int batch, int low, int high, void* ws, size_t ws_bytes,
              hipStream_t st)
{
    const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
    const long long ntiles_ll = (long long)tiles_x * tiles_y * batch;
    IPA_REQUIRE(ntiles_ll < 0x7fffffffLL, IPA_ERR_ARG, "Canny: too many tiles");
    IPA_REQUIRE(w <= 65535 && h <= 65535, IPA_ERR_ARG, "Canny: image larger than 65535 pixels on a side");
    IPA_REQUIRE(ws && ws_bytes >= canny_ws_bytes(w, h, batch), IPA_ERR_WORKSPACE, "Canny: workspace too small");
    const size_t ntiles = (size_t)ntiles_ll;
    CannyWs W;
    W.has_weak = (u8*)ws;
    W.again = (u32*)((u8*)ws + align_up(ntiles, 256));
    {
        ProfScope ps(PROF_CANNY_NMS, st);
        hipLaunchKernelGGL(k_canny_nms, dim3((unsigned)ntiles), dim3(256), 0, st, src, dst, w, h, tiles_x, tiles_y, low,
                           high, W);
        IPA_LAUNCH_CHECK();
    }
    ProfScope ps(PROF_HYSTERESIS, st);
    // regions of at most DFS_MAX_TILES tiles (64 x 64 tiles = 8192 x 2048 pixels)
    int rt_x = tiles_x, rt_y = tiles_y;
    if ((long long)rt_x * rt_y > DFS_MAX_TILES) {
        rt_x = tiles_x < 64 ? tiles_x : 64;
        rt_y = DFS_MAX_TILES / rt_x;
        if (rt_y > tiles_y) rt_y = tiles_y;
    }
    const int regions_x = cdiv(tiles_x, rt_x), regions_y = cdiv(tiles_y, rt_y);
    const long long blocks = (long long)regions_x * regions_y * batch;
    IPA_REQUIRE(blocks < 0x7fffffffLL, IPA_ERR_ARG, "Canny: too many regions");
    const bool multi = regions_x * regions_y > 1;
    for (int guard = 0; guard < 100000; guard++) {
        if (multi) IPA_HIP(hipMemsetAsync(W.again, 0, 4, st));
        hipLaunchKernelGGL(k_hyst_dfs, dim3((unsigned)blocks), dim3(DFS_THREADS), 0, st, dst, w, h, tiles_x, tiles_y, rt_x,
                           rt_y, regions_x, regions_y, W);
        IPA_LAUNCH_CHECK();
        if (!multi) break;
        u32 again;
        {
            std::lock_guard<std::mutex> lk(g_pin_mu);
            if (!g_pinned) IPA_HIP(hipHostMalloc((void**)&g_pinned, 256, hipHostMallocDefault));
            IPA_HIP(hipMemcpyAsync(g_pinned, W.again, 4, hipMemcpyDeviceToHost, st));
            IPA_HIP(hipStreamSynchronize(st));
            again = g_pinned[0];
        }
        if (!again) break;
    }
    const unsigned grid = (unsigned)(ntiles < 4096 ? ntiles : 4096);
    hipLaunchKernelGGL(k_hyst_final, dim3(grid), dim3(256), 0, st, dst, w, h, tiles_x, tiles_y, (u32)ntiles,
                       (const u8*)W.has_weak);
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
