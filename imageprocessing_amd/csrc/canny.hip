// Canny edge detector (aperture 3, L1 gradient magnitude), OpenCV semantics.
//
//  k_canny_nms : one workgroup = one 128x32 tile.  Gray tile + 2-pixel halo in
//                LDS -> Sobel magnitude for tile + 1 ring -> non-maximum
//                suppression -> edge bytes: 255 above the high threshold, WEAK
//                for candidates between the thresholds, 0 otherwise.  Tiles that
//                hold WEAK pixels are flagged.
//  k_hyst_tiles: hysteresis.  One workgroup owns one image (or one region of a
//                huge image) and works through rounds of a tile worklist; a wave
//                resolves one tile in registers (bit masks + lane shuffles) and
//                queues the neighbours when its rim changed.
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

constexpr int HT_THREADS = 512;      // 8 waves per workgroup
constexpr int HT_MAX_TILES = 4096;   // tiles per region (one workgroup)

// bit 0 of each of the 4 bytes of x -> 4 adjacent bits
__device__ __forceinline__ u32 gather4(u32 x) { return ((x & 0x01010101u) * 0x10204080u) >> 28; }

struct TileBits {
    u64 S, W;  // edge / WEAK masks of this lane's 64 pixels (bit i = pixel i)
};

// edge bytes of 64 consecutive pixels -> masks.  Pixels at or beyond `nvalid` are empty.
__device__ __forceinline__ TileBits load_bits64(const u8* p, int nvalid)
{
    TileBits t;
    t.S = 0;
    t.W = 0;
    if (nvalid <= 0) return t;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (q * 16 >= nvalid) break;
        const uint4 v = *(const uint4*)(p + 16 * q);
        const u32 d[4] = {v.x, v.y, v.z, v.w};
        u32 s16 = 0, w16 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 top = d[k] >> 7;
            s16 |= gather4(top) << (4 * k);
            w16 |= gather4(d[k] & ~top) << (4 * k);
        }
        t.S |= (u64)s16 << (16 * q);
        t.W |= (u64)w16 << (16 * q);
    }
    if (nvalid < 64) {
        const u64 m = (1ull << nvalid) - 1ull;
        t.S &= m;
        t.W &= m;
    }
    return t;
}

// Hysteresis on tiles.  One workgroup owns one image (or one region of a huge image) and
// works through rounds of a tile worklist held in LDS; the first round holds every tile
// that has WEAK pixels.  One wave resolves one 128x32 tile at a time entirely in
// registers: lane = (row, 64-pixel half), the edge / WEAK state as two 64-bit masks, a
// 1-pixel ring of the neighbours' edge bits around it, and a fixed point of
//     S |= W & dilate3x3(S)
// with row neighbours exchanged by lane shuffles.  Promoted pixels are written back with
// atomic ORs; if the rim of the tile changed, the neighbouring tiles with WEAK pixels go on
// the next round's list.  Rounds are separated by a workgroup barrier (which also drains
// the stores) and an L1 invalidate, so a round sees everything the previous one wrote.
__global__ __launch_bounds__(HT_THREADS) void k_hyst_tiles(Plane dst, int w, int h, int tiles_x, int tiles_y,
                                                           int rt_x, int rt_y, int regions_x, int regions_y,
                                                           CannyWs W)
{
    __shared__ u16 s_list[2][HT_MAX_TILES];
    __shared__ u32 s_inq[HT_MAX_TILES / 32];
    __shared__ u32 s_flag[HT_MAX_TILES / 32];
    __shared__ u32 s_cnt[2];
    __shared__ u32 s_cursor;
    const int tid = threadIdx.x, lane = tid & 63;
    int blk = blockIdx.x;
    const int rx = blk % regions_x;
    blk /= regions_x;
    const int ry = blk % regions_y, b = blk / regions_y;
    const int tx_base = rx * rt_x, ty_base = ry * rt_y;
    const int nx_t = min(rt_x, tiles_x - tx_base), ny_t = min(rt_y, tiles_y - ty_base);

    for (int i = tid; i < HT_MAX_TILES / 32; i += HT_THREADS) {
        s_inq[i] = 0;
        s_flag[i] = 0;
    }
    if (tid == 0) {
        s_cnt[0] = 0;
        s_cnt[1] = 0;
        s_cursor = 0;
    }
    __syncthreads();
    for (int t = tid; t < nx_t * ny_t; t += HT_THREADS) {
        const int lx = t % nx_t, ly = t / nx_t;
        if (W.has_weak[((size_t)b * tiles_y + ty_base + ly) * tiles_x + tx_base + lx]) {
            atomicOr(&s_flag[t >> 5], 1u << (t & 31));
            const u32 pos = atomicAdd(&s_cnt[0], 1u);
            s_list[0][pos] = (u16)t;
        }
    }
    __syncthreads();

    const int r = lane >> 1, hf = lane & 1;
    int cur = 0;
    for (int round = 0; round < (1 << 24); round++) {
        const u32 n = s_cnt[cur];
        if (n == 0) break;
        const int nxt = cur ^ 1;
        for (;;) {
            u32 i = 0;
            if (lane == 0) i = atomicAdd(&s_cursor, 1u);
            i = __shfl(i, 0);
            if (i >= n) break;
            const int t = s_list[cur][i];
            const int lx = t % nx_t, ly = t / nx_t;
            const int x0 = (tx_base + lx) * TW, y0 = (ty_base + ly) * TH;
            const int iy = y0 + r, xh = x0 + 64 * hf;
            // own 64 pixels
            TileBits me;
            me.S = me.W = 0;
            if (iy < h) me = load_bits64(dst.row(b, iy) + xh, w - xh);
            // the edge bit just outside this lane's end of the row (ring column)
            const int xs = hf ? x0 + TW : x0 - 1;
            u32 side = 0;
            if (iy < h && xs >= 0 && xs < w) side = dst.row(b, iy)[xs] >> 7;
            // ring rows: lanes of row 0 hold the row above the tile, lanes of row 31 the row below
            u64 ringS = 0;
            u32 ring_side = 0;
            const int ry_ = r == 0 ? y0 - 1 : y0 + TH;
            if ((r == 0 || r == TH - 1) && ry_ >= 0 && ry_ < h) {
                ringS = load_bits64(dst.row(b, ry_) + xh, w - xh).S;
                if (xs >= 0 && xs < w) ring_side = dst.row(b, ry_)[xs] >> 7;
            }
            // horizontally dilated ring row (constant during the iteration)
            u64 ringH;
            {
                const u64 pr = __shfl_xor(ringS, 1);
                const u64 lin = hf ? (pr >> 63) : (u64)ring_side;
                const u64 rin = hf ? (u64)ring_side : (pr & 1ull);
                ringH = ringS | (ringS << 1) | (ringS >> 1) | lin | (rin << 63);
            }
            const u64 S0 = me.S;
            u64 S = me.S;
            if (__any(me.W != 0)) {
                for (;;) {
                    const u64 pS = __shfl_xor(S, 1);
                    const u64 lin = hf ? (pS >> 63) : (u64)side;
                    const u64 rin = hf ? (u64)side : (pS & 1ull);
                    const u64 H = S | (S << 1) | (S >> 1) | lin | (rin << 63);
                    u64 up = __shfl_up(H, 2), dn = __shfl_down(H, 2);
                    if (r == 0) up = ringH;
                    if (r == TH - 1) dn = ringH;
                    const u64 Sn = S | (me.W & (H | up | dn));
                    const bool ch = Sn != S;
                    S = Sn;
                    if (!__any(ch)) break;
                }
            }
            const u64 prom = S & ~S0;
            if (prom) {
                u8* row = dst.row(b, iy) + xh;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const u32 nib = (u32)(prom >> (4 * q)) & 0xfu;
                    if (nib) atomicOr((u32*)(row + 4 * q), ((nib * 0x00204081u) & 0x01010101u) * 0xffu);
                }
            }
            // rim changed -> neighbours with WEAK pixels are due in the next round
            const bool rim = prom && (r == 0 || r == TH - 1 || (hf == 0 && (prom & 1ull)) || (hf == 1 && (prom >> 63)));
            if (__any(rim) && lane < 9 && lane != 4) {
                const int nlx = lx + (lane % 3) - 1, nly = ly + (lane / 3) - 1;
                const int gtx = tx_base + nlx, gty = ty_base + nly;
                if (gtx >= 0 && gtx < tiles_x && gty >= 0 && gty < tiles_y) {
                    if (nlx >= 0 && nlx < nx_t && nly >= 0 && nly < ny_t) {
                        const int nt = nly * nx_t + nlx;
                        if ((s_flag[nt >> 5] >> (nt & 31)) & 1u) {
                            const u32 old = atomicOr(&s_inq[nt >> 5], 1u << (nt & 31));
                            if (!((old >> (nt & 31)) & 1u)) {
                                const u32 pos = atomicAdd(&s_cnt[nxt], 1u);
                                s_list[nxt][pos] = (u16)nt;
                            }
                        }
                    } else if (W.has_weak[((size_t)b * tiles_y + gty) * tiles_x + gtx]) {
                        atomicExch(W.again, 1u);  // belongs to another workgroup's region: one more launch
                    }
                }
            }
        }
        __syncthreads();  // also waits for this round's stores and atomics
        if (tid == 0) {
            s_cnt[cur] = 0;
            s_cursor = 0;
        }
        for (int i = tid; i < HT_MAX_TILES / 32; i += HT_THREADS) s_inq[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale L1 lines before re-reading tiles
        __syncthreads();
        cur = nxt;
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

// Enqueues the whole detector.  Images of up to HT_MAX_TILES tiles (e.g. 4K) are
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
    // regions of at most HT_MAX_TILES tiles (64 x 64 tiles = 8192 x 2048 pixels)
    int rt_x = tiles_x, rt_y = tiles_y;
    if ((long long)rt_x * rt_y > HT_MAX_TILES) {
        rt_x = tiles_x < 64 ? tiles_x : 64;
        rt_y = HT_MAX_TILES / rt_x;
        if (rt_y > tiles_y) rt_y = tiles_y;
    }
    const int regions_x = cdiv(tiles_x, rt_x), regions_y = cdiv(tiles_y, rt_y);
    const long long blocks = (long long)regions_x * regions_y * batch;
    IPA_REQUIRE(blocks < 0x7fffffffLL, IPA_ERR_ARG, "Canny: too many regions");
    const bool multi = regions_x * regions_y > 1;
    for (int guard = 0; guard < 100000; guard++) {
        if (multi) IPA_HIP(hipMemsetAsync(W.again, 0, 4, st));
        hipLaunchKernelGGL(k_hyst_tiles, dim3((unsigned)blocks), dim3(HT_THREADS), 0, st, dst, w, h, tiles_x, tiles_y, rt_x,
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
