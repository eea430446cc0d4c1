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
#include <stdlib.h>
#include <math.h>

namespace {

constexpr int TW = 128, TH = 32;
constexpr int GH = TH + 4;     // gray rows in LDS: tile + 2 above / below
constexpr int GP = 160;        // gray row pitch in bytes: LDS column = x - x0 + 16 (tile body 16-byte aligned)
constexpr int MH = TH + 2;     // magnitude rows: tile + 1 ring
constexpr int MP = 136;        // magnitude row pitch in u16: column m <-> x = x0 - 4 + m
constexpr int MGROUPS = 34;    // groups of 4 magnitudes per row
constexpr u8 WEAK = 1;

struct CannyWs {
    u8* has_weak;   // per tile: the tile holds WEAK pixels
    u32* again;     // set when a tile of another workgroup's region has to be revisited
    u8* pending;    // per tile, or nullptr: receives the tiles of other regions that have to be revisited
    const u8* pending_in;  // per tile: the marks a seed_pending launch starts from (may be the same array as `pending`)
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 as_s2(u32 v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ u32 as_u32(s2 v) { return __builtin_bit_cast(u32, v); }
// two bytes of the 8-byte value {hi, lo} as two zero-extended 16-bit lanes (a -> low lane)
#define PAIR_SEL(a, b) ((0x0cu << 24) | ((u32)(b) << 16) | (0x0cu << 8) | (u32)(a))

// LDS layout (one 128x32 tile):
//   g[36][160]   gray bytes, image (y0-2+r, x0-16+c), borders replicated
//   mag[34][136] |dx|+|dy| as u16, image (y0-1+r, x0-4+m), zero outside the image
//   M[32][128]   output codes 0 / WEAK / 255
//   Q[4096]      tile-local coordinates of pixels whose magnitude exceeds the low threshold
// Phases: load gray (16-byte loads) -> Sobel with packed 16-bit arithmetic, two pixels
// per instruction, four per lane, magnitudes to LDS and candidates to the queue ->
// direction test + neighbour compare for queued pixels only -> store codes.  Only ~3 %
// of the pixels of a diagram are candidates, so the branchy non-maximum test runs on a
// compacted list with all lanes busy.
__global__ __launch_bounds__(256) void k_canny_nms(Plane src, Plane dst, int w, int h, int tiles_x, int tiles_y,
                                                   int low, int high, CannyWs W)
{
    __shared__ __align__(16) u8 g[GH][GP];
    __shared__ __align__(16) u16 mag[MH][MP];
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

    // ---- gray tile: per row 8 x 16 bytes of body + one dword of halo on each side
    for (int e = tid; e < GH * 10; e += 256) {
        const int r = e / 10, k = e - r * 10;
        const int gy = clampi(y0 - 2 + r, 0, h - 1);
        const u8* row = src.row(b, gy);
        if (k < 8) {
            const int gx = x0 + 16 * k;
            uint4 v;
            if (gx + 16 <= w) {
                v = *(const uint4*)(row + gx);
            } else {
                u32 d[4] = {0, 0, 0, 0};
                for (int j = 0; j < 16; j++) d[j >> 2] |= (u32)row[clampi(gx + j, 0, w - 1)] << (8 * (j & 3));
                v = make_uint4(d[0], d[1], d[2], d[3]);
            }
            *(uint4*)&g[r][16 + 16 * k] = v;
        } else {
            const int gx = k == 8 ? x0 - 4 : x0 + TW;
            u32 v;
            if (gx >= 0 && gx + 4 <= w) {
                v = *(const u32*)(row + gx);
            } else {
                v = 0;
                for (int j = 0; j < 4; j++) v |= (u32)row[clampi(gx + j, 0, w - 1)] << (8 * j);
            }
            *(u32*)&g[r][k == 8 ? 12 : 16 + TW] = v;
        }
    }
    ((uint4*)&M[0][0])[tid] = make_uint4(0, 0, 0, 0);
    if (tid == 0) {
        s_qn = 0;
        s_weak = 0;
    }
    __syncthreads();

    // ---- Sobel + magnitude, 4 pixels (two 16-bit pairs) per lane and step.
    // Group q of magnitude row r covers m = 4q .. 4q+3, i.e. x = x0-4+4q ..; it needs the
    // gray columns x-1 .. x+4 = LDS bytes 4q+11 .. 4q+16 = dwords q+2 (byte 3), q+3, q+4 (byte 0).
    // vs = top + 2 mid + bot and vd = bot - top are formed for the column pairs
    // A = (x-1, x), B = (x+1, x+2), C = (x+3, x+4); then
    //   dx(0,1) = vs_B - vs_A            dx(2,3) = vs_C - vs_B
    //   dy(0,1) = vd_A + 2 vd(x, x+1) + vd_B   with (x, x+1) = the lanes between A and B, etc.
    const bool border_tile = x0 == 0 || y0 == 0 || x0 + TW + 1 > w || y0 + TH + 1 > h;
    for (int e = tid; e < MH * MGROUPS; e += 256) {
        const int r = e / MGROUPS, q = e - r * MGROUPS;
        s2 vs[3], vd[3];
        {
            const u32* gt = (const u32*)&g[r][0] + q + 2;
            const u32* gm = (const u32*)&g[r + 1][0] + q + 2;
            const u32* gb = (const u32*)&g[r + 2][0] + q + 2;
            const u32 t0 = gt[0], t1 = gt[1], t2 = gt[2];
            const u32 m0 = gm[0], m1 = gm[1], m2 = gm[2];
            const u32 b0 = gb[0], b1 = gb[1], b2 = gb[2];
            // pair A: byte 3 of dword 0, byte 0 of dword 1; B: bytes 1, 2 of dword 1; C: byte 3 of dword 1, byte 0 of dword 2
            const s2 tA = as_s2(__builtin_amdgcn_perm(t1, t0, PAIR_SEL(3, 4))), tB = as_s2(__builtin_amdgcn_perm(t1, t1, PAIR_SEL(1, 2))),
                     tC = as_s2(__builtin_amdgcn_perm(t2, t1, PAIR_SEL(3, 4)));
            const s2 mA = as_s2(__builtin_amdgcn_perm(m1, m0, PAIR_SEL(3, 4))), mB = as_s2(__builtin_amdgcn_perm(m1, m1, PAIR_SEL(1, 2))),
                     mC = as_s2(__builtin_amdgcn_perm(m2, m1, PAIR_SEL(3, 4)));
            const s2 bA = as_s2(__builtin_amdgcn_perm(b1, b0, PAIR_SEL(3, 4))), bB = as_s2(__builtin_amdgcn_perm(b1, b1, PAIR_SEL(1, 2))),
                     bC = as_s2(__builtin_amdgcn_perm(b2, b1, PAIR_SEL(3, 4)));
            vs[0] = tA + mA + mA + bA; vs[1] = tB + mB + mB + bB; vs[2] = tC + mC + mC + bC;
            vd[0] = bA - tA; vd[1] = bB - tB; vd[2] = bC - tC;
        }
        const s2 dx01 = vs[1] - vs[0], dx23 = vs[2] - vs[1];
        // the pairs that straddle A|B and B|C
        const s2 vdAB = as_s2(__builtin_amdgcn_alignbit(as_u32(vd[1]), as_u32(vd[0]), 16));
        const s2 vdBC = as_s2(__builtin_amdgcn_alignbit(as_u32(vd[2]), as_u32(vd[1]), 16));
        const s2 dy01 = vd[0] + vdAB + vdAB + vd[1], dy23 = vd[1] + vdBC + vdBC + vd[2];
        s2 m01 = __builtin_elementwise_abs(dx01) + __builtin_elementwise_abs(dy01);
        s2 m23 = __builtin_elementwise_abs(dx23) + __builtin_elementwise_abs(dy23);
        u32 mm[2] = {as_u32(m01), as_u32(m23)};
        const int iy = y0 - 1 + r, ix = x0 - 4 + 4 * q;
        if (border_tile) {
            // magnitude outside the image is 0
            const bool row_in = iy >= 0 && iy < h;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool in = row_in && ix + j >= 0 && ix + j < w;
                if (!in) mm[j >> 1] &= (j & 1) ? 0x0000ffffu : 0xffff0000u;
            }
        }
        *(uint2*)&mag[r][4 * q] = make_uint2(mm[0], mm[1]);
        // candidates: tile pixels (rows 1..32, groups 1..32) above the low threshold
        if (r >= 1 && r <= TH && q >= 1 && q <= 32) {
            const int lr = r - 1, lc = 4 * q - 4;
            u32 cand = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int mv = (int)((mm[j >> 1] >> (16 * (j & 1))) & 0xffffu);
                if (mv > low && iy < h && ix + j < w) cand |= 1u << j;
            }
            if (cand) {
                u32 pos = atomicAdd(&s_qn, (u32)__popc(cand));
                while (cand) {
                    const int j = __ffs(cand) - 1;
                    cand &= cand - 1;
                    Q[pos++] = (u16)((lr << 7) | (lc + j));
                }
            }
        }
    }
    __syncthreads();

    // ---- non-maximum suppression on the queued pixels
    const u32 qn = s_qn;
    bool weak = false;
    for (u32 i = tid; i < qn; i += 256) {
        const int e = Q[i];
        const int r = e >> 7, c = e & 127;
        const u8* p0 = &g[r + 1][c + 15];  // row above, column x-1
        const u8* p1 = p0 + GP;
        const u8* p2 = p1 + GP;
        const int dx = (p0[2] - p0[0]) + 2 * (p1[2] - p1[0]) + (p2[2] - p2[0]);
        const int dy = (p2[0] - p0[0]) + 2 * (p2[1] - p0[1]) + (p2[2] - p0[2]);
        const u16* ctr = &mag[r + 1][c + 4];
        const int m = ctr[0];
        const int ax = abs(dx), ay = abs(dy) << 15;
        const int tg22x = ax * 13573;
        bool is_max;
        if (ay < tg22x) {
            is_max = m > ctr[-1] && m >= ctr[1];
        } else {
            const int tg67x = tg22x + (ax << 16);
            if (ay > tg67x) {
                is_max = m > ctr[-MP] && m >= ctr[MP];
            } else {
                const int s = ((dx ^ dy) < 0) ? -1 : 1;
                is_max = m > ctr[-MP - s] && m > ctr[MP + s];
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
        const int lr = tid >> 3, lc0 = (tid & 7) * 16;
        const int iy = y0 + lr, ix0 = x0 + lc0;
        if (iy < h && ix0 < w) *(uint4*)(dst.row(b, iy) + ix0) = ((const uint4*)&M[0][0])[tid];
    }
    if (tid == 0) W.has_weak[blockIdx.x] = s_weak ? 1 : 0;
}

// ---- streaming variant: no LDS, no barriers ---------------------------------------
// One wave walks a 256-pixel-wide strip of the image downwards, one row per step, lane =
// 4 adjacent pixels.  Everything lives in registers: the last three gray rows (as 16-bit
// pairs), the last three magnitude rows and the gradients of the middle one.  Column
// neighbours come from the adjacent lanes by shuffles; the outermost lanes only feed their
// neighbours (strips overlap by 8 pixels), so a strip yields 248 columns.  Rows are read
// and written as contiguous 256-byte segments, and neighbouring strips advance together,
// which keeps the HBM access pattern row-sequential.
constexpr int ST_ROWS = 64, ST_ROWS_SHORT = 16;
constexpr int ST_COLS = 248;

struct MagRow {
    u32 m01, m23;  // magnitudes of the lane's 4 pixels (16-bit pairs)
    u32 nL, nR;    // the left lane's m23 (columns -2,-1) / the right lane's m01 (columns 4,5)
};

// Value of the same register in lane-1 / lane+1 (one DPP move each; lane 0 / lane 63 get 0).
__device__ __forceinline__ u32 from_lane_below(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ u32 from_lane_above(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xf, 0xf, true); }

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(u32 v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ u32 us2_u32(us2 v) { return __builtin_bit_cast(u32, v); }
// per 16-bit half: max(a - b, 0)
__device__ __forceinline__ us2 sub_sat(us2 a, us2 b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ u32 pk_sub_sat_u16(u32 a, u32 b) { return us2_u32(sub_sat(as_us2(a), as_us2(b))); }
// Non-maximum suppression of two adjacent pixels at once, in packed 16-bit arithmetic and
// without branches.  m = magnitudes, ax/ay = |dx|,|dy|, sgn = dx ^ dy; CL/CR = magnitudes one
// column to the left/right in the same row, PL/Pc/PR and NL/Nc/NR the three neighbours in the
// row above / below; low2/high2 = thresholds in both halves, span2 = 255 - WEAK in both halves
// (WEAK must be 1).  Returns per half 0, WEAK or 255.
//   horizontal  <=>  |dy| * 2^15 < |dx| * 13573          <=>  |dy| <  ceil(|dx| * 13573 / 2^15)
//   vertical    <=>  |dy| * 2^15 > |dx| * (13573 + 2^16)  <=>  |dy| >  floor(|dx| * 13573 / 2^15) + 2 |dx|
// (|dx| * 13573 is a multiple of 2^15 only for dx = 0), and floor(|dx| * 13573 / 2^15) is
// ((|dx| * 53) + ((|dx| * 5) >> 8)) >> 7 with every intermediate below 2^16 for |dx| <= 1020.
// Written as one block of packed instructions: the optimiser otherwise splits these idioms
// into per-half compares and selects, which doubles the instruction count.
static_assert(WEAK == 1, "nms_pair encodes WEAK as 1");
__device__ __forceinline__ u32 nms_pair(u32 m, u32 ax, u32 ay, u32 sgn, u32 CL, u32 CR, u32 PL, u32 Pc, u32 PR,
                                        u32 NL, u32 Nc, u32 NR, u32 low2, u32 high2, u32 span2)
{
    // ax, ay and sgn are dead early and serve as temporaries afterwards (t4, t2, t3): the block needs two
    // registers beyond its operands
    u32 t0, t1;
    asm("v_pk_mul_lo_u16 %[t0], %[ax], 5 op_sel_hi:[1,0]\n\t"
        "v_pk_lshrrev_b16 %[t0], 8, %[t0] op_sel_hi:[0,1]\n\t"
        "v_pk_mad_u16 %[t0], %[ax], 53, %[t0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_lshrrev_b16 %[t0], 7, %[t0] op_sel_hi:[0,1]\n\t"       // f = floor(|dx| * 13573 / 2^15)
        "v_pk_min_u16 %[t1], %[ax], 1 op_sel_hi:[1,0]\n\t"
        "v_pk_add_u16 %[t1], %[t0], %[t1]\n\t"                       // ceil(...)
        "v_pk_sub_u16 %[t1], %[t1], %[ay] clamp\n\t"
        "v_pk_min_u16 %[t1], %[t1], 1 op_sel_hi:[1,0]\n\t"           // hb = 1: horizontal gradient
        "v_pk_mad_u16 %[t0], %[ax], 2, %[t0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_sub_u16 %[t0], %[ay], %[t0] clamp\n\t"
        "v_pk_min_u16 %[t0], %[t0], 1 op_sel_hi:[1,0]\n\t"           // vb = 1: vertical gradient
        "v_pk_sub_u16 %[ay], 1, %[t1] op_sel_hi:[0,1]\n\t"
        "v_pk_sub_u16 %[ay], %[ay], %[t0]\n\t"                       // 1 on the diagonals
        "v_pk_sub_u16 %[t1], 0, %[t1] op_sel_hi:[0,1]\n\t"           // hb, vb -> 0xffff masks
        "v_pk_sub_u16 %[t0], 0, %[t0] op_sel_hi:[0,1]\n\t"
        "v_pk_ashrrev_i16 %[sg], 15, %[sg] op_sel_hi:[0,1]\n\t"      // 0xffff: dx, dy of opposite sign
        "v_bfi_b32 %[ax], %[sg], %[PR], %[PL]\n\t"
        "v_bfi_b32 %[ax], %[t0], %[Pc], %[ax]\n\t"
        "v_bfi_b32 %[ax], %[t1], %[CL], %[ax]\n\t"                   // neighbour that must be below m
        "v_bfi_b32 %[sg], %[sg], %[NL], %[NR]\n\t"
        "v_bfi_b32 %[sg], %[t0], %[Nc], %[sg]\n\t"
        "v_bfi_b32 %[sg], %[t1], %[CR], %[sg]\n\t"                   // neighbour that must not exceed m
        "v_pk_add_u16 %[sg], %[sg], %[ay]\n\t"                       //   (must be below m on the diagonals)
        "v_pk_sub_u16 %[sg], %[sg], %[m] clamp\n\t"                  // != 0: fails
        "v_pk_max_u16 %[ax], %[ax], %[low]\n\t"
        "v_pk_sub_u16 %[ax], %[m], %[ax] clamp\n\t"
        "v_pk_min_u16 %[ax], %[ax], 1 op_sel_hi:[1,0]\n\t"           // 1: m > neighbour and m > low
        "v_pk_sub_u16 %[ax], %[ax], %[sg] clamp\n\t"                 // 1: local maximum
        "v_pk_sub_u16 %[t0], %[m], %[high] clamp\n\t"
        "v_pk_min_u16 %[t0], %[t0], 1 op_sel_hi:[1,0]\n\t"           // 1: above the high threshold
        "v_pk_mad_u16 %[t0], %[t0], %[span], 1 op_sel_hi:[1,1,0]\n\t"
        "v_pk_mul_lo_u16 %[t0], %[ax], %[t0]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [ax] "+v"(ax), [ay] "+v"(ay), [sg] "+v"(sgn)
        : [m] "v"(m), [CL] "v"(CL), [CR] "v"(CR), [PL] "v"(PL), [Pc] "v"(Pc),
          [PR] "v"(PR), [NL] "v"(NL), [Nc] "v"(Nc), [NR] "v"(NR), [low] "s"(low2), [high] "s"(high2), [span] "s"(span2));
    return t0;
}

// Deferred non-maximum suppression (QUEUE).  On a diagram about 55 % of a wave's rows hold a magnitude above
// the low threshold, but in such a row only about 11 of the 64 lanes do, and the suppression is half of
// the vector instructions of a row.  Lanes with a candidate therefore put what the suppression needs
// (the three magnitude rows of their columns -1 .. 4, |dx| of their four pixels, the signs of dx ^ dy,
// their row and lane) into a per-wave LDS stack of 48-byte entries; when 64 entries wait, every lane
// takes one and the suppression runs once with all lanes busy.  A lane writes zeros for pixels it has
// queued; the lane that takes the entry overwrites them (after the row's own stores, see ZFILL).
// Rows in which NQ_DENSE or more lanes have a candidate are suppressed in place.
constexpr int NQ_CAP = 104;    // at most 63 waiting + NQ_DENSE - 1 new; 4 waves x 104 x 48 B = 19968 B: 8 workgroups per CU
constexpr int NQ_DENSE = 40;
static_assert(63 + NQ_DENSE - 1 <= NQ_CAP, "queue capacity");

// ROWS = rows per band (ROWS + 2 a multiple of 3): 64 for batches, 16 when a launch would otherwise have
// too few waves to fill the chip (a wave walks its band row by row, so short bands also shorten the
// longest dependent chain of the launch)
template <bool EMIT, int ROWS, bool QUEUE>
__global__ __launch_bounds__(256) void k_canny_stream(Plane src, Plane dst, int w, int h, int strips_x, int bands,
                                                      long long nwaves, int low, int high, int tiles_x, int tiles_y,
                                                      CannyWs W, PointSink sink)
{
    static_assert(!(EMIT && QUEUE), "the edge-list variant suppresses in place");
    static_assert(!QUEUE || ROWS <= 128, "a queue entry holds the row within the band in 7 bits");
    __shared__ u32 s_pts[EMIT ? 4 : 1][EMIT ? 256 : 1];  // per wave: edge pixels waiting to be appended
    __shared__ uint4 s_q[QUEUE ? 4 : 1][3][QUEUE ? NQ_CAP : 1];  // per wave: lanes waiting for their suppression
    // Lanes that queue their pixels still write the row's zeros (whole 248-byte row pieces: masked stores
    // leave partial cache lines, 10 % more HBM write traffic measured); the lane that takes the entry writes
    // the codes later, after a wait for every store of the wave so far, so the two stores cannot swap.
    constexpr bool ZFILL = QUEUE;
    // wave-uniform values are forced into scalar registers so that the row loop, the row
    // addresses and every "whole wave" decision below run on the scalar unit
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // neighbouring strips and bands (they share cache lines and halo rows) on one XCD
    const long long wave_id = (long long)xcd_contiguous_id(blockIdx.x, gridDim.x) * 4 + wv;
    if (wave_id >= nwaves) return;
    const int lane = threadIdx.x & 63;
    u32* pbuf = s_pts[EMIT ? wv : 0];
    u32 fill = 0;  // wave-uniform
    // append the buffered points to the image's list: one atomic per flush, coalesced stores
    auto flush = [&](int bimg) {
        if (!EMIT || fill == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u32 base = 0;
        if (lane == 0) base = (u32)atomicAdd(&sink.counts[bimg], (int32_t)fill);
        base = __shfl(base, 0);
        u32* out = sink.pts + (size_t)bimg * sink.stride;
        for (u32 i = lane; i < fill; i += 64)
            if (base + i < sink.cap) out[base + i] = pbuf[i];
        __builtin_amdgcn_wave_barrier();
        fill = 0;
    };
    // (the host keeps the number of waves below 2^31: 32-bit divisions)
    const u32 wid = (u32)wave_id;
    const u32 t = wid / (u32)strips_x;
    const int sx = (int)(wid - t * (u32)strips_x);
    const int b = (int)(t / (u32)bands), band = (int)(t - (u32)b * (u32)bands);
    const int x = sx * ST_COLS - 4 + 4 * lane;
    static_assert((ROWS + 2) % 3 == 0, "the row loop is unrolled by three");
    const int y0 = band * ROWS, y1 = min(y0 + ROWS, h);
    // magnitude is 0 outside the image
    const u32 cm01 = ((x >= 0 && x < w) ? 0x0000ffffu : 0u) | ((x + 1 >= 0 && x + 1 < w) ? 0xffff0000u : 0u);
    const u32 cm23 = ((x + 2 >= 0 && x + 2 < w) ? 0x0000ffffu : 0u) | ((x + 3 >= 0 && x + 3 < w) ? 0xffff0000u : 0u);
    const bool writer = lane >= 1 && lane <= 62 && x < w;
    // Every lane reads one (possibly unaligned) 32-bit word that lies inside the row; the byte
    // selectors below pick the replicated border pixel for columns outside the image while they
    // widen the four bytes to 16-bit pairs, so there is no separate border path.
    const int xc = clampi(x, 0, w - 4);
    u32 sel01, sel23;
    {
        const u32 i0 = (u32)(clampi(x, 0, w - 1) - xc), i1 = (u32)(clampi(x + 1, 0, w - 1) - xc);
        const u32 i2 = (u32)(clampi(x + 2, 0, w - 1) - xc), i3 = (u32)(clampi(x + 3, 0, w - 1) - xc);
        sel01 = PAIR_SEL(i0, i1);
        sel23 = PAIR_SEL(i2, i3);
    }
    const u8* const simg = src.p + (size_t)b * src.stride;  // wave-uniform base, 32-bit offsets below
    u8* const dimg = dst.p + (size_t)b * dst.stride;
    const u32 spitch = (u32)src.pitch, dpitch = (u32)dst.pitch;
    auto load_row = [&](int yy) -> u32 {
        const u32 off = (u32)clampi(yy, 0, h - 1) * spitch + (u32)xc;
        return *(const u32*)(simg + (size_t)off);
    };
    u8* const weak_row0 = W.has_weak + (size_t)b * tiles_y * tiles_x + x / TW;
    u8* const weak_img = W.has_weak + (size_t)b * tiles_y * tiles_x;  // wave-uniform
    uint4(*const q)[QUEUE ? NQ_CAP : 1] = s_q[QUEUE ? wv : 0];
    u32 pending = 0;  // wave-uniform: entries on the stack
    const int x_lane0 = sx * ST_COLS - 4;

    // gray rows yy-2 (a), yy-1 (bq) as 16-bit pairs (columns 0,1 and 2,3)
    s2 a0, a1, bq0, bq1;
    {
        u32 g = load_row(y0 - 2);
        a0 = as_s2(__builtin_amdgcn_perm(0, g, sel01));
        a1 = as_s2(__builtin_amdgcn_perm(0, g, sel23));
        g = load_row(y0 - 1);
        bq0 = as_s2(__builtin_amdgcn_perm(0, g, sel01));
        bq1 = as_s2(__builtin_amdgcn_perm(0, g, sel23));
    }
    MagRow mp = {0, 0, 0, 0}, mc = {0, 0, 0, 0};  // magnitude rows yy-3, yy-2 (set as they appear)
    u32 axc01 = 0, axc23 = 0, ayc01 = 0, ayc23 = 0, sgc01 = 0, sgc23 = 0;  // |dx|, |dy|, dx ^ dy of row yy-2
    // A gray row is "flat" when every pixel the wave sees has one value; three equal flat rows
    // in a row have zero gradients, so the Sobel step is skipped (wave-uniform test).
    int flat_run = 0;      // consecutive flat rows of value flat_val ending at the newest row
    u32 flat_val = 0;
    bool mp_zero = true, mc_zero = true;   // magnitude rows yy-3 / yy-2 are known to be all zero
    const u32 low2 = (u32)__builtin_amdgcn_readfirstlane(min(max(low, 0), 0x7fff) * 0x00010001);  // magnitudes never exceed 2040
    const u32 high2 = (u32)__builtin_amdgcn_readfirstlane(min(max(high, 0), 0x7fff) * 0x00010001);
    const u32 span2 = (u32)__builtin_amdgcn_readfirstlane((int)((255u - WEAK) * 0x00010001u));
    // entries first .. first + count - 1 (count <= 64) leave the stack: lane i suppresses entry first + i
    auto pop = [&](u32 first, u32 count) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((u32)lane < count) {
            const uint4 e0 = q[0][first + lane], e1 = q[1][first + lane], e2 = q[2][first + lane];
            const u32 info = e2.w;
            const bool pz = (info & (1u << 16)) != 0, nz = (info & (1u << 17)) != 0;  // row above / below known to be zero
            const u32 c01 = e0.x, c23 = e0.y, cO = e0.z;
            const u32 p01 = pz ? 0u : e1.x, p23 = pz ? 0u : e1.y, pO = pz ? 0u : e1.z;
            const u32 n01 = nz ? 0u : e2.x, n23 = nz ? 0u : e2.y, nO = nz ? 0u : e2.z;
            // xO: column -1 in the high half, column 4 in the low half
            const u32 cA = __builtin_amdgcn_alignbit(c01, cO, 16), cB = __builtin_amdgcn_alignbit(c23, c01, 16),
                      cC = __builtin_amdgcn_alignbit(cO, c23, 16);
            const u32 pA = __builtin_amdgcn_alignbit(p01, pO, 16), pB = __builtin_amdgcn_alignbit(p23, p01, 16),
                      pC = __builtin_amdgcn_alignbit(pO, p23, 16);
            const u32 nA = __builtin_amdgcn_alignbit(n01, nO, 16), nB = __builtin_amdgcn_alignbit(n23, n01, 16),
                      nC = __builtin_amdgcn_alignbit(nO, n23, 16);
            // |dy| = magnitude - |dx| (a column outside the image has magnitude 0 and yields 0 whatever |dy| is)
            const u32 ax01 = e0.w, ax23 = e1.w;
            const u32 ay01 = as_u32(as_s2(c01) - as_s2(ax01)), ay23 = as_u32(as_s2(c23) - as_s2(ax23));
            // sign of dx ^ dy of pixel k in bit 8 k + 7 of info -> bit 15 of its 16-bit half
            const u32 sg01 = __builtin_amdgcn_perm(0, info, 0x01010000u), sg23 = __builtin_amdgcn_perm(0, info, 0x03030202u);
            const u32 k01 = nms_pair(c01, ax01, ay01, sg01, cA, cB, pA, p01, pB, nA, n01, nB, low2, high2, span2);
            const u32 k23 = nms_pair(c23, ax23, ay23, sg23, cB, cC, pB, p23, pC, nB, n23, nC, low2, high2, span2);
            const u32 codes = __builtin_amdgcn_perm(k23, k01, 0x06040200u);
            const u32 ey = (u32)y0 + (info & 0x7fu), ex = (u32)x_lane0 + 4u * ((info >> 8) & 0x3fu);
            if (ZFILL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zeros written for this row have arrived
            *(u32*)(dimg + (size_t)(ey * dpitch + ex)) = codes;
            if (codes & ~(codes >> 7) & 0x01010101u) weak_img[(ey / TH) * (u32)tiles_x + ex / TW] = 1;
        }
        __builtin_amdgcn_wave_barrier();
    };
    const u32 lane_info = (u32)lane << 8;
    // three gray rows are always in flight ahead of the row being consumed
    u32 gq[3];
#pragma unroll
    for (int k = 0; k < 3; k++) gq[k] = load_row(y0 + k);
    // fixed trip count (rows past the image are clamped reads and skipped writes) so that the
    // loop can be unrolled by three, the period of the rolling row registers
#pragma unroll 3
    for (int it = 0; it < ROWS + 2; it++) {
        const int yy = y0 + it;
        const u32 g = gq[it % 3];
        gq[it % 3] = load_row(yy + 3);
        const s2 c0 = as_s2(__builtin_amdgcn_perm(0, g, sel01)), c1 = as_s2(__builtin_amdgcn_perm(0, g, sel23));
        // every pixel value the wave uses comes from the loaded words, so the row is flat when
        // all words equal one word of four equal bytes
        const u32 val_c = (u32)__builtin_amdgcn_readfirstlane((int)g);
        const bool flat_c = val_c == __builtin_rotateleft32(val_c, 8) && __all(g == val_c);
        flat_run = flat_c ? (val_c == flat_val ? flat_run + 1 : 1) : 0;
        flat_val = val_c;
        // rows outside the image have magnitude 0 as well
        const bool mn_zero = flat_run >= 3 || yy - 1 < 0 || yy - 1 >= h;
        MagRow mn;
        u32 ax01, ax23, ay01, ay23, sg01, sg23;
        if (mn_zero) {
            // The row is known to be zero: the *_zero flags stand for the zeros and nothing reads
            // the gradients, so the registers are only declared written (no instruction).
            asm("" : "=v"(mn.m01), "=v"(mn.m23), "=v"(mn.nL), "=v"(mn.nR));
            asm("" : "=v"(ax01), "=v"(ax23), "=v"(ay01), "=v"(ay23), "=v"(sg01), "=v"(sg23));
        } else {
            // ---- gradients and magnitude of row yy-1 (gray rows yy-2, yy-1, yy)
            const s2 vs0 = a0 + bq0 + bq0 + c0, vs1 = a1 + bq1 + bq1 + c1;
            const s2 vd0 = c0 - a0, vd1 = c1 - a1;
            const u32 vsL = from_lane_below(as_u32(vs1)), vsR = from_lane_above(as_u32(vs0));
            const u32 vdL = from_lane_below(as_u32(vd1)), vdR = from_lane_above(as_u32(vd0));
            const s2 vsm10 = as_s2(__builtin_amdgcn_alignbit(as_u32(vs0), vsL, 16));          // columns -1, 0
            const s2 vs12 = as_s2(__builtin_amdgcn_alignbit(as_u32(vs1), as_u32(vs0), 16));   // columns 1, 2
            const s2 vs34 = as_s2(__builtin_amdgcn_alignbit(vsR, as_u32(vs1), 16));           // columns 3, 4
            const s2 vdm10 = as_s2(__builtin_amdgcn_alignbit(as_u32(vd0), vdL, 16));
            const s2 vd12 = as_s2(__builtin_amdgcn_alignbit(as_u32(vd1), as_u32(vd0), 16));
            const s2 vd34 = as_s2(__builtin_amdgcn_alignbit(vdR, as_u32(vd1), 16));
            const s2 dx01 = vs12 - vsm10, dx23 = vs34 - vs12;
            const s2 dy01 = vdm10 + vd0 + vd0 + vd12, dy23 = vd12 + vd1 + vd1 + vd34;
            const s2 adx01 = __builtin_elementwise_abs(dx01), adx23 = __builtin_elementwise_abs(dx23);
            const s2 ady01 = __builtin_elementwise_abs(dy01), ady23 = __builtin_elementwise_abs(dy23);
            ax01 = as_u32(adx01); ax23 = as_u32(adx23); ay01 = as_u32(ady01); ay23 = as_u32(ady23);
            sg01 = as_u32(dx01) ^ as_u32(dy01);
            sg23 = as_u32(dx23) ^ as_u32(dy23);
            if (QUEUE) {
                // fewer registers across the loop: the high bytes of the four signs in one word (pixel k in byte k),
                // and |dy| is recomputed as magnitude - |dx| where it is needed
                sg01 = __builtin_amdgcn_perm(sg23, sg01, 0x07050301u);
                sg23 = 0;
                ay01 = 0;
                ay23 = 0;
            }
            mn.m01 = as_u32(adx01 + ady01) & cm01;
            mn.m23 = as_u32(adx23 + ady23) & cm23;
            mn.nL = from_lane_below(mn.m23);
            mn.nR = from_lane_above(mn.m01);
        }
        // ---- non-maximum suppression of row yy-2 (magnitude rows yy-3 = mp, yy-2 = mc, yy-1 = mn)
        const int ry = yy - 2;
        if (it >= 2 && ry < y1) {
            u32 codes = 0;
            bool queued = false;  // this lane's pixels of the row are on the stack
            // any magnitude of the row above the low threshold?  (per 16-bit half: max(m - low, 0))
            const bool cand = !mc_zero && (pk_sub_sat_u16(mc.m01, low2) | pk_sub_sat_u16(mc.m23, low2)) != 0 && (!QUEUE || writer);
            const unsigned long long cmask = mc_zero ? 0ull : __ballot(cand);
            const u32 ncand = (u32)__popcll(cmask);  // wave-uniform
            if (ncand != 0 && (!QUEUE || ncand >= (u32)NQ_DENSE)) {
                const MagRow zero = {0, 0, 0, 0};
                const MagRow mpv = mp_zero ? zero : mp, mnv = mn_zero ? zero : mn;
                // neighbour pairs: columns (-1,0), (1,2), (3,4) of the three rows
                const u32 cA = __builtin_amdgcn_alignbit(mc.m01, mc.nL, 16), cB = __builtin_amdgcn_alignbit(mc.m23, mc.m01, 16),
                          cC = __builtin_amdgcn_alignbit(mc.nR, mc.m23, 16);
                const u32 pA = __builtin_amdgcn_alignbit(mpv.m01, mpv.nL, 16), pB = __builtin_amdgcn_alignbit(mpv.m23, mpv.m01, 16),
                          pC = __builtin_amdgcn_alignbit(mpv.nR, mpv.m23, 16);
                const u32 nA = __builtin_amdgcn_alignbit(mnv.m01, mnv.nL, 16), nB = __builtin_amdgcn_alignbit(mnv.m23, mnv.m01, 16),
                          nC = __builtin_amdgcn_alignbit(mnv.nR, mnv.m23, 16);
                // (QUEUE: |dy| = magnitude - |dx|; a column outside the image has magnitude 0 and yields 0 whatever |dy| is)
                const u32 ayd01 = QUEUE ? as_u32(as_s2(mc.m01) - as_s2(axc01)) : ayc01, ayd23 = QUEUE ? as_u32(as_s2(mc.m23) - as_s2(axc23)) : ayc23;
                const u32 sgd01 = QUEUE ? __builtin_amdgcn_perm(0, sgc01, 0x01010000u) : sgc01, sgd23 = QUEUE ? __builtin_amdgcn_perm(0, sgc01, 0x03030202u) : sgc23;
                const u32 k01 = nms_pair(mc.m01, axc01, ayd01, sgd01, cA, cB, pA, mpv.m01, pB, nA, mnv.m01, nB, low2, high2, span2);
                const u32 k23 = nms_pair(mc.m23, axc23, ayd23, sgd23, cB, cC, pB, mpv.m23, pC, nB, mnv.m23, nC, low2, high2, span2);
                codes = __builtin_amdgcn_perm(k23, k01, 0x06040200u);  // low byte of each 16-bit code
            } else if (QUEUE && ncand != 0) {
                // row within the band, lane, "row above / below is zero" and the four signs of dx ^ dy in one word
                const u32 row_info = (u32)__builtin_amdgcn_readfirstlane((int)((u32)(it - 2) | (mp_zero ? 1u << 16 : 0u) | (mn_zero ? 1u << 17 : 0u)));
                if (cand) {
                    const u32 pos = pending + __builtin_amdgcn_mbcnt_hi((u32)(cmask >> 32), __builtin_amdgcn_mbcnt_lo((u32)cmask, 0u));
                    const u32 sgb = sgc01 & 0x80808080u;
                    // column -1 (high half of nL) and column 4 (low half of nR) in one word
                    q[0][pos] = make_uint4(mc.m01, mc.m23, __builtin_amdgcn_perm(mc.nL, mc.nR, 0x07060100u), axc01);
                    q[1][pos] = make_uint4(mp.m01, mp.m23, __builtin_amdgcn_perm(mp.nL, mp.nR, 0x07060100u), axc23);
                    q[2][pos] = make_uint4(mn.m01, mn.m23, __builtin_amdgcn_perm(mn.nL, mn.nR, 0x07060100u), sgb | lane_info | row_info);
                    queued = true;
                }
                pending += ncand;
            }
            if (writer && (ZFILL || !queued)) {
                *(u32*)(dimg + (size_t)((u32)ry * dpitch + (u32)x)) = codes;
                // any WEAK byte (value 1, top bit clear) flags the 128x32 tile for the hysteresis
                if (codes & ~(codes >> 7) & 0x01010101u) weak_row0[(u32)(ry / TH) * (u32)tiles_x] = 1;
            }
            if (QUEUE && pending >= 64) {
                pop(pending - 64, 64);
                pending -= 64;
            }
            if (EMIT) {
                // strong pixels (255) of this row -> the wave's buffer, in lane order per pixel column
                const u32 st4 = writer ? ((codes >> 7) & 0x01010101u) : 0u;
                if (__any(st4 != 0)) {
                    // in x order: pixels of lower lanes first, then this lane's own from left to right
                    const unsigned long long m0 = __ballot(st4 & 0x1u), m1 = __ballot(st4 & 0x100u),
                                             m2 = __ballot(st4 & 0x10000u), m3 = __ballot(st4 & 0x1000000u);
                    const u32 total = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
                    if (fill + total > 256) flush(b);
                    const unsigned long long lt = (1ull << lane) - 1ull;
                    u32 pos = fill + __popcll(m0 & lt) + __popcll(m1 & lt) + __popcll(m2 & lt) + __popcll(m3 & lt);
                    const u32 xy = ((u32)ry << 16) | (u32)x;
                    if (st4 & 0x1u) pbuf[pos++] = xy;
                    if (st4 & 0x100u) pbuf[pos++] = xy + 1;
                    if (st4 & 0x10000u) pbuf[pos++] = xy + 2;
                    if (st4 & 0x1000000u) pbuf[pos++] = xy + 3;
                    fill += total;
                }
            }
        }
        // ---- roll the window
        a0 = bq0; a1 = bq1; bq0 = c0; bq1 = c1;
        mp = mc; mc = mn; mp_zero = mc_zero; mc_zero = mn_zero;
        axc01 = ax01; axc23 = ax23; ayc01 = ay01; ayc23 = ay23; sgc01 = sg01; sgc23 = sg23;
    }
    if (QUEUE && pending != 0) pop(0, pending);
    flush(b);
}

// ---- wide variant: whole aligned lines, optionally fused with BGR -> gray / threshold / bit plane ----------
// One wave walks a strip of 64 * PX pixels (PX = 2 NP pixels per lane: 512 or 1024 columns) downwards and
// writes every plane as whole aligned lines: a row piece of a wave is 512 or 1024 contiguous bytes that start
// at a multiple of their length, so no cache line is shared with the neighbouring strip (the 248-byte pieces
// of k_canny_stream are).  The price is the halo: the two columns left and right of the strip are loaded and
// converted redundantly by lanes 0 and 63 (a pair of pixels each), and the band's two rows above and below
// are read again by the neighbouring bands.
//   BGR_IN: the source is the interleaved BGR image; gray, thresholded plane and bit plane of the band's own
//           rows are written on the way (the arithmetic of k_bgr2gray_flat), so the gray plane is never read
//           back.  Otherwise the source is a gray plane (cv2.Canny alone).
// Non-maximum suppression is always deferred: a group of four pixels with a magnitude above the low threshold
// goes onto the wave's LDS stack (the entry format of k_canny_stream plus the group's number) and 64 entries
// at a time are suppressed with all lanes busy.  A row's own store writes zeros everywhere except into the
// queued groups, whose four bytes are written once, by the lane that takes the entry.
constexpr int WQ_CAP = 128;  // at most 63 waiting + 64 new entries per group of four pixels


struct WideArgs {
    Plane src, gray, bin, dst;
    u32* bits;
    size_t bpw, bsw;
    int w, h, strips_x, bands;
    u32 nwaves;
    int low, high, tiles_x, tiles_y;
    u8* has_weak;
    // threshold: "above" <=> gray + k >= 256 with k = 255 - t in 0 .. 255 (0: never)
    u32 k7;    // (k & 0x7f) in every byte
    u32 kb;    // 0x80808080 when k >= 128, else 0
    u32 a4, b4, nz_a, nz_b;  // value above / below in every byte; 0xffff when that value is non-zero
    ClearList clr;
};

// Q15 luma, scaled by two so that the gray value is byte 2 of the result: 2 * 3735 = 29 * 256 + 46,
// 2 * 19235 = 150 * 256 + 70, 2 * 9798 = 76 * 256 + 140; `sh` = byte position of B inside `px` (0 or 1).
__device__ __forceinline__ u32 luma2_at(u32 px, int sh)
{
    const u32 hi = __builtin_amdgcn_udot4(px, sh ? 0x4c961d00u : 0x004c961du, 0u, false);
    const u32 lo = __builtin_amdgcn_udot4(px, sh ? 0x8c462e00u : 0x008c462eu, 32768u, false);
    return (hi << 8) + lo;
}

// four bytes at any byte address (global memory is accessed in the unaligned mode, as by the loads of k_canny_stream)
struct __attribute__((packed, aligned(4))) U1 { u32 x; };

template <bool BGR_IN, int NP, int ROWS>
// (16 pixels per lane need about 170 registers: three waves per SIMD are asked for, which costs a few spills
// and measures faster than two waves)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NP == 8 ? 3 : 4, 8))) void k_canny_wide(WideArgs A)
{
    constexpr int PX = 2 * NP, NQ = NP / 2, ND = PX / 4, NB = BGR_IN ? 3 * ND : ND, COLS = 64 * PX, NIT = ROWS + 4;
    static_assert(NP == 4 || NP == 8, "8 or 16 pixels per lane");
    static_assert(NIT % 3 == 0 && ROWS <= 125, "the row loop is unrolled by three; 7 bits of row in a queue entry");
    __shared__ uint4 s_q[4][3][WQ_CAP];
    // per wave, strip side (0 = left, 1 = right) and gray row of the band: what the column next to the strip
    // contributes (.x = vertical sum | vertical difference << 16 and .y = magnitude | "the row's halo pixels equal
    // their neighbour inside the strip" << 16; sum, difference and magnitude belong to the row before)
    __shared__ uint2 s_h[4][2][NIT];
    if (A.clr.n[0] | A.clr.n[1] | A.clr.n[2]) {  // (uniform) the first workgroup zeroes the scratch of the stages that follow
        if (blockIdx.x == 0) {
            for (int a = 0; a < 3; a++)
                for (u32 i = threadIdx.x; i < A.clr.n[a]; i += 256) A.clr.p[a][i] = 0;
        }
    }
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 wid = xcd_contiguous_id(blockIdx.x, gridDim.x) * 4u + (u32)wv;
    if (wid >= A.nwaves) return;
    const int lane = threadIdx.x & 63;
    const int w = A.w, h = A.h;
    const u32 t_ = wid / (u32)A.strips_x;
    const int sx = (int)(wid - t_ * (u32)A.strips_x);
    const int b = (int)(t_ / (u32)A.bands), band = (int)(t_ - (u32)b * (u32)A.bands);
    const int X0 = sx * COLS;  // first column of the strip (wave-uniform)
    const int x0 = X0 + PX * lane;
    const int y0 = band * ROWS, y1 = min(y0 + ROWS, h);
    // the image width is a multiple of PX (checked by the host): a lane lies inside the image or beyond it
    const bool live = x0 < w;
    const bool wave_edge = X0 + COLS > w;  // (uniform) the strip reaches beyond the image
    const u8* const simg = A.src.p + (size_t)b * A.src.stride;
    const u32 spitch = (u32)A.src.pitch;
    const int xc = live ? x0 : w - PX;  // a lane beyond the image loads the row's last pixels and replicates the last one
    const u32 xoff = (u32)xc * (BGR_IN ? 3u : 1u);
    struct Raw {
        u32 d[NB];
    };
    auto load_row = [&](int gy) -> Raw {
        Raw r;
        const u32 row = (u32)clampi(gy, 0, h - 1) * spitch;
        const u8* p = simg + (size_t)(row + xoff);
        if (NB % 4 == 0) {
#pragma unroll
            for (int k = 0; k < NB / 4; k++) {
                const uint4 v = ((const uint4*)p)[k];
                r.d[4 * k] = v.x; r.d[4 * k + 1] = v.y; r.d[4 * k + 2] = v.z; r.d[4 * k + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NB / 2; k++) {
                const uint2 v = ((const uint2*)p)[k];
                r.d[2 * k] = v.x; r.d[2 * k + 1] = v.y;
            }
        }
        return r;
    };
    // the first rows of the strip are on their way while the halo table is made
    Raw buf[3];
#pragma unroll
    for (int k = 0; k < 3; k++) buf[k] = load_row(y0 - 2 + k);
    // ---- the columns next to the strip, for all rows of the band at once: lane = row.  A lane reads the three
    // pixels around the strip border of its row (left: columns X0 - 2, X0 - 1, X0; right: Xe - 1, Xe, Xe + 1 with
    // Xe = the column behind the strip's last one inside the image; replicated at the image border), the rows
    // above come from the lanes below, and the middle column's Sobel terms go to s_h.
    {
        constexpr int NPASS = (NIT + 61) / 62;
        const int Xe = min(X0 + COLS, w);
        // raw words of the six pixels of every pass first (all loads in flight together), then the arithmetic
        u32 raw[NPASS][6];
        int shf[6];
        u32 coff[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int cc = clampi(k < 3 ? X0 - 2 + k : Xe - 4 + k, 0, w - 1);
            const int o = BGR_IN ? max(3 * cc - 1, 0) : cc;  // (the 4 bytes read end inside the row's last pixel)
            coff[k] = (u32)o;
            shf[k] = BGR_IN ? 8 * (3 * cc - o) : 0;
        }
#pragma unroll
        for (int t = 0; t < NPASS; t++) {
            const int j = 62 * t + lane - 2;  // gray row of the band this lane holds (rows j - 1, j - 2: the lanes below)
            const u32 rowoff = (u32)clampi(y0 - 2 + j, 0, h - 1) * spitch;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (BGR_IN) raw[t][k] = (*(const U1*)(simg + (size_t)(rowoff + coff[k]))).x;
                else raw[t][k] = simg[(size_t)(rowoff + coff[k])];
            }
        }
#pragma unroll
        for (int t = 0; t < NPASS; t++) {
            const int j = 62 * t + lane - 2;
#pragma unroll
            for (int side = 0; side < 2; side++) {
                u32 gC[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const u32 rw = raw[t][3 * side + k];
                    gC[k] = BGR_IN ? (luma2_at(rw >> shf[3 * side + k], 0) >> 16) & 0xffu : rw;
                }
                const u32 pk = gC[0] | (gC[1] << 8) | (gC[2] << 16);
                const u32 pB = (u32)__builtin_amdgcn_update_dpp(0, (int)pk, 0x138, 0xf, 0xf, false);  // row j - 1
                const u32 pA = (u32)__builtin_amdgcn_update_dpp(0, (int)pB, 0x138, 0xf, 0xf, false);  // row j - 2
                int vs[3], vd[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int a_ = (int)((pA >> (8 * k)) & 0xffu), b_ = (int)((pB >> (8 * k)) & 0xffu), c_ = (int)gC[k];
                    vs[k] = a_ + 2 * b_ + c_;
                    vd[k] = c_ - a_;
                }
                const int dx = vs[2] - vs[0], dy = vd[0] + 2 * vd[1] + vd[2];
                // magnitude is 0 outside the image
                const bool inside = side ? Xe < w : X0 > 0;
                const u32 m = inside ? (u32)(abs(dx) + abs(dy)) : 0u;
                const u32 same = (gC[0] == gC[1] && gC[1] == gC[2]) ? 1u : 0u;
                if (lane >= 2 && j < NIT) s_h[wv][side][j] = make_uint2(((u32)vs[1] & 0xffffu) | ((u32)vd[1] << 16), m | (same << 16));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const uint2* const hrow = s_h[wv][lane == 63 ? 1 : 0];  // (lanes 1 .. 62 read the left side's entry and do not use it)
    const bool isE = lane == 0 || lane == 63;

    const u32 lane_mask = live ? 0xffffffffu : 0u;
    u8* const dimg = A.dst.p + (size_t)b * A.dst.stride;
    u8* const gimg = BGR_IN ? A.gray.p + (size_t)b * A.gray.stride : nullptr;
    u8* const timg = BGR_IN && A.bin.p ? A.bin.p + (size_t)b * A.bin.stride : nullptr;
    u8* const bimg = BGR_IN && A.bits ? (u8*)(A.bits + (size_t)b * A.bsw) : nullptr;
    const u32 dpitch = (u32)A.dst.pitch, gpitch = BGR_IN ? (u32)A.gray.pitch : 0u, tpitch = timg ? (u32)A.bin.pitch : 0u;
    u8* const weak_img = A.has_weak + (size_t)b * A.tiles_y * A.tiles_x;
    uint4(*const q)[WQ_CAP] = s_q[wv];
    u32 pending = 0;  // (uniform) entries on the stack
    const u32 low2 = (u32)__builtin_amdgcn_readfirstlane(min(max(A.low, 0), 0x7fff) * 0x00010001);
    const u32 high2 = (u32)__builtin_amdgcn_readfirstlane(min(max(A.high, 0), 0x7fff) * 0x00010001);
    const u32 span2 = (u32)__builtin_amdgcn_readfirstlane((int)((255u - WEAK) * 0x00010001u));

    // entries first .. first + count - 1 (count <= 64) leave the stack: lane i suppresses entry first + i
    auto pop = [&](u32 first, u32 count) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((u32)lane < count) {
            const uint4 e0 = q[0][first + lane], e1 = q[1][first + lane], e2 = q[2][first + lane];
            const u32 info = e2.w;
            const bool pz = (info & (1u << 16)) != 0, nz = (info & (1u << 17)) != 0;  // row above / below known to be zero
            const u32 c01 = e0.x, c23 = e0.y, cO = e0.z;
            const u32 p01 = pz ? 0u : e1.x, p23 = pz ? 0u : e1.y, pO = pz ? 0u : e1.z;
            const u32 n01 = nz ? 0u : e2.x, n23 = nz ? 0u : e2.y, nO = nz ? 0u : e2.z;
            // xO: column -1 in the high half, column 4 in the low half
            const u32 cA = __builtin_amdgcn_alignbit(c01, cO, 16), cB = __builtin_amdgcn_alignbit(c23, c01, 16),
                      cC = __builtin_amdgcn_alignbit(cO, c23, 16);
            const u32 pA = __builtin_amdgcn_alignbit(p01, pO, 16), pB = __builtin_amdgcn_alignbit(p23, p01, 16),
                      pC = __builtin_amdgcn_alignbit(pO, p23, 16);
            const u32 nA = __builtin_amdgcn_alignbit(n01, nO, 16), nB = __builtin_amdgcn_alignbit(n23, n01, 16),
                      nC = __builtin_amdgcn_alignbit(nO, n23, 16);
            // |dx| with the sign of dx ^ dy in bit 15 of each half; |dy| = magnitude - |dx|
            const u32 ax01 = e0.w & 0x7fff7fffu, ax23 = e1.w & 0x7fff7fffu;
            const u32 ay01 = as_u32(as_s2(c01) - as_s2(ax01)), ay23 = as_u32(as_s2(c23) - as_s2(ax23));
            const u32 k01 = nms_pair(c01, ax01, ay01, e0.w, cA, cB, pA, p01, pB, nA, n01, nB, low2, high2, span2);
            const u32 k23 = nms_pair(c23, ax23, ay23, e1.w, cB, cC, pB, p23, pC, nB, n23, nC, low2, high2, span2);
            const u32 codes = __builtin_amdgcn_perm(k23, k01, 0x06040200u);
            const u32 ey = (u32)y0 + (info & 0x7fu), ex = (u32)X0 + (u32)PX * ((info >> 8) & 0x3fu) + 4u * ((info >> 18) & 3u);
            *(u32*)(dimg + (size_t)(ey * dpitch + ex)) = codes;
            if (codes & ~(codes >> 7) & 0x01010101u) weak_img[(ey / TH) * (u32)A.tiles_x + ex / TW] = 1;
        }
        __builtin_amdgcn_wave_barrier();
    };

    // rolling state: gray rows gy-2 (a), gy-1 (bq) as 16-bit pairs; magnitude rows gy-3 (mp), gy-2 (mc) with the
    // neighbouring lanes' columns; |dx| (sign of dx ^ dy in bit 15) of row gy-2
    u32 a[NP], bq[NP];
    u32 mp[NP], mc[NP], mpL = 0, mpR = 0, mcL = 0, mcR = 0, axc[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) a[p] = bq[p] = mp[p] = mc[p] = axc[p] = 0;
    int flat_run = 0;  // consecutive flat rows of value flat_val ending at the newest row
    u32 flat_val = 0;
    bool mp_zero = true, mc_zero = true;
    const u32 lane_info = (u32)lane << 8;

#pragma unroll 3
    for (int it = 0; it < NIT; it++) {
        const int gy = y0 - 2 + it;
        // (uniform) past the band's last row nothing is left to do; the trip count stays fixed so that the loop can be
        // unrolled by three, the period of the rolling registers
        if (gy - 2 < y1) {
        const Raw cur = buf[it % 3];
        buf[it % 3] = load_row(gy + 3);
        const uint2 hv = hrow[it];
        const bool own_row = it >= 2 && gy < y1;  // (uniform) a row of this band: its planes are written
        // ---- the lane's pixels of gray row gy as bytes (g) and as 16-bit pairs (c)
        u32 g[ND], c[NP];
        bool flat_c;
        u32 val_c;
        {
            const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)cur.d[0]);
            u32 diff = d0 ^ __builtin_rotateleft32(d0, 8);
#pragma unroll
            for (int k = 0; k < NB; k++) diff |= cur.d[k] ^ d0;
            if (isE) diff |= ~hv.y & 0x10000u;  // the pixels beside the strip differ from it
            flat_c = __all(diff == 0);
            val_c = d0 & 0xffu;  // (B = G = R = v gives gray v)
        }
        if (flat_c) {
            const u32 v4 = val_c * 0x01010101u, v2 = val_c * 0x00010001u;
#pragma unroll
            for (int k = 0; k < ND; k++) g[k] = v4;
#pragma unroll
            for (int p = 0; p < NP; p++) c[p] = v2;
        } else {
            if (BGR_IN) {
                u32 t2[PX];
#pragma unroll
                for (int k = 0; k < PX; k++) {
                    const int qd = (3 * k) >> 2, sb = (3 * k) & 3;
                    // pixel k starts at byte sb of dword qd: sb = 0: as is; 1: coefficients one byte up; 2, 3: across two dwords
                    if (sb == 0) t2[k] = luma2_at(cur.d[qd], 0);
                    else if (sb == 1) t2[k] = luma2_at(cur.d[qd], 1);
                    else t2[k] = luma2_at(__builtin_amdgcn_alignbyte(cur.d[qd + 1], cur.d[qd], (u32)sb), 0);
                }
#pragma unroll
                for (int p = 0; p < NP; p++) c[p] = __builtin_amdgcn_perm(t2[2 * p + 1], t2[2 * p], 0x0c060c02u);
#pragma unroll
                for (int k = 0; k < ND; k++) g[k] = __builtin_amdgcn_perm(c[2 * k + 1], c[2 * k], 0x06040200u);
            } else {
#pragma unroll
                for (int k = 0; k < ND; k++) g[k] = cur.d[k];
#pragma unroll
                for (int p = 0; p < NP; p++) c[p] = __builtin_amdgcn_perm(0u, g[p >> 1], (p & 1) ? PAIR_SEL(2, 3) : PAIR_SEL(0, 1));
            }
            if (wave_edge) {  // lanes beyond the right image border: the last pixel of the row, replicated
                const u32 rep4 = (g[ND - 1] >> 24) * 0x01010101u, rep2 = (g[ND - 1] >> 24) * 0x00010001u;
#pragma unroll
                for (int k = 0; k < ND; k++) g[k] = live ? g[k] : rep4;
#pragma unroll
                for (int p = 0; p < NP; p++) c[p] = live ? c[p] : rep2;
            }
        }
        // ---- gray, thresholded plane and bit plane of the band's own rows
        if (BGR_IN && own_row) {
            u32 tb[ND], above = 0;  // bit k of `above`: pixel k is above the threshold
            if (timg) {
#pragma unroll
                for (int k = 0; k < ND; k++) {
                    // carry out of every byte of g + k: into bit 7 of L from the low seven bits, then the majority with the top bits
                    const u32 L = (g[k] & 0x7f7f7f7fu) + A.k7;
                    const u32 R = (L & (g[k] | A.kb)) | (g[k] & A.kb);
                    const u32 b1 = (R >> 7) & 0x01010101u;  // 1 in every byte above the threshold
                    const u32 m = (b1 << 8) - b1;           // 0xff in those bytes
                    tb[k] = (m & A.a4) | (~m & A.b4);
                    above |= __builtin_amdgcn_udot4(b1, 0x08040201u, 0u, false) << (4 * k);
                }
            }
            if (live) {
                u8* const gd = gimg + (size_t)((u32)gy * gpitch + (u32)x0);
                if (ND == 4) *(uint4*)gd = make_uint4(g[0], g[1], g[2], g[ND - 1]);
                else *(uint2*)gd = make_uint2(g[0], g[1]);
                if (timg) {
                    u8* const td = timg + (size_t)((u32)gy * tpitch + (u32)x0);
                    if (ND == 4) *(uint4*)td = make_uint4(tb[0], tb[1], tb[2], tb[ND - 1]);
                    else *(uint2*)td = make_uint2(tb[0], tb[1]);
                }
            }
            if (bimg) {
                // Byte or halfword stores from all 64 lanes are slow: the bits of 16 (PX = 8) or 8 (PX = 16) neighbouring
                // lanes are gathered into one lane by lane moves inside each row of 16 lanes, and that lane stores 16 bytes,
                // so that the wave's 64 or 128 bytes of the bit-plane row leave in one instruction of 4 or 8 lanes.
                // (lanes beyond the image contribute the zero bits of the row's last words)
                const u32 mybits = ((above & A.nz_a) | (~above & A.nz_b)) & (live ? (PX == 16 ? 0xffffu : 0xffu) : 0u);
                u32 w0;
                if (PX == 16) {
                    w0 = mybits | ((u32)__builtin_amdgcn_update_dpp(0, (int)mybits, 0x101, 0xf, 0xf, true) << 16);  // lanes 2k: bits of 2k, 2k+1
                } else {
                    const u32 h0 = mybits | ((u32)__builtin_amdgcn_update_dpp(0, (int)mybits, 0x101, 0xf, 0xf, true) << 8);
                    w0 = h0 | ((u32)__builtin_amdgcn_update_dpp(0, (int)h0, 0x102, 0xf, 0xf, true) << 16);          // lanes 4k: bits of 4k .. 4k+3
                }
                constexpr int S1 = PX == 16 ? 2 : 4;  // lanes per dword
                const u32 w1 = (u32)__builtin_amdgcn_update_dpp(0, (int)w0, 0x100 + S1, 0xf, 0xf, true);
                const u32 w2 = (u32)__builtin_amdgcn_update_dpp(0, (int)w0, 0x100 + 2 * S1, 0xf, 0xf, true);
                const u32 w3 = (u32)__builtin_amdgcn_update_dpp(0, (int)w1, 0x100 + 2 * S1, 0xf, 0xf, true);
                if ((lane & (4 * S1 - 1)) == 0)
                    *(uint4*)(bimg + (size_t)gy * (A.bpw * 4) + (size_t)(x0 >> 3)) = make_uint4(w0, w1, w2, w3);
            }
        }
        // ---- flat rows: three equal ones in a row have zero gradients
        flat_run = flat_c ? (val_c == flat_val && flat_run > 0 ? flat_run + 1 : 1) : 0;
        flat_val = val_c;
        const bool mn_zero = it < 2 || flat_run >= 3 || gy - 1 < 0 || gy - 1 >= h;
        u32 mn[NP], mnL, mnR, axn[NP];
        if (mn_zero) {
            // known to be zero: the flag stands for the values, the registers are only declared written
#pragma unroll
            for (int p = 0; p < NP; p++) asm("" : "=v"(mn[p]), "=v"(axn[p]));
            asm("" : "=v"(mnL), "=v"(mnR));
        } else {
            // ---- gradients and magnitude of row gy-1 (gray rows gy-2 = a, gy-1 = bq, gy = c)
            u32 vs[NP], vd[NP];
#pragma unroll
            for (int p = 0; p < NP; p++) {
                vs[p] = as_u32(as_s2(a[p]) + as_s2(bq[p]) + as_s2(bq[p]) + as_s2(c[p]));
                vd[p] = as_u32(as_s2(c[p]) - as_s2(a[p]));
            }
            // the column beside the strip (lanes 0 and 63), in both halves of a word
            const u32 vsH = __builtin_amdgcn_perm(0u, hv.x, 0x01000100u), vdH = __builtin_amdgcn_perm(0u, hv.x, 0x03020302u);
            const u32 mH = __builtin_amdgcn_perm(0u, hv.y, 0x01000100u);
            // columns -1 (high half of xL; lane 0: from the table) and PX (low half of xR; lane 63: from the table)
            const u32 vsL = (u32)__builtin_amdgcn_update_dpp((int)vsH, (int)vs[NP - 1], 0x138, 0xf, 0xf, false);
            const u32 vsR = (u32)__builtin_amdgcn_update_dpp((int)vsH, (int)vs[0], 0x130, 0xf, 0xf, false);
            const u32 vdL = (u32)__builtin_amdgcn_update_dpp((int)vdH, (int)vd[NP - 1], 0x138, 0xf, 0xf, false);
            const u32 vdR = (u32)__builtin_amdgcn_update_dpp((int)vdH, (int)vd[0], 0x130, 0xf, 0xf, false);
            // odd pairs: o[k] = columns 2k-1, 2k
            u32 os[NP + 1], od[NP + 1];
            os[0] = __builtin_amdgcn_alignbit(vs[0], vsL, 16);
            od[0] = __builtin_amdgcn_alignbit(vd[0], vdL, 16);
#pragma unroll
            for (int k = 1; k < NP; k++) {
                os[k] = __builtin_amdgcn_alignbit(vs[k], vs[k - 1], 16);
                od[k] = __builtin_amdgcn_alignbit(vd[k], vd[k - 1], 16);
            }
            os[NP] = __builtin_amdgcn_alignbit(vsR, vs[NP - 1], 16);
            od[NP] = __builtin_amdgcn_alignbit(vdR, vd[NP - 1], 16);
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const s2 dx = as_s2(os[p + 1]) - as_s2(os[p]);
                const s2 dy = as_s2(od[p]) + as_s2(od[p + 1]) + as_s2(vd[p]) + as_s2(vd[p]);
                const s2 adx = __builtin_elementwise_abs(dx), ady = __builtin_elementwise_abs(dy);
                mn[p] = as_u32(adx + ady);
                axn[p] = as_u32(adx) | ((as_u32(dx) ^ as_u32(dy)) & 0x80008000u);
            }
            if (wave_edge) {  // magnitude is 0 outside the image
#pragma unroll
                for (int p = 0; p < NP; p++) mn[p] &= lane_mask;
            }
            mnL = (u32)__builtin_amdgcn_update_dpp((int)mH, (int)mn[NP - 1], 0x138, 0xf, 0xf, false);
            mnR = (u32)__builtin_amdgcn_update_dpp((int)mH, (int)mn[0], 0x130, 0xf, 0xf, false);
        }
        // ---- row ry = gy - 2: candidates go onto the stack, everything else is written as zero
        const int ry = gy - 2;
        if (it >= 4) {  // (ry < y1 by the loop's exit test)
            bool cq[NQ];
            bool anyc = false;
#pragma unroll
            for (int k = 0; k < NQ; k++) {
                cq[k] = !mc_zero && (pk_sub_sat_u16(mc[2 * k], low2) | pk_sub_sat_u16(mc[2 * k + 1], low2)) != 0;
                anyc = anyc || cq[k];
            }
            u8* const ed = dimg + (size_t)((u32)ry * dpitch + (u32)x0);
            if (!mc_zero && __any(anyc)) {
                const u32 row_info = (u32)__builtin_amdgcn_readfirstlane((int)((u32)(it - 4) | (mp_zero ? 1u << 16 : 0u) | (mn_zero ? 1u << 17 : 0u)));
#pragma unroll
                for (int k = 0; k < NQ; k++) {
                    const unsigned long long cmask = __ballot(cq[k]);
                    if (cmask == 0) continue;  // (uniform)
                    if (cq[k]) {
                        const u32 pos = pending + __builtin_amdgcn_mbcnt_hi((u32)(cmask >> 32), __builtin_amdgcn_mbcnt_lo((u32)cmask, 0u));
                        // columns -1 (high half) and 4 (low half) of the group in one word, per row
                        const u32 cl = k == 0 ? mcL : mc[2 * k - 1], cr = k == NQ - 1 ? mcR : mc[2 * k + 2 < NP ? 2 * k + 2 : 0];
                        const u32 pl = k == 0 ? mpL : mp[2 * k - 1], pr = k == NQ - 1 ? mpR : mp[2 * k + 2 < NP ? 2 * k + 2 : 0];
                        const u32 nl = k == 0 ? mnL : mn[2 * k - 1], nr = k == NQ - 1 ? mnR : mn[2 * k + 2 < NP ? 2 * k + 2 : 0];
                        q[0][pos] = make_uint4(mc[2 * k], mc[2 * k + 1], __builtin_amdgcn_perm(cl, cr, 0x07060100u), axc[2 * k]);
                        q[1][pos] = make_uint4(mp[2 * k], mp[2 * k + 1], __builtin_amdgcn_perm(pl, pr, 0x07060100u), axc[2 * k + 1]);
                        q[2][pos] = make_uint4(mn[2 * k], mn[2 * k + 1], __builtin_amdgcn_perm(nl, nr, 0x07060100u),
                                               lane_info | row_info | ((u32)k << 18));
                    }
                    pending += (u32)__popcll(cmask);
                    if (pending >= 64) {
                        pop(pending - 64, 64);
                        pending -= 64;
                    }
                }
                if (live) {
                    if (!anyc) {
                        if (ND == 4) *(uint4*)ed = make_uint4(0, 0, 0, 0);
                        else *(uint2*)ed = make_uint2(0, 0);
                    } else {
#pragma unroll
                        for (int k = 0; k < NQ; k++)
                            if (!cq[k]) *(u32*)(ed + 4 * k) = 0;
                    }
                }
            } else if (live) {
                if (ND == 4) *(uint4*)ed = make_uint4(0, 0, 0, 0);
                else *(uint2*)ed = make_uint2(0, 0);
            }
        }
        // ---- roll the window
#pragma unroll
        for (int p = 0; p < NP; p++) {
            a[p] = bq[p]; bq[p] = c[p];
            mp[p] = mc[p]; mc[p] = mn[p];
            axc[p] = axn[p];
        }
        mpL = mcL; mpR = mcR; mcL = mnL; mcR = mnR;
        mp_zero = mc_zero; mc_zero = mn_zero;
        }
    }
    if (pending != 0) pop(0, pending);
}

constexpr int HT_THREADS = 512;      // 8 waves per workgroup (16 when there are few workgroups)
constexpr int HT_MAX_TILES = 4096;   // tiles per region (one workgroup)

// bit 0 of each of the 4 bytes of x -> 4 adjacent bits
__device__ __forceinline__ u32 gather4(u32 x) { return ((x & 0x01010101u) * 0x10204080u) >> 28; }

struct TileBits {
    u64 S, W;  // edge / WEAK masks of this lane's 64 pixels (bit i = pixel i)
};

// edge bytes of 64 consecutive pixels, as loaded (pieces at or beyond `nvalid` are not read and stay zero)
struct Raw64 {
    uint4 v[4];
};
__device__ __forceinline__ Raw64 load_raw64(const u8* p, int nvalid)
{
    Raw64 r;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        r.v[q] = make_uint4(0, 0, 0, 0);
        if (q * 16 < nvalid) r.v[q] = *(const uint4*)(p + 16 * q);
    }
    return r;
}
// -> masks.  Pixels at or beyond `nvalid` are empty.
__device__ __forceinline__ TileBits bits_of_raw64(const Raw64& r, int nvalid)
{
    TileBits t;
    t.S = 0;
    t.W = 0;
    if (nvalid <= 0) return t;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const u32 d[4] = {r.v[q].x, r.v[q].y, r.v[q].z, r.v[q].w};
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
__device__ __forceinline__ TileBits load_bits64(const u8* p, int nvalid)
{
    if (nvalid <= 0) {
        TileBits t;
        t.S = 0;
        t.W = 0;
        return t;
    }
    return bits_of_raw64(load_raw64(p, nvalid), nvalid);
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
// One region (rx, ry) of image b, by the whole workgroup.  any_mark (LDS, or nullptr): set when a tile of another
// region was marked for a revisit.
template <int NT>
__device__ __forceinline__ void hyst_region(Plane dst, int w, int h, int tiles_x, int tiles_y, int rt_x, int rt_y, int rx,
                                            int ry, int b, const CannyWs& W, const PointSink& sink, int seed_pending,
                                            int round_no, u32* any_mark)
{
    __shared__ u16 s_list[2][HT_MAX_TILES];
    __shared__ u32 s_inq[HT_MAX_TILES / 32];
    __shared__ u32 s_flag[HT_MAX_TILES / 32];
    __shared__ u32 s_cnt[2];
    __shared__ u32 s_cursor;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tx_base = rx * rt_x, ty_base = ry * rt_y;
    const int nx_t = min(rt_x, tiles_x - tx_base), ny_t = min(rt_y, tiles_y - ty_base);

    for (int i = tid; i < HT_MAX_TILES / 32; i += NT) {
        s_inq[i] = 0;
        s_flag[i] = 0;
    }
    if (tid == 0) {
        s_cnt[0] = 0;
        s_cnt[1] = 0;
        s_cursor = 0;
    }
    __syncthreads();
    for (int t = tid; t < nx_t * ny_t; t += NT) {
        const int lx = t % nx_t, ly = t / nx_t;
        const size_t gt = ((size_t)b * tiles_y + ty_base + ly) * tiles_x + tx_base + lx;
        if (W.has_weak[gt]) {
            atomicOr(&s_flag[t >> 5], 1u << (t & 31));
            // seed_pending: only the tiles an earlier launch left marked start on the list
            if (!seed_pending || W.pending_in[gt]) {
                const u32 pos = atomicAdd(&s_cnt[0], 1u);
                s_list[0][pos] = (u16)t;
            }
        }
    }
    __syncthreads();

    // lane = (64-pixel half, row): the rows of a half are neighbouring lanes, so the row above / below comes from a
    // one-lane wave shift (two DPP moves per 64-bit mask) and only one bit crosses to the other half per iteration
    const int r = lane & 31, hf = lane >> 5;
    int cur = 0;
    for (int round = 0; round < (1 << 24); round++) {
        const u32 n = s_cnt[cur];
        if (n == 0) break;
        const int nxt = cur ^ 1;
        // What a tile visit reads, fetched one visit ahead: while a wave resolves a tile, the loads of its next
        // one are under way (a visit is a chain of memory round trips otherwise).  Reading a neighbour's ring a
        // little earlier is harmless: a neighbour whose rim changes afterwards puts this tile on the next round's list.
        struct Visit {
            int t;        // tile within the region, -1 = none
            Raw64 own, ring;
            u32 side, ring_side;
        };
        auto fetch = [&]() -> Visit {
            Visit v;
            u32 i = 0;
            if (lane == 0) i = atomicAdd(&s_cursor, 1u);
            i = __shfl(i, 0);
            v.t = i < n ? (int)s_list[cur][i] : -1;
            v.side = v.ring_side = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) v.own.v[q] = v.ring.v[q] = make_uint4(0, 0, 0, 0);
            if (v.t < 0) return v;
            const int lx = v.t % nx_t, ly = v.t / nx_t;
            const int x0 = (tx_base + lx) * TW, y0 = (ty_base + ly) * TH;
            const int iy = y0 + r, xh = x0 + 64 * hf;
            if (iy < h && w - xh > 0) v.own = load_raw64(dst.row(b, iy) + xh, w - xh);
            const int xs = hf ? x0 + TW : x0 - 1;
            if (iy < h && xs >= 0 && xs < w) v.side = dst.row(b, iy)[xs];
            const int ry_ = r == 0 ? y0 - 1 : y0 + TH;
            if ((r == 0 || r == TH - 1) && ry_ >= 0 && ry_ < h) {
                if (w - xh > 0) v.ring = load_raw64(dst.row(b, ry_) + xh, w - xh);
                if (xs >= 0 && xs < w) v.ring_side = dst.row(b, ry_)[xs];
            }
            return v;
        };
        Visit nextv = fetch();
        for (;;) {
            const Visit v = nextv;
            if (v.t < 0) break;  // (wave-uniform)
            nextv = fetch();
            const int t = v.t;
            const int lx = t % nx_t, ly = t / nx_t;
            const int x0 = (tx_base + lx) * TW, y0 = (ty_base + ly) * TH;
            const int iy = y0 + r, xh = x0 + 64 * hf;
            // own 64 pixels
            TileBits me;
            me.S = me.W = 0;
            if (iy < h) me = bits_of_raw64(v.own, w - xh);
            // the edge bit just outside this lane's end of the row (ring column)
            const u32 side = v.side >> 7;
            // ring rows: lanes of row 0 hold the row above the tile, lanes of row 31 the row below
            u64 ringS = 0;
            u32 ring_side = 0;
            const int ry_ = r == 0 ? y0 - 1 : y0 + TH;
            if ((r == 0 || r == TH - 1) && ry_ >= 0 && ry_ < h) {
                ringS = bits_of_raw64(v.ring, w - xh).S;
                ring_side = v.ring_side >> 7;
            }
            // horizontally dilated ring row (constant during the iteration)
            u64 ringH;
            {
                const u32 pr = (u32)__shfl_xor((int)(hf ? (u32)(ringS & 1ull) : (u32)(ringS >> 63)), 32);
                const u64 lin = hf ? (u64)pr : (u64)ring_side;
                const u64 rin = hf ? (u64)ring_side : (u64)pr;
                ringH = ringS | (ringS << 1) | (ringS >> 1) | lin | (rin << 63);
            }
            const u64 S0 = me.S;
            u64 S = me.S;
            if (__any(me.W != 0)) {
                for (;;) {
                    const u64 S_in = S;
                    {   // runs of passable pixels (WEAK or edge) along the row are filled from their edge pixels in one
                        // step, upwards by the carries of M + S, downwards the same on the reversed words
                        const u64 M = me.W | S;
                        S |= M & ((M + S) ^ M ^ S);
                        const u64 Mr = __brevll(M), Sr = __brevll(S);
                        S |= __brevll(Mr & ((Mr + Sr) ^ Mr ^ Sr));
                    }
                    // the pixel next to this lane's inner end belongs to the other half of the row
                    const u32 pS = (u32)__shfl_xor((int)(hf ? (u32)(S & 1ull) : (u32)(S >> 63)), 32);
                    const u64 lin = hf ? (u64)pS : (u64)side;
                    const u64 rin = hf ? (u64)side : (u64)pS;
                    const u64 H = S | (S << 1) | (S >> 1) | lin | (rin << 63);
                    const u32 Hl = (u32)H, Hh = (u32)(H >> 32);
                    u64 up = (u64)(u32)__builtin_amdgcn_update_dpp(0, (int)Hl, 0x138, 0xf, 0xf, false) |
                             ((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)Hh, 0x138, 0xf, 0xf, false) << 32);  // lane - 1
                    u64 dn = (u64)(u32)__builtin_amdgcn_update_dpp(0, (int)Hl, 0x130, 0xf, 0xf, false) |
                             ((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)Hh, 0x130, 0xf, 0xf, false) << 32);  // lane + 1
                    if (r == 0) up = ringH;
                    if (r == TH - 1) dn = ringH;
                    const u64 Sn = S | (me.W & (H | up | dn));
                    const bool ch = Sn != S_in;
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
            if (sink.pts && __any(prom != 0)) {
                // promoted pixels join the image's edge-pixel list: one atomic per tile visit
                const u32 cnt = (u32)__popcll(prom);
                u32 scan = cnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const u32 tt = __shfl_up(scan, o);
                    if (lane >= o) scan += tt;
                }
                const u32 total = __shfl(scan, 63);
                u32 base = 0;
                if (lane == 63) base = (u32)atomicAdd(&sink.counts[b], (int32_t)total);
                base = __shfl(base, 63);
                u32 pos = base + scan - cnt;
                u32* out = sink.pts + (size_t)b * sink.stride;
                u64 pm = prom;
                while (pm) {
                    const int bit = __ffsll((long long)pm) - 1;
                    pm &= pm - 1;
                    if (pos < sink.cap) out[pos] = ((u32)iy << 16) | (u32)(xh + bit);
                    pos++;
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
                        // belongs to another workgroup's region: left to a later launch
                        if (W.pending) W.pending[((size_t)b * tiles_y + gty) * tiles_x + gtx] = 1;
                        if (round_no >= 0) atomicMax(W.again, (u32)round_no + 1u);
                        if (any_mark) *any_mark = 1u;
                    }
                }
            }
        }
        __syncthreads();  // also waits for this round's stores and atomics
        if (tid == 0) {
            s_cnt[cur] = 0;
            s_cursor = 0;
        }
        for (int i = tid; i < HT_MAX_TILES / 32; i += NT) s_inq[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale L1 lines before re-reading tiles
        __syncthreads();
        cur = nxt;
    }
}


// A region of at most 8 x 8 tiles (1024 x 256 pixels) with its state in LDS: the edge and WEAK bits of the whole region
// (two bit planes, 64 pixels per word, the edge plane with a ring of one pixel around the region) are built once from
// the bytes, the worklist rounds of hyst_region then run on LDS alone -- a round costs a barrier and a few LDS round
// trips instead of loads, atomics and a cache invalidate in global memory, which is what a weak chain that winds
// through the tiles of a region pays per tile -- and the promoted pixels are written back at the end.  The ring is what
// the neighbouring regions held when the launch started; tiles across the border are marked for the next launch as in
// hyst_region.
constexpr int HL_T = 8;                      // tiles per region side
constexpr int HL_ROWS = HL_T * TH;           // 256 rows
constexpr int HL_WORDS = HL_T * (TW / 64);   // 16 words per row
template <int NT>
__global__ __launch_bounds__(NT) void k_hyst_tiles_lds(Plane dst, int w, int h, int tiles_x, int tiles_y, int regions_x,
                                                       int regions_y, CannyWs W, int seed_pending, int round_no)
{
    if (round_no > 0 && *(volatile const u32*)W.again < (u32)round_no) return;
    __shared__ u64 sS[HL_ROWS + 2][HL_WORDS + 2];  // row -1 .. 256, word -1 .. 16 (index + 1)
    __shared__ u64 sW[HL_ROWS][HL_WORDS];
    __shared__ u16 s_list[2][HL_T * HL_T];
    __shared__ u32 s_inq[2], s_flag[2], s_cnt[2], s_cursor;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;
    int blk = blockIdx.x;
    const int rx = blk % regions_x;
    blk /= regions_x;
    const int ry = blk % regions_y, b = blk / regions_y;
    const int tx_base = rx * HL_T, ty_base = ry * HL_T;
    const int nx_t = min(HL_T, tiles_x - tx_base), ny_t = min(HL_T, tiles_y - ty_base);
    const int ntile = nx_t * ny_t;
    const int X0 = tx_base * TW, Y0 = ty_base * TH;  // first pixel of the region

    if (tid < 2) {
        s_inq[tid] = 0;
        s_flag[tid] = 0;
        s_cnt[tid] = 0;
    }
    if (tid == 0) s_cursor = 0;
    __syncthreads();
    // worklist of the first round, and is there anything to do at all
    for (int t = tid; t < ntile; t += NT) {
        const int lx = t % nx_t, ly = t / nx_t;
        const size_t gt = ((size_t)b * tiles_y + ty_base + ly) * tiles_x + tx_base + lx;
        if (W.has_weak[gt]) {
            atomicOr(&s_flag[t >> 5], 1u << (t & 31));
            if (!seed_pending || W.pending_in[gt]) {
                const u32 pos = atomicAdd(&s_cnt[0], 1u);
                s_list[0][pos] = (u16)t;
            }
        }
    }
    __syncthreads();
    if (s_cnt[0] == 0) return;  // (uniform)

    // ---- build the bit planes -------------------------------------------------------------------------
    for (int i = tid; i < (HL_ROWS + 2) * (HL_WORDS + 2); i += NT) (&sS[0][0])[i] = 0;
    for (int i = tid; i < HL_ROWS * HL_WORDS; i += NT) (&sW[0][0])[i] = 0;
    __syncthreads();
    const int r = lane & 31, hf = lane >> 5;  // lane = (64-pixel half, row) of a tile
    for (int t = wave; t < ntile; t += NW) {
        const int lx = t % nx_t, ly = t / nx_t;
        const int iy = Y0 + ly * TH + r, xh = X0 + lx * TW + 64 * hf;
        if (iy < h && w - xh > 0) {
            const TileBits me = bits_of_raw64(load_raw64(dst.row(b, iy) + xh, w - xh), w - xh);
            sS[ly * TH + r + 1][lx * 2 + hf + 1] = me.S;
            sW[ly * TH + r][lx * 2 + hf] = me.W;
        }
    }
    // the ring: the row above and the row below the region (a word per lane), the columns left and right of it
    {
        const int nwords = nx_t * 2, nrows = ny_t * TH;
        for (int i = tid; i < 2 * nwords; i += NT) {
            const int below = i >= nwords, wd = below ? i - nwords : i;
            const int iy = below ? Y0 + nrows : Y0 - 1, xh = X0 + wd * 64;
            if (iy >= 0 && iy < h && w - xh > 0)
                sS[below ? nrows + 1 : 0][wd + 1] = bits_of_raw64(load_raw64(dst.row(b, iy) + xh, w - xh), w - xh).S;
        }
        for (int i = tid; i < 2 * (nrows + 2); i += NT) {
            const int right = i >= nrows + 2, rr = (right ? i - (nrows + 2) : i) - 1;  // row -1 .. nrows
            const int iy = Y0 + rr, ix = right ? X0 + nwords * 64 : X0 - 1;
            if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
                const u64 bit = (u64)(dst.row(b, iy)[ix] >> 7);
                sS[rr + 1][right ? nwords + 1 : 0] = right ? bit : bit << 63;
            }
        }
    }
    __syncthreads();

    // ---- rounds on LDS ----------------------------------------------------------------------------------
    int cur = 0;
    for (int round = 0; round < (1 << 24); round++) {
        const u32 n = s_cnt[cur];
        if (n == 0) break;
        const int nxt = cur ^ 1;
        for (;;) {
            u32 i = 0;
            if (lane == 0) i = atomicAdd(&s_cursor, 1u);
            i = (u32)__builtin_amdgcn_readfirstlane((int)i);
            if (i >= n) break;
            const int t = (int)s_list[cur][i];
            const int lx = t % nx_t, ly = t / nx_t;
            const int row = ly * TH + r + 1, word = lx * 2 + hf + 1;  // (indices into sS)
            const u64 S0 = sS[row][word], Wm = sW[row - 1][word - 1];
            // the edge bit just outside this lane's outer end of the row
            const u32 side = hf ? (u32)(sS[row][word + 1] & 1ull) : (u32)(sS[row][word - 1] >> 63);
            u64 ringS = 0;
            u32 ring_side = 0;
            if (r == 0 || r == TH - 1) {
                const int rr = r == 0 ? row - 1 : row + 1;
                ringS = sS[rr][word];
                ring_side = hf ? (u32)(sS[rr][word + 1] & 1ull) : (u32)(sS[rr][word - 1] >> 63);
            }
            u64 ringH;
            {
                const u32 pr = (u32)__shfl_xor((int)(hf ? (u32)(ringS & 1ull) : (u32)(ringS >> 63)), 32);
                const u64 lin = hf ? (u64)pr : (u64)ring_side;
                const u64 rin = hf ? (u64)ring_side : (u64)pr;
                ringH = ringS | (ringS << 1) | (ringS >> 1) | lin | (rin << 63);
            }
            u64 S = S0;
            if (__any(Wm != 0)) {
                for (;;) {
                    const u64 S_in = S;
                    {
                        const u64 M = Wm | S;
                        S |= M & ((M + S) ^ M ^ S);
                        const u64 Mr = __brevll(M), Sr = __brevll(S);
                        S |= __brevll(Mr & ((Mr + Sr) ^ Mr ^ Sr));
                    }
                    const u32 pS = (u32)__shfl_xor((int)(hf ? (u32)(S & 1ull) : (u32)(S >> 63)), 32);
                    const u64 lin = hf ? (u64)pS : (u64)side;
                    const u64 rin = hf ? (u64)side : (u64)pS;
                    const u64 H = S | (S << 1) | (S >> 1) | lin | (rin << 63);
                    const u32 Hl = (u32)H, Hh = (u32)(H >> 32);
                    u64 up = (u64)(u32)__builtin_amdgcn_update_dpp(0, (int)Hl, 0x138, 0xf, 0xf, false) |
                             ((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)Hh, 0x138, 0xf, 0xf, false) << 32);
                    u64 dn = (u64)(u32)__builtin_amdgcn_update_dpp(0, (int)Hl, 0x130, 0xf, 0xf, false) |
                             ((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)Hh, 0x130, 0xf, 0xf, false) << 32);
                    if (r == 0) up = ringH;
                    if (r == TH - 1) dn = ringH;
                    const u64 Sn = S | (Wm & (H | up | dn));
                    const bool ch = Sn != S_in;
                    S = Sn;
                    if (!__any(ch)) break;
                }
            }
            const u64 prom = S & ~S0;
            if (prom) sS[row][word] = S;
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
                        W.pending[((size_t)b * tiles_y + gty) * tiles_x + gtx] = 1;
                        atomicMax(W.again, (u32)round_no + 1u);
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            s_cnt[cur] = 0;
            s_cursor = 0;
        }
        if (tid < 2) s_inq[tid] = 0;
        __syncthreads();
        cur = nxt;
    }

    // ---- promoted pixels back to the bytes --------------------------------------------------------------
    for (int t = wave; t < ntile; t += NW) {
        if (!((s_flag[t >> 5] >> (t & 31)) & 1u)) continue;  // (uniform over the wave)
        const int lx = t % nx_t, ly = t / nx_t;
        const int iy = Y0 + ly * TH + r, xh = X0 + lx * TW + 64 * hf;
        const u64 prom = sS[ly * TH + r + 1][lx * 2 + hf + 1] & sW[ly * TH + r][lx * 2 + hf];
        if (prom) {
            u8* row = dst.row(b, iy) + xh;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const u32 nib = (u32)(prom >> (4 * q)) & 0xfu;
                if (nib) atomicOr((u32*)(row + 4 * q), ((nib * 0x00204081u) & 0x01010101u) * 0xffu);
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_hyst_tiles(Plane dst, int w, int h, int tiles_x, int tiles_y,
                                                           int rt_x, int rt_y, int regions_x, int regions_y,
                                                           CannyWs W, PointSink sink, int seed_pending, int round_no)
{
    // Launch `round_no` of a sequence over the regions of huge images: nothing is left to do when the
    // launch before it did not ask for another one (it would have raised *again to round_no).
    if (round_no > 0 && *(volatile const u32*)W.again < (u32)round_no) return;
    int blk = blockIdx.x;
    const int rx = blk % regions_x;
    blk /= regions_x;
    const int ry = blk % regions_y, b = blk / regions_y;
    hyst_region<NT>(dst, w, h, tiles_x, tiles_y, rt_x, rt_y, rx, ry, b, W, sink, seed_pending, round_no, nullptr);
}

// What the region launches of a huge image left undone after their fixed number of rounds, to the fixed point and
// without the host: one workgroup per image sweeps over the image's regions, works on those that hold marked
// tiles (marks of the previous sweep or of the last region launch), collects the marks of this sweep in the other
// set, and stops after a sweep that marked nothing.  Returns at once when the last region launch asked for nothing.
template <int NT>
__global__ __launch_bounds__(NT) void k_hyst_drain(Plane dst, int w, int h, int tiles_x, int tiles_y, int rt_x, int rt_y,
                                                   int regions_x, int regions_y, CannyWs W, PointSink sink, int first_round)
{
    if (*(volatile const u32*)W.again < (u32)first_round) return;
    __shared__ u32 s_any, s_has;
    const int b = blockIdx.x, tid = threadIdx.x;
    u8 *in = (u8*)W.pending_in, *out = W.pending;
    const size_t t0 = (size_t)b * tiles_y * tiles_x;
    const int nt = tiles_x * tiles_y;
    for (int sweep = 0; sweep < (1 << 24); sweep++) {
        for (int i = tid; i < nt; i += NT) out[t0 + i] = 0;
        if (tid == 0) s_any = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int ry = 0; ry < regions_y; ry++)
            for (int rx = 0; rx < regions_x; rx++) {
                const int tx_base = rx * rt_x, ty_base = ry * rt_y;
                const int nx_t = min(rt_x, tiles_x - tx_base), ny_t = min(rt_y, tiles_y - ty_base);
                if (tid == 0) s_has = 0;
                __syncthreads();
                for (int t = tid; t < nx_t * ny_t; t += NT) {
                    const size_t gt = t0 + (size_t)(ty_base + t / nx_t) * tiles_x + tx_base + t % nx_t;
                    if (in[gt] && W.has_weak[gt]) s_has = 1u;
                }
                __syncthreads();
                if (s_has) {  // (uniform over the workgroup)
                    CannyWs Wr = W;
                    Wr.pending = out;
                    Wr.pending_in = in;
                    hyst_region<NT>(dst, w, h, tiles_x, tiles_y, rt_x, rt_y, rx, ry, b, Wr, sink, 1, -1, &s_any);
                }
                __syncthreads();
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (!s_any) break;  // (uniform: read after the barrier)
        __syncthreads();
        u8* const tmp = in;
        in = out;
        out = tmp;
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

// The same for launches of many tiles: a workgroup reads 256 tile flags at once, collects the flagged tiles in LDS
// and then clears them four at a time (all four loads in flight) -- the loop above waits for one flag and then for
// one tile per iteration, 540 times per workgroup at 1024 x 4K.
__global__ __launch_bounds__(256) void k_hyst_final_batched(Plane dst, int w, int h, int tiles_x, int tiles_y, u32 ntiles,
                                                            const u8* __restrict__ has_weak)
{
    __shared__ u32 s_list[256];
    __shared__ u32 s_n;
    const int lr = threadIdx.x >> 3, lc0 = (threadIdx.x & 7) * 16;
    for (u32 base = blockIdx.x * 256u; base < ntiles; base += gridDim.x * 256u) {  // (uniform over the workgroup)
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const u32 mine = base + threadIdx.x;
        if (mine < ntiles && has_weak[mine]) s_list[atomicAdd(&s_n, 1u)] = mine;
        __syncthreads();
        const u32 n = s_n;
        for (u32 i = 0; i < n; i += 4) {
            uint4* p[4];
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                p[k] = nullptr;
                if (i + k < n) {
                    const u32 tile = s_list[i + k];
                    const int tx = tile % tiles_x;
                    const u32 t2 = tile / tiles_x;
                    const int ty = t2 % tiles_y, b = t2 / tiles_y;
                    const int iy = ty * TH + lr, ix0 = tx * TW + lc0;
                    if (iy < h && ix0 < w) {
                        p[k] = (uint4*)(dst.row(b, iy) + ix0);
                        v[k] = *p[k];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!p[k]) continue;
                u32 d[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                bool any = false;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const u32 keep = ((d[q] >> 7) & 0x01010101u) * 0xffu;  // keep a byte only if its top bit is set
                    any |= (d[q] & ~keep) != 0;
                    d[q] &= keep;
                }
                if (any) *p[k] = make_uint4(d[0], d[1], d[2], d[3]);
            }
        }
        __syncthreads();  // s_list and s_n are reused
    }
}

// bytes that are still WEAK -> 0 in the flagged tiles
static void launch_hyst_final(Plane dst, int w, int h, int tiles_x, int tiles_y, size_t ntiles, const u8* has_weak, hipStream_t st)
{
    // (the option that forces the large-launch form of the strip kernel forces this one too: tests on small images)
    if (ntiles >= 256u * 1024u || option_canny_bands() == ST_ROWS) {
        const unsigned grid = (unsigned)((ntiles + 255) / 256 < 4096 ? (ntiles + 255) / 256 : 4096);
        hipLaunchKernelGGL(k_hyst_final_batched, dim3(grid), dim3(256), 0, st, dst, w, h, tiles_x, tiles_y, (u32)ntiles, has_weak);
    } else {
        const unsigned grid = (unsigned)(ntiles < 4096 ? ntiles : 4096);
        hipLaunchKernelGGL(k_hyst_final, dim3(grid), dim3(256), 0, st, dst, w, h, tiles_x, tiles_y, (u32)ntiles, has_weak);
    }
}

}  // namespace

// ---- launch of the wide kernel -------------------------------------------------------------------------
// np = 16-bit pixel pairs per lane (4 or 8).  The kernel keeps 32-bit offsets within an image and whole pixel
// groups of a lane inside a row.
bool canny_wide_fits(int w, int h, size_t src_pitch, size_t dst_pitch, size_t gray_pitch, size_t bin_pitch, int np)
{
    if (np != 4 && np != 8) return false;
    if (w < 2 * np || w % (2 * np) != 0 || h < 1 || w > 65535 || h > 65535) return false;  // whole pixel groups of a lane
    const unsigned long long lim = 0xffffffffull - 64;
    return (unsigned long long)h * src_pitch < lim && (unsigned long long)h * dst_pitch < lim &&
           (unsigned long long)h * gray_pitch < lim && (unsigned long long)h * bin_pitch < lim;
}

constexpr int WIDE_ROWS = 125, WIDE_ROWS_SHORT = 14;

// Pixel pairs per lane the wide kernel would use for this launch: 0 = not used.  Option canny_wide: 0 = never,
// 4 / 8 = always (where it fits), -1 = for launches that fill the chip with bands of WIDE_ROWS rows.
int canny_wide_np(int w, int h, int batch)
{
    const int opt = option_canny_wide();
    if (opt == 0) return 0;
    if (opt == 4 || opt == 8) return opt;
    const int np = 8;
    const long long waves = (long long)cdiv(w, 128 * np) * cdiv(h, WIDE_ROWS) * batch;
    return waves >= 4096 ? np : 0;
}

int launch_canny_wide(bool bgr_in, int np, Plane src, Plane gray, Plane bin, u32* bits, size_t bpw, size_t bsw, Plane dst, int w,
                      int h, int batch, int ithresh, int maxval, int inv, int low, int high, void* canny_ws, ClearList clr,
                      hipStream_t st)
{
    const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
    IPA_REQUIRE(canny_wide_fits(w, h, src.pitch, dst.pitch, bgr_in ? gray.pitch : 0, bgr_in && bin.p ? bin.pitch : 0, np),
                IPA_ERR_ARG, "Canny (wide kernel): image shape not supported");
    for (int b0 = 0; b0 < batch; b0 += 32768) {
        const int nb = batch - b0 < 32768 ? batch - b0 : 32768;
        WideArgs A;
        A.src = src; A.gray = gray; A.bin = bin; A.dst = dst;
        A.src.p += (size_t)b0 * src.stride;
        A.dst.p += (size_t)b0 * dst.stride;
        if (A.gray.p) A.gray.p += (size_t)b0 * gray.stride;
        if (A.bin.p) A.bin.p += (size_t)b0 * bin.stride;
        A.bits = bits ? bits + (size_t)b0 * bsw : nullptr;
        A.bpw = bpw; A.bsw = bsw;
        A.w = w; A.h = h;
        A.strips_x = cdiv(w, 128 * np);
        const bool short_bands = (long long)A.strips_x * cdiv(h, WIDE_ROWS) * nb < 2048;
        const int rows = short_bands ? WIDE_ROWS_SHORT : WIDE_ROWS;
        A.bands = cdiv(h, rows);
        const long long nwaves = (long long)A.strips_x * A.bands * nb;
        IPA_REQUIRE(nwaves < 0x7fffffffLL, IPA_ERR_ARG, "Canny (wide kernel): too many strips");
        A.nwaves = (u32)nwaves;
        A.low = low; A.high = high; A.tiles_x = tiles_x; A.tiles_y = tiles_y;
        A.has_weak = (u8*)canny_ws + (size_t)b0 * tiles_x * tiles_y;
        // "above" <=> gray > t <=> gray + (255 - t) >= 256; t < 0: every pixel is above
        u32 va = inv ? 0u : (u32)maxval, vb = inv ? (u32)maxval : 0u;
        int k = 255 - (ithresh > 255 ? 255 : ithresh);
        if (ithresh < 0) {
            const u32 tmp = va;
            va = vb;
            vb = tmp;
            k = 0;
        }
        A.k7 = (u32)(k & 0x7f) * 0x01010101u;
        A.kb = k >= 128 ? 0x80808080u : 0u;
        A.a4 = va * 0x01010101u; A.b4 = vb * 0x01010101u;
        A.nz_a = va ? 0xffffu : 0u; A.nz_b = vb ? 0xffffu : 0u;
        A.clr = clr;
        const dim3 grid((unsigned)((nwaves + 3) / 4));
#define IPA_WIDE(BGRV, NPV, ROWSV) hipLaunchKernelGGL((k_canny_wide<BGRV, NPV, ROWSV>), grid, dim3(256), 0, st, A)
        if (bgr_in) {
            if (np == 8) { if (short_bands) IPA_WIDE(true, 8, WIDE_ROWS_SHORT); else IPA_WIDE(true, 8, WIDE_ROWS); }
            else { if (short_bands) IPA_WIDE(true, 4, WIDE_ROWS_SHORT); else IPA_WIDE(true, 4, WIDE_ROWS); }
        } else {
            if (np == 8) { if (short_bands) IPA_WIDE(false, 8, WIDE_ROWS_SHORT); else IPA_WIDE(false, 8, WIDE_ROWS); }
            else { if (short_bands) IPA_WIDE(false, 4, WIDE_ROWS_SHORT); else IPA_WIDE(false, 4, WIDE_ROWS); }
        }
#undef IPA_WIDE
        IPA_LAUNCH_CHECK();
        clr.n[0] = clr.n[1] = clr.n[2] = 0;  // (only the first launch clears)
    }
    return IPA_OK;
}

size_t canny_ws_bytes(int w, int h, int batch)
{
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    size_t ntiles = (size_t)cdiv(w, TW) * cdiv(h, TH) * batch;
    // per tile: "holds WEAK pixels" and two sets of "revisit" marks (written / read by alternate launches); one counter
    return 3 * align_up(ntiles, 256) + 512;
}

// Enqueues the whole detector; nothing here waits for the stream.  Images of up to HT_MAX_TILES tiles
// (e.g. 4K) are resolved by one workgroup each; larger images are cut into regions, worked on in a
// fixed number of launches and finished by one workgroup per image (k_hyst_drain).
// The part of the workspace run_canny wants zeroed before its first kernel (tile flags and the first set of marks).
void canny_flag_area(int w, int h, int batch, void* ws, u32** p, u32* dwords)
{
    const size_t ntiles = (size_t)cdiv(w, TW) * cdiv(h, TH) * batch;
    *p = (u32*)ws;
    *dwords = (u32)(2 * align_up(ntiles, 256) / 4);
}

int run_canny(Plane src, Plane dst, int w, int h, int batch, int low, int high, void* ws, size_t ws_bytes,
              PointSink sink, hipStream_t st, bool flags_cleared, bool nms_done)
{
    const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
    const long long ntiles_ll = (long long)tiles_x * tiles_y * batch;
    IPA_REQUIRE(ntiles_ll < 0x7fffffffLL, IPA_ERR_ARG, "Canny: too many tiles");
    IPA_REQUIRE(w <= 65535 && h <= 65535, IPA_ERR_ARG, "Canny: image larger than 65535 pixels on a side");
    IPA_REQUIRE(ws && ws_bytes >= canny_ws_bytes(w, h, batch), IPA_ERR_WORKSPACE, "Canny: workspace too small");
    const size_t ntiles = (size_t)ntiles_ll;
    CannyWs W;
    W.has_weak = (u8*)ws;
    W.pending = nullptr;
    W.pending_in = nullptr;
    u8* const pending = (u8*)ws + align_up(ntiles, 256);  // directly behind has_weak: one memset clears both
    W.again = (u32*)((u8*)ws + 3 * align_up(ntiles, 256));
    // Few images of at most HT_MAX_TILES tiles: one workgroup per image would leave the chip idle, so a
    // first launch works on small regions side by side and marks the tiles across region borders that
    // would have to be revisited; a second launch (one workgroup per image, started from the marked
    // tiles only) then runs to the fixed point.
    const int tiles_x0 = cdiv(w, TW), tiles_y0 = cdiv(h, TH);
    static int side_env = -2;  // IPA_HYST_SMALL_SIDE: region side in tiles of the first launch (0 = off; default: by batch)
    if (side_env == -2) {
        const char* e = getenv("IPA_HYST_SMALL_SIDE");
        side_env = e ? atoi(e) : -1;
        if (side_env < -1 || side_env > 64) side_env = -1;
    }
    // measured on 4K images: 8 x 8 tiles for up to 4 images, 16 x 16 up to 48, one workgroup per image beyond
    const int small_side = side_env >= 0 ? side_env : (batch <= 4 ? 8 : (batch <= 48 ? 16 : 0));
    const bool two_step = small_side > 0 && batch < 128 && (long long)tiles_x0 * tiles_y0 <= HT_MAX_TILES &&
                          (tiles_x0 > small_side || tiles_y0 > small_side);
    static int mode = -1;  // IPA_CANNY_MODE: 1 = streaming register kernel (default), 0 = LDS tile kernel
    if (mode < 0) {
        const char* e = getenv("IPA_CANNY_MODE");
        mode = e ? atoi(e) : 1;
    }
    // nms_done: the fused BGR kernel has already written the codes and the tile flags of this batch
    const int wide_np = nms_done || sink.pts || mode != 1 ? 0 : canny_wide_np(w, h, batch);
    if (nms_done) {
    } else if (wide_np && canny_wide_fits(w, h, src.pitch, dst.pitch, 0, 0, wide_np)) {
        ProfScope ps(PROF_CANNY_NMS, st);
        if (!flags_cleared) IPA_HIP(hipMemsetAsync(W.has_weak, 0, (two_step ? 2 : 1) * align_up(ntiles, 256), st));
        const ClearList none = {{nullptr, nullptr, nullptr}, {0, 0, 0}};
        const int rc = launch_canny_wide(false, wide_np, src, mkplane(nullptr, 0, 0), mkplane(nullptr, 0, 0), nullptr, 0, 0, dst, w, h,
                                         batch, 0, 0, 0, low, high, W.has_weak, none, st);
        if (rc) return rc;
    } else {
        ProfScope ps(PROF_CANNY_NMS, st);
        // the streaming kernel reads whole 32-bit words inside a row and uses 32-bit offsets within an image
        const bool stream_ok = w >= 4 && (unsigned long long)h * src.pitch < 0xffffffffull && (unsigned long long)h * dst.pitch < 0xffffffffull;
        if (mode == 1 && (stream_ok || sink.pts)) {
            IPA_REQUIRE(stream_ok, IPA_ERR_ARG, "Canny: image too small or too large for the fused edge-list path");
            // (whole multiples of 256 bytes: the runtime then clears them with one kernel instead of two)
            if (!flags_cleared) IPA_HIP(hipMemsetAsync(W.has_weak, 0, (two_step ? 2 : 1) * align_up(ntiles, 256), st));
            const int strips_x = cdiv(w, ST_COLS);
            // short bands while 64-row bands would give fewer than 4096 waves (16 per CU)
            const int bands_opt = option_canny_bands();  // run-time option canny_bands: 16 / 64 force a band height
            const bool short_bands = bands_opt == ST_ROWS_SHORT || (bands_opt != ST_ROWS && (long long)strips_x * cdiv(h, ST_ROWS) * batch < 4096);
            const int bands = cdiv(h, short_bands ? ST_ROWS_SHORT : ST_ROWS);
            const long long nwaves = (long long)strips_x * bands * batch;
            IPA_REQUIRE(nwaves < 0x7fffffffLL, IPA_ERR_ARG, "Canny: too many strips");
            static const size_t pad = lds_pad_kib("IPA_PAD_CANNY");
            const dim3 grid((unsigned)((nwaves + 3) / 4));
            const bool queue = option_canny_queue() != 0;  // run-time option canny_queue
#define IPA_CANNY_STREAM(EMITV, ROWSV, QUEUEV)                                                                      \
    hipLaunchKernelGGL((k_canny_stream<EMITV, ROWSV, QUEUEV>), grid, dim3(256), pad, st, src, dst, w, h, strips_x, bands, \
                       nwaves, low, high, tiles_x, tiles_y, W, sink)
            if (sink.pts && short_bands) IPA_CANNY_STREAM(true, ST_ROWS_SHORT, false);
            else if (sink.pts) IPA_CANNY_STREAM(true, ST_ROWS, false);
            else if (short_bands) IPA_CANNY_STREAM(false, ST_ROWS_SHORT, false);
            else if (queue) IPA_CANNY_STREAM(false, ST_ROWS, true);
            else IPA_CANNY_STREAM(false, ST_ROWS, false);
#undef IPA_CANNY_STREAM
        } else {
            IPA_REQUIRE(!sink.pts, IPA_ERR_ARG, "Canny: the LDS tile kernel cannot emit the edge-pixel list");
            hipLaunchKernelGGL(k_canny_nms, dim3((unsigned)ntiles), dim3(256), 0, st, src, dst, w, h, tiles_x, tiles_y, low,
                               high, W);
        }
        IPA_LAUNCH_CHECK();
    }
    ProfScope ps(PROF_HYSTERESIS, st);
    if (two_step) {
        if (mode != 1 && !flags_cleared) IPA_HIP(hipMemsetAsync(pending, 0, ntiles, st));  // (the tile kernel writes has_weak itself)
        const int rx = cdiv(tiles_x, small_side), ry = cdiv(tiles_y, small_side);
        CannyWs W1 = W;
        W1.pending = pending;
        W1.pending_in = pending;
        hipLaunchKernelGGL(k_hyst_tiles<HT_THREADS>, dim3((unsigned)(rx * ry * batch)), dim3(HT_THREADS), 0, st, dst, w, h, tiles_x,
                           tiles_y, small_side, small_side, rx, ry, W1, sink, 0, -1);
        IPA_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_hyst_tiles<2 * HT_THREADS>, dim3((unsigned)batch), dim3(2 * HT_THREADS), 0, st, dst, w, h, tiles_x,
                           tiles_y, tiles_x, tiles_y, 1, 1, W1, sink, 1, -1);
        IPA_LAUNCH_CHECK();
        launch_hyst_final(dst, w, h, tiles_x, tiles_y, ntiles, (const u8*)W.has_weak, st);
        IPA_LAUNCH_CHECK();
        return IPA_OK;
    }
    // regions of at most HT_MAX_TILES tiles (64 x 64 tiles = 8192 x 2048 pixels)
    int rt_x = tiles_x, rt_y = tiles_y;
    if ((long long)rt_x * rt_y > HT_MAX_TILES) {
        // one workgroup per region: small regions keep the whole chip busy on a single large image,
        // at the price of more rounds for chains that cross many of them
        static int side = -1;  // IPA_HYST_REGION: region side in tiles (square of tiles; 1..64)
        if (side < 0) {
            const char* e = getenv("IPA_HYST_REGION");
            side = e ? atoi(e) : 8;  // (16384 x 16384 noisy page: 0.73 ms with 8, 0.80 with 16, 1.05 with 32, 2.4 with 4)
            if (side < 1 || side > 64) side = 8;
        }
        rt_x = tiles_x < side ? tiles_x : side;
        rt_y = tiles_y < side ? tiles_y : side;
    }
    const int regions_x = cdiv(tiles_x, rt_x), regions_y = cdiv(tiles_y, rt_y);
    const long long blocks = (long long)regions_x * regions_y * batch;
    IPA_REQUIRE(blocks < 0x7fffffffLL, IPA_ERR_ARG, "Canny: too many regions");
    const bool multi = regions_x * regions_y > 1;
    // Several regions per image: launch r starts from the tiles launch r-1 marked across region borders
    // (launch 0 from every tile with WEAK pixels) and marks the tiles it changes across borders in turn;
    // the two sets of marks alternate.  A fixed number of launches is queued -- one that has nothing to do
    // returns at once -- and a last kernel (one workgroup per image) takes whatever they left to the fixed
    // point on the device: the host never waits for the stream and nothing is copied back.
    u8* const pend[2] = {pending, pending + align_up(ntiles, 256)};
    if (!multi) {
        if (blocks < 1024)
            hipLaunchKernelGGL(k_hyst_tiles<2 * HT_THREADS>, dim3((unsigned)blocks), dim3(2 * HT_THREADS), 0, st, dst, w, h, tiles_x,
                               tiles_y, rt_x, rt_y, regions_x, regions_y, W, sink, 0, -1);
        else
            hipLaunchKernelGGL(k_hyst_tiles<HT_THREADS>, dim3((unsigned)blocks), dim3(HT_THREADS), 0, st, dst, w, h, tiles_x,
                               tiles_y, rt_x, rt_y, regions_x, regions_y, W, sink, 0, -1);
        IPA_LAUNCH_CHECK();
    } else {
        IPA_HIP(hipMemsetAsync(W.again, 0, 16, st));
        int rounds = option_hyst_rounds();  // region launches before the drain kernel (run-time option hyst_rounds)
        rounds = rounds < 1 ? 1 : (rounds > 64 ? 64 : rounds);
        // regions of 8 x 8 tiles keep their state in LDS (not with the fused edge-pixel list; IPA_HYST_LDS=0: A/B)
        static int lds_env = -1;
        if (lds_env < 0) {
            const char* e = getenv("IPA_HYST_LDS");
            lds_env = e ? atoi(e) : 1;
        }
        const bool lds_regions = lds_env != 0 && !sink.pts && rt_x <= HL_T && rt_y <= HL_T && (rt_x == HL_T || regions_x == 1) &&
                                 (rt_y == HL_T || regions_y == 1);
        for (int r = 0; r < rounds; r++) {
            CannyWs Wr = W;
            Wr.pending = pend[r & 1];
            Wr.pending_in = pend[(r + 1) & 1];
            IPA_HIP(hipMemsetAsync(pend[r & 1], 0, align_up(ntiles, 256), st));
            // few workgroups (small batches, single images): twice the waves per workgroup
            if (lds_regions)
                hipLaunchKernelGGL(k_hyst_tiles_lds<1024>, dim3((unsigned)blocks), dim3(1024), 0, st, dst, w, h, tiles_x, tiles_y,
                                   regions_x, regions_y, Wr, r > 0 ? 1 : 0, r);
            else if (blocks < 1024)
                hipLaunchKernelGGL(k_hyst_tiles<2 * HT_THREADS>, dim3((unsigned)blocks), dim3(2 * HT_THREADS), 0, st, dst, w, h, tiles_x,
                                   tiles_y, rt_x, rt_y, regions_x, regions_y, Wr, sink, r > 0 ? 1 : 0, r);
            else
                hipLaunchKernelGGL(k_hyst_tiles<HT_THREADS>, dim3((unsigned)blocks), dim3(HT_THREADS), 0, st, dst, w, h, tiles_x,
                                   tiles_y, rt_x, rt_y, regions_x, regions_y, Wr, sink, r > 0 ? 1 : 0, r);
            IPA_LAUNCH_CHECK();
        }
        CannyWs Wd = W;
        Wd.pending = pend[rounds & 1];           // (cleared by the kernel itself)
        Wd.pending_in = pend[(rounds + 1) & 1];  // the marks of the last region launch
        hipLaunchKernelGGL(k_hyst_drain<2 * HT_THREADS>, dim3((unsigned)batch), dim3(2 * HT_THREADS), 0, st, dst, w, h, tiles_x,
                           tiles_y, rt_x, rt_y, regions_x, regions_y, Wd, sink, rounds);
        IPA_LAUNCH_CHECK();
    }
    launch_hyst_final(dst, w, h, tiles_x, tiles_y, ntiles, (const u8*)W.has_weak, st);
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
    PointSink none = {nullptr, 0, 0, nullptr};
    return run_canny(s, d, width, height, batch, low, high, ws, ws_bytes, none, (hipStream_t)stream);
}

// ---- Canny for callers that cannot allocate -------------------------------------------------
// Same detector, same kernels; the per-tile scratch lives in a device array of the library
// instead of a caller's workspace.  Every GPU holds its own copy of the array at its own
// address, looked up once per device.
constexpr size_t CANNY_STATIC_BYTES = IPA_CANNY_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[CANNY_STATIC_BYTES];

namespace {
std::mutex g_scratch_mu;
void* g_scratch_ptr[64] = {};

// Device address of `scratch` on the current device (nullptr + error text on failure).
void* static_scratch_ptr()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        ipa_set_error("Canny (static scratch): no current device");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            ipa_set_error("hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = p;
    }
    return g_scratch_ptr[dev];
}
}  // namespace

IPA_EXPORT size_t ipa_canny_static_max_bytes(void) { return CANNY_STATIC_BYTES; }

IPA_EXPORT int ipa_canny_static_u8(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst, size_t dst_pitch,
                                   size_t dst_stride, int width, int height, int batch, double threshold1,
                                   double threshold2, void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0, IPA_ERR_ARG, "negative image size");
    if (width == 0 || height == 0 || batch == 0) return IPA_OK;
    const size_t need = canny_ws_bytes(width, height, batch);
    IPA_REQUIRE(need <= CANNY_STATIC_BYTES, IPA_ERR_WORKSPACE,
                "Canny (static scratch): %d x %d x %d images need %zu bytes of scratch, the built-in array holds %zu "
                "(use ipa_canny_u8 with a workspace)",
                width, height, batch, need, CANNY_STATIC_BYTES);
    if (stream) {  // the array of the CURRENT device is used: the stream must belong to it
        int cur = -1;
        hipDevice_t sd = -1;
        if (hipGetDevice(&cur) == hipSuccess && hipStreamGetDevice((hipStream_t)stream, &sd) == hipSuccess && (int)sd != cur) {
            ipa_set_error("Canny (static scratch): the stream belongs to device %d but device %d is current", (int)sd, cur);
            return IPA_ERR_ARG;
        }
    }
    void* ws = static_scratch_ptr();
    if (!ws) return IPA_ERR_HIP;
    return ipa_canny_u8(src, src_pitch, src_stride, dst, dst_pitch, dst_stride, width, height, batch, threshold1, threshold2,
                        ws, CANNY_STATIC_BYTES, stream);
}
