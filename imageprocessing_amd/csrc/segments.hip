// Bounding rectangles of the external contours of a mask: the get_segments step of the
// parsers (cv2.findContours(lines, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) + cv2.boundingRect,
// umlClassdiagram.py:33-36 and its three siblings).
//
// RETR_EXTERNAL keeps the 8-connected components of the non-zero pixels that are not enclosed
// by another component; cv2 lists them by their first pixel in raster order, last first.
// Everything is done on row SEGMENTS (maximal runs of equal pixels of a row, both colours) of
// the 1-bit plane of the mask, so the cost follows the number of runs, not of pixels:
//
//   k_seg_count   transitions per row -> segments per row
//   k_seg_scan    per image: row offsets into the segment arrays, capacity check
//   k_seg_fill    segment start columns, parent = self, boxes = own extent
//   k_seg_union   lock-free union-find between the segments of adjacent rows: foreground with
//                 8-connectivity, background with 4-connectivity; the root of a set is its
//                 smallest index = its first segment in raster order (cv2's contour start)
//   k_seg_boxes   roots collect the extent of their foreground set; background sets that touch
//                 the image border are marked as the outside
//   k_seg_extern  a foreground set is external when one of its segments touches the image
//                 border or is 4-adjacent to an outside background segment
//   k_seg_emit    external roots in descending index order -> (x, y, w, h)
#include "common.hpp"

namespace {

constexpr u32 SEG_OUTSIDE = 1u;   // background set reaches the image border
constexpr u32 SEG_EXTERNAL = 2u;  // foreground set is not enclosed

struct SegArrays {
    u32* rowcnt;   // [batch][h]      segments per row
    u32* rowoff;   // [batch][h + 1]  exclusive prefix of rowcnt
    u32* xs;       // [batch][cap]    first column of the segment
    u32* yy;       // [batch][cap]    row of the segment
    u32* parent;   // [batch][cap]
    u32* x0;       // [batch][cap]    extent of the set (valid at roots)
    u32* x1;
    u32* y1;
    u32* flags;    // [batch][cap]
    int32_t* total;  // [batch]       segments of the image, -1: more than cap
    u32 cap;
};

// Transition bits of word j of a row: bit i set <=> pixel 32 j + i differs from the pixel before it
// (never for pixel 0 of the row, never beyond the width).
__device__ __forceinline__ u32 transitions(const u32* row, int j, int words, int w)
{
    const u32 cur = row[j];
    const u32 prevmsb = j > 0 ? row[j - 1] >> 31 : (cur & 1u);  // pixel 0 compares with itself
    u32 t = cur ^ ((cur << 1) | prevmsb);
    const int rem = w - 32 * j;
    if (rem < 32) t &= (1u << rem) - 1u;
    (void)words;
    return t;
}

// one wave per row
__global__ __launch_bounds__(256) void k_seg_count(const u32* __restrict__ bits, size_t bpw, size_t bsw, int w, int h,
                                                   SegArrays A)
{
    const int b = blockIdx.y;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    const int lane = threadIdx.x & 63;
    const int words = (w + 31) >> 5;
    const u32* row = bits + (size_t)b * bsw + (size_t)y * bpw;
    u32 n = 0;
    for (int j = lane; j < words; j += 64) n += __popc(transitions(row, j, words, w));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += (u32)__shfl_xor((int)n, o);
    if (lane == 0) A.rowcnt[(size_t)b * h + y] = n + 1;
}

// one workgroup per image: exclusive scan of the row counts
__global__ __launch_bounds__(1024) void k_seg_scan(int h, SegArrays A)
{
    __shared__ u32 s_wave[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const u32* cnt = A.rowcnt + (size_t)b * h;
    u32* off = A.rowoff + (size_t)b * (h + 1);
    u32 base = 0;
    for (int y0 = 0; y0 < h; y0 += 1024) {
        const int y = y0 + tid;
        const u32 c = y < h ? cnt[y] : 0u;
        u32 total;
        const u32 incl = block_scan_1024(c, s_wave, total);
        if (y < h) off[y] = base + incl - c;
        base += total;
    }
    if (tid == 0) {
        off[h] = base;
        A.total[b] = base > A.cap ? -1 : (int32_t)base;
    }
}

// one wave per row: start column of every segment, parent = self, own extent
__global__ __launch_bounds__(256) void k_seg_fill(const u32* __restrict__ bits, size_t bpw, size_t bsw, int w, int h,
                                                  SegArrays A)
{
    const int b = blockIdx.y;
    if (A.total[b] < 0) return;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    const int lane = threadIdx.x & 63;
    const int words = (w + 31) >> 5;
    const u32* row = bits + (size_t)b * bsw + (size_t)y * bpw;
    const size_t base = (size_t)b * A.cap;
    u32 k = A.rowoff[(size_t)b * (h + 1) + y];  // index of the row's next segment (wave-uniform)
    auto put = [&](u32 idx, u32 xstart) {
        A.xs[base + idx] = xstart;
        A.yy[base + idx] = (u32)y;
        A.parent[base + idx] = idx;
        A.x0[base + idx] = 0xffffffffu;  // extents are collected at the roots by k_seg_boxes
        A.x1[base + idx] = 0;
        A.y1[base + idx] = 0;
        A.flags[base + idx] = 0;
    };
    if (lane == 0) put(k, 0);
    k += 1;
    for (int j0 = 0; j0 < words; j0 += 64) {
        const int j = j0 + lane;
        u32 t = j < words ? transitions(row, j, words, w) : 0u;
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
            put(pos++, (u32)(32 * j + i));
        }
        k += (u32)__shfl((int)scan, 63);
    }
}

__device__ __forceinline__ u32 seg_find(const u32* parent, u32 i)
{
    // agent-scope loads: other workgroups move these links with atomics while we read
    u32 p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != i) {
        i = p;
        p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return i;
}

__device__ __forceinline__ void seg_union(u32* parent, u32 a, u32 c)
{
    const u32 a_in = a, c_in = c;
    for (;;) {
        a = seg_find(parent, a);
        c = seg_find(parent, c);
        if (a == c) {
            // shorten the chains for later finds: a link may always move to a smaller member of its set
            if (a_in != a) atomicMin(&parent[a_in], a);
            if (c_in != a) atomicMin(&parent[c_in], a);
            return;
        }
        if (a < c) {
            const u32 t = a;
            a = c;
            c = t;
        }
        // a > c: hang the larger root under the smaller one
        const u32 old = atomicMin(&parent[a], c);
        if (old == a) return;
        a = old;  // somebody else moved a meanwhile: unite its new parent with c
    }
}

// value (0 / 1) of segment k of row y: segments alternate, the first has the value of pixel 0
__device__ __forceinline__ u32 seg_value(const u32* bits, size_t bpw, size_t bsw, int b, int y, u32 k_in_row)
{
    return ((bits[(size_t)b * bsw + (size_t)y * bpw] & 1u) ^ k_in_row) & 1u;
}

// index (within row y) of the segment that contains column x
__device__ __forceinline__ u32 seg_locate(const u32* xs, u32 first, u32 n, u32 x)
{
    u32 lo = 0, hi = n;  // last segment with xs <= x
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (xs[first + mid] <= x) lo = mid;
        else hi = mid;
    }
    return lo;
}

// one wave per row y >= 1: unite its segments with the touching segments of equal value of row y-1
__global__ __launch_bounds__(256) void k_seg_union(const u32* __restrict__ bits, size_t bpw, size_t bsw, int w, int h,
                                                   SegArrays A)
{
    const int b = blockIdx.y;
    if (A.total[b] < 0) return;
    const int y = 1 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    const int lane = threadIdx.x & 63;
    const u32* off = A.rowoff + (size_t)b * (h + 1);
    const u32 f1 = off[y], n1 = off[y + 1] - f1, f0 = off[y - 1], n0 = f1 - f0;
    const u32* xs = A.xs + (size_t)b * A.cap;
    u32* parent = A.parent + (size_t)b * A.cap;
    const u32 v0first = seg_value(bits, bpw, bsw, b, y - 1, 0), v1first = seg_value(bits, bpw, bsw, b, y, 0);
    for (u32 k = lane; k < n1; k += 64) {
        const u32 val = (v1first ^ k) & 1u;
        const u32 sx = xs[f1 + k], ex = (k + 1 < n1 ? xs[f1 + k + 1] : (u32)w) - 1u;
        // columns of row y-1 that touch this segment: one more on each side for the foreground
        const u32 lo = val && sx > 0 ? sx - 1 : sx, hi = val && ex + 1 < (u32)w ? ex + 1 : ex;
        for (u32 j = seg_locate(xs, f0, n0, lo); j < n0 && xs[f0 + j] <= hi; j++)
            if (((v0first ^ j) & 1u) == val) seg_union(parent, f1 + k, f0 + j);
    }
}

// flatten; foreground roots collect extents; background sets touching the border are the outside
__global__ __launch_bounds__(256) void k_seg_boxes(const u32* __restrict__ bits, size_t bpw, size_t bsw, int w, int h,
                                                   SegArrays A)
{
    const int b = blockIdx.y;
    const int32_t tot = A.total[b];
    if (tot < 0) return;
    const size_t base = (size_t)b * A.cap;
    u32* parent = A.parent + base;
  // the grid is sized for a typical mask, not for the capacity: workgroups walk the segments
  for (u32 i = blockIdx.x * 256u + threadIdx.x; i < (u32)tot; i += gridDim.x * 256u) {
    const u32 r = seg_find(parent, i);
    if (r != i) parent[i] = r;  // roots never change once the unions are done
    const u32 y = A.yy[base + i];
    const u32 krow = i - A.rowoff[(size_t)b * (h + 1) + y];
    const u32 val = seg_value(bits, bpw, bsw, b, (int)y, krow);
    const u32 sx = A.xs[base + i];
    const u32 ex = (i + 1 < A.rowoff[(size_t)b * (h + 1) + y + 1] ? A.xs[base + i + 1] : (u32)w) - 1u;
    // The extents only ever move outwards and the flag is only ever set, so a (possibly stale)
    // plain read that already satisfies this segment makes the atomic unnecessary: large sets
    // would otherwise serialise on their root.
    if (val) {
        if (__builtin_nontemporal_load(&A.x0[base + r]) > sx) atomicMin(&A.x0[base + r], sx);
        if (__builtin_nontemporal_load(&A.x1[base + r]) < ex) atomicMax(&A.x1[base + r], ex);
        if (__builtin_nontemporal_load(&A.y1[base + r]) < y) atomicMax(&A.y1[base + r], y);
    } else if (y == 0 || y == (u32)h - 1 || sx == 0 || ex == (u32)w - 1) {
        if (!(__builtin_nontemporal_load(&A.flags[base + r]) & SEG_OUTSIDE)) atomicOr(&A.flags[base + r], SEG_OUTSIDE);
    }
  }
}

__global__ __launch_bounds__(256) void k_seg_extern(const u32* __restrict__ bits, size_t bpw, size_t bsw, int w, int h,
                                                    SegArrays A)
{
    const int b = blockIdx.y;
    const int32_t tot = A.total[b];
    if (tot < 0) return;
    const size_t base = (size_t)b * A.cap;
    const u32* off = A.rowoff + (size_t)b * (h + 1);
    const u32* xs = A.xs + base;
    const u32* parent = A.parent + base;
  for (u32 i = blockIdx.x * 256u + threadIdx.x; i < (u32)tot; i += gridDim.x * 256u) {
    const u32 y = A.yy[base + i];
    const u32 f = off[y], n = off[y + 1] - f, k = i - f;
    if (!seg_value(bits, bpw, bsw, b, (int)y, k)) continue;  // foreground segments only
    const u32 sx = xs[i], ex = (k + 1 < n ? xs[i + 1] : (u32)w) - 1u;
    bool ext = y == 0 || y == (u32)h - 1 || sx == 0 || ex == (u32)w - 1;
    auto outside = [&](u32 seg) -> bool { return (A.flags[base + parent[seg]] & SEG_OUTSIDE) != 0; };
    // left and right neighbours in the row are background
    if (!ext && k > 0) ext = outside(i - 1);
    if (!ext && k + 1 < n) ext = outside(i + 1);
    // background segments of the rows above and below that share a column with this segment
    for (int dy = -1; dy <= 1 && !ext; dy += 2) {
        const u32 yn = y + dy;
        const u32 fn = off[yn], nn = off[yn + 1] - fn;
        const u32 vfirst = seg_value(bits, bpw, bsw, b, (int)yn, 0);
        for (u32 j = seg_locate(xs, fn, nn, sx); j < nn && xs[fn + j] <= ex && !ext; j++)
            if (((vfirst ^ j) & 1u) == 0u) ext = outside(fn + j);
    }
    if (ext && !(__builtin_nontemporal_load(&A.flags[base + parent[i]]) & SEG_EXTERNAL))
        atomicOr(&A.flags[base + parent[i]], SEG_EXTERNAL);
  }
}

// one workgroup per image: external foreground roots, largest index first
__global__ __launch_bounds__(1024) void k_seg_emit(const u32* __restrict__ bits, size_t bpw, size_t bsw, int h, SegArrays A,
                                                   int32_t* __restrict__ boxes, int max_boxes, int32_t* __restrict__ counts)
{
    __shared__ u32 s_wave[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t tot = A.total[b];
    if (tot < 0) {
        if (tid == 0) counts[b] = -1;
        return;
    }
    const size_t base = (size_t)b * A.cap;
    u32 done = 0;
    for (u32 c0 = 0; c0 < (u32)tot; c0 += 1024) {
        const u32 rev = c0 + tid;  // position counted from the end
        bool is = false;
        u32 i = 0;
        if (rev < (u32)tot) {
            i = (u32)tot - 1 - rev;
            is = A.parent[base + i] == i && (A.flags[base + i] & SEG_EXTERNAL);
            if (is) {
                const u32 y = A.yy[base + i];
                is = seg_value(bits, bpw, bsw, b, (int)y, i - A.rowoff[(size_t)b * (h + 1) + y]) != 0;
            }
        }
        u32 total;
        const u32 incl = block_scan_1024(is ? 1u : 0u, s_wave, total);
        if (is) {
            const u32 pos = done + incl - 1;
            if (pos < (u32)max_boxes) {
                int32_t* o = boxes + ((size_t)b * max_boxes + pos) * 4;
                const u32 x0 = A.x0[base + i], y0 = A.yy[base + i];
                o[0] = (int32_t)x0;
                o[1] = (int32_t)y0;
                o[2] = (int32_t)(A.x1[base + i] - x0 + 1);
                o[3] = (int32_t)(A.y1[base + i] - y0 + 1);
            }
        }
        done += total;
    }
    if (tid == 0) counts[b] = (int32_t)done;
}

u32 seg_capacity(int w, int h, int max_segments)
{
    const long long full = (long long)w * h;  // a row of w pixels has at most w segments
    long long cap = max_segments > 0 ? max_segments : (long long)w * h / 64 + 4 * (long long)h + 4096;
    if (cap > full) cap = full;
    if (cap < (long long)h) cap = h;
    return (u32)(cap > 0x7fffffffLL ? 0x7fffffffLL : cap);
}

}  // namespace

int launch_pack_bits(Plane src, u32* bits, size_t bpw, size_t bsw, int w, int h, int batch, int32_t* vmax, hipStream_t st);

IPA_EXPORT size_t ipa_segments_workspace_bytes(int width, int height, int batch, int max_segments)
{
    if (width <= 0 || height <= 0 || batch <= 0) return 0;
    const size_t cap = seg_capacity(width, height, max_segments);
    Arena a(nullptr, 0);
    a.take(bits_pitch_words(width) * 4 * (size_t)height * batch);  // bit plane
    a.take(sizeof(int32_t) * (size_t)batch);                       // largest pixel (by-product of packing)
    a.take(sizeof(u32) * (size_t)height * batch);                  // rowcnt
    a.take(sizeof(u32) * ((size_t)height + 1) * batch);            // rowoff
    for (int i = 0; i < 7; i++) a.take(sizeof(u32) * cap * batch); // xs yy parent x0 x1 y1 flags
    a.take(sizeof(int32_t) * (size_t)batch);                       // total
    return a.off + 256;
}

namespace {
// the passes behind the bit plane; `bits` is read only
int segments_from_bits(const u32* bits, size_t bpw, size_t bsw, int width, int height, int batch, int max_segments,
                       int32_t* boxes, int max_boxes, int32_t* counts, Arena& ar, hipStream_t st)
{
    const u32 cap = seg_capacity(width, height, max_segments);
    SegArrays A;
    A.rowcnt = (u32*)ar.take(sizeof(u32) * (size_t)height * batch);
    A.rowoff = (u32*)ar.take(sizeof(u32) * ((size_t)height + 1) * batch);
    A.xs = (u32*)ar.take(sizeof(u32) * (size_t)cap * batch);
    A.yy = (u32*)ar.take(sizeof(u32) * (size_t)cap * batch);
    A.parent = (u32*)ar.take(sizeof(u32) * (size_t)cap * batch);
    A.x0 = (u32*)ar.take(sizeof(u32) * (size_t)cap * batch);
    A.x1 = (u32*)ar.take(sizeof(u32) * (size_t)cap * batch);
    A.y1 = (u32*)ar.take(sizeof(u32) * (size_t)cap * batch);
    A.flags = (u32*)ar.take(sizeof(u32) * (size_t)cap * batch);
    A.total = (int32_t*)ar.take(sizeof(int32_t) * (size_t)batch);
    A.cap = cap;
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        SegArrays B = A;
        B.rowcnt += (size_t)b0 * height;
        B.rowoff += (size_t)b0 * (height + 1);
        B.xs += (size_t)b0 * cap; B.yy += (size_t)b0 * cap; B.parent += (size_t)b0 * cap;
        B.x0 += (size_t)b0 * cap; B.x1 += (size_t)b0 * cap; B.y1 += (size_t)b0 * cap; B.flags += (size_t)b0 * cap;
        B.total += b0;
        const u32* bb = bits + (size_t)b0 * bsw;
        const dim3 rows((unsigned)cdiv(height, 4), (unsigned)nb), segs((unsigned)(cdiv((long long)cap, 256) < 64 ? cdiv((long long)cap, 256) : 64), (unsigned)nb);
        hipLaunchKernelGGL(k_seg_count, rows, dim3(256), 0, st, bb, bpw, bsw, width, height, B);
        hipLaunchKernelGGL(k_seg_scan, dim3((unsigned)nb), dim3(1024), 0, st, height, B);
        hipLaunchKernelGGL(k_seg_fill, rows, dim3(256), 0, st, bb, bpw, bsw, width, height, B);
        if (height > 1)
            hipLaunchKernelGGL(k_seg_union, dim3((unsigned)cdiv(height - 1, 4), (unsigned)nb), dim3(256), 0, st, bb, bpw, bsw,
                               width, height, B);
        hipLaunchKernelGGL(k_seg_boxes, segs, dim3(256), 0, st, bb, bpw, bsw, width, height, B);
        hipLaunchKernelGGL(k_seg_extern, segs, dim3(256), 0, st, bb, bpw, bsw, width, height, B);
        hipLaunchKernelGGL(k_seg_emit, dim3((unsigned)nb), dim3(1024), 0, st, bb, bpw, bsw, height, B,
                           boxes + (size_t)b0 * max_boxes * 4, max_boxes, counts + b0);
        IPA_LAUNCH_CHECK();
    }
    return IPA_OK;
}
}  // namespace

IPA_EXPORT int ipa_segments_u8(const uint8_t* mask, size_t mask_pitch, size_t mask_stride, int width, int height, int batch,
                               int max_segments, int32_t* boxes, int max_boxes, int32_t* counts, void* ws, size_t ws_bytes,
                               void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0 && max_boxes >= 0, IPA_ERR_ARG, "segments: negative size");
    if (batch == 0) return IPA_OK;
    IPA_REQUIRE(counts != nullptr && (boxes != nullptr || max_boxes == 0), IPA_ERR_ARG, "segments: NULL output");
    hipStream_t st = (hipStream_t)stream;
    if (width == 0 || height == 0) {
        IPA_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)batch, st));
        return IPA_OK;
    }
    IPA_REQUIRE(width <= 65535 && height <= 65535, IPA_ERR_ARG, "segments: image larger than 65535 pixels on a side");
    Plane s = mkplane(mask, mask_pitch, mask_stride);
    IPA_REQUIRE(plane_ok(s, (size_t)width, batch), IPA_ERR_ALIGN,
                "segments: pointer/pitch must be 16-byte aligned and pitch >= row bytes");
    IPA_REQUIRE(ws && ws_bytes >= ipa_segments_workspace_bytes(width, height, batch, max_segments), IPA_ERR_WORKSPACE,
                "segments: workspace too small");
    const size_t bpw = bits_pitch_words(width), bsw = bpw * (size_t)height;
    Arena ar(ws, ws_bytes);
    u32* bits = (u32*)ar.take(bpw * 4 * (size_t)height * batch);
    int32_t* vmax = (int32_t*)ar.take(sizeof(int32_t) * (size_t)batch);
    int rc = launch_pack_bits(s, bits, bpw, bsw, width, height, batch, vmax, st);
    if (rc) return rc;
    return segments_from_bits(bits, bpw, bsw, width, height, batch, max_segments, boxes, max_boxes, counts, ar, st);
}

IPA_EXPORT int ipa_segments_bits_u32(const uint32_t* bits, size_t bits_pitch_words_, size_t bits_stride_words, int width,
                                     int height, int batch, int max_segments, int32_t* boxes, int max_boxes, int32_t* counts,
                                     void* ws, size_t ws_bytes, void* stream)
{
    IPA_REQUIRE(width >= 0 && height >= 0 && batch >= 0 && max_boxes >= 0, IPA_ERR_ARG, "segments: negative size");
    if (batch == 0) return IPA_OK;
    IPA_REQUIRE(counts != nullptr && (boxes != nullptr || max_boxes == 0), IPA_ERR_ARG, "segments: NULL output");
    hipStream_t st = (hipStream_t)stream;
    if (width == 0 || height == 0) {
        IPA_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)batch, st));
        return IPA_OK;
    }
    IPA_REQUIRE(width <= 65535 && height <= 65535, IPA_ERR_ARG, "segments: image larger than 65535 pixels on a side");
    IPA_REQUIRE(bits != nullptr && ((uintptr_t)bits & 3) == 0 && bits_pitch_words_ >= (size_t)cdiv(width, 32) &&
                    (batch == 1 || bits_stride_words >= bits_pitch_words_ * (size_t)height),
                IPA_ERR_ALIGN, "segments: bit plane pointer / pitch / stride");
    IPA_REQUIRE(ws && ws_bytes >= ipa_segments_workspace_bytes(width, height, batch, max_segments), IPA_ERR_WORKSPACE,
                "segments: workspace too small");
    Arena ar(ws, ws_bytes);
    return segments_from_bits(bits, bits_pitch_words_, bits_stride_words, width, height, batch, max_segments, boxes, max_boxes,
                              counts, ar, st);
}
