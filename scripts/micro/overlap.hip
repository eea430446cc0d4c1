// Do two kernels of different streams share the CUs?  Three synthetic kernels, each bound by one
// unit (vector ALU, HBM streaming, LDS atomics), timed alone and in pairs on two streams.
// build: hipcc --offload-arch=gfx950 -O3 -o overlap overlap.hip ; run: ./overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_alu(float* out, int iters)
{
    float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; i++) {
        a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 0.00001f;
        a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 0.00001f;
    }
    if (a + c + d == 123.456f) out[0] = a;
}

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + k * 256 < n) dst[i + k * 256] = src[i + k * 256];
}

// persistent when the grid is smaller than the number of items; optional accumulator flush to
// global memory (flush) and optional streaming list read (list != nullptr), like the vote kernel
template <int NT>
__global__ __launch_bounds__(NT) void k_lds(unsigned* out, int iters, int items, unsigned* flush, const uint4* list)
{
    extern __shared__ unsigned hist[];
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        for (int i = threadIdx.x; i < 16384; i += NT) hist[i] = 0;
        __syncthreads();
        unsigned x = threadIdx.x * 2654435761u + it;
        if (list) {
            const uint4* l = list + (size_t)(it % 64) * 40000;
            for (int i = 0; i < iters / 4; i++) {
                const uint4 p = l[i * NT + threadIdx.x];
                atomicAdd(&hist[(p.x >> 10) & 16383], 1u);
                atomicAdd(&hist[(p.y >> 10) & 16383], 1u);
                atomicAdd(&hist[(p.z >> 10) & 16383], 1u);
                atomicAdd(&hist[(p.w >> 10) & 16383], 1u);
            }
        } else {
            for (int i = 0; i < iters; i++) {
                x = x * 1664525u + 1013904223u;
                atomicAdd(&hist[(x >> 10) & 16383], 1u);
            }
        }
        __syncthreads();
        if (flush) {
            unsigned* row = flush + (size_t)it * 49152;  // 192 KiB per item
            for (int i = threadIdx.x; i < 49152; i += NT) row[i] = hist[i & 16383];
        }
        if (hist[threadIdx.x] == 0xffffffffu) out[0] = 1;
        __syncthreads();
    }
}

int main()
{
    const size_t n16 = (size_t)1 << 27;  // 2 GiB in, 2 GiB out
    uint4 *src, *dst; float* o; unsigned* o2;
    CK(hipMalloc(&src, n16 * 16)); CK(hipMalloc(&dst, n16 * 16)); CK(hipMalloc(&o, 4096)); CK(hipMalloc(&o2, 4096));
    CK(hipMemset(src, 1, n16 * 16));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k_lds<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    auto alu = [&](hipStream_t s) { hipLaunchKernelGGL(k_alu, dim3(256 * 64), dim3(256), 0, s, o, 6000); };
    auto cpy = [&](hipStream_t s) { hipLaunchKernelGGL(k_copy, dim3((unsigned)(n16 / 1024)), dim3(256), 0, s, src, dst, n16); };
    const int items = 256 * 24;
    unsigned* fl; uint4* list;
    CK(hipMalloc(&fl, (size_t)items * 49152 * 4)); CK(hipMalloc(&list, (size_t)64 * 40000 * 16 + 1024 * 600 * 16));
    {
        const size_t nw = ((size_t)64 * 40000 * 16 + 1024 * 600 * 16) / 4;
        unsigned* hl = (unsigned*)malloc(nw * 4);
        unsigned x = 12345u;
        for (size_t i = 0; i < nw; i++) { x = x * 1664525u + 1013904223u; hl[i] = x; }
        CK(hipMemcpy(list, hl, nw * 4, hipMemcpyHostToDevice));
        free(hl);
    }
    auto lds = [&](hipStream_t s) { hipLaunchKernelGGL(k_lds<1024>, dim3(items), dim3(1024), 64 * 1024, s, o2, 2000, items, nullptr, nullptr); };
    auto ldsr = [&](hipStream_t s) { hipLaunchKernelGGL(k_lds<1024>, dim3(256), dim3(1024), 90 * 1024, s, o2, 2000, items, nullptr, nullptr); };
    auto ldsr_f = [&](hipStream_t s) { hipLaunchKernelGGL(k_lds<1024>, dim3(256), dim3(1024), 90 * 1024, s, o2, 2000, items, fl, nullptr); };
    auto ldsr_l = [&](hipStream_t s) { hipLaunchKernelGGL(k_lds<1024>, dim3(256), dim3(1024), 90 * 1024, s, o2, 2000, items, nullptr, list); };
    auto ldsr_fl = [&](hipStream_t s) { hipLaunchKernelGGL(k_lds<1024>, dim3(256), dim3(1024), 90 * 1024, s, o2, 2000, items, fl, list); };
    struct { const char* name; decltype(alu)* f; } dummy; (void)dummy;
    auto timed = [&](auto fa, auto fb, bool two) {
        float best = 1e9f;
        for (int r = 0; r < 3; r++) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            CK(hipStreamWaitEvent(s1, e0, 0)); CK(hipStreamWaitEvent(s2, e0, 0));
            fa(s1);
            if (two) fb(s2);
            hipEvent_t j1, j2; CK(hipEventCreate(&j1)); CK(hipEventCreate(&j2));
            CK(hipEventRecord(j1, s1)); CK(hipEventRecord(j2, s2));
            CK(hipStreamWaitEvent(0, j1, 0)); CK(hipStreamWaitEvent(0, j2, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        return best;
    };
    const float ta = timed(alu, alu, false), tc = timed(cpy, cpy, false), tl = timed(lds, lds, false), tr = timed(ldsr, ldsr, false);
    printf("alone: alu %.3f  copy %.3f (%.0f GB/s)  lds %.3f  lds-resident %.3f ms\n", ta, tc, n16 * 32 / tc / 1e6, tl, tr);
    printf("alu+copy  %.3f (sum %.3f)\n", timed(alu, cpy, true), ta + tc);
    printf("copy+alu  %.3f (sum %.3f)\n", timed(cpy, alu, true), ta + tc);
    printf("lds+copy  %.3f (sum %.3f)\n", timed(lds, cpy, true), tl + tc);
    printf("ldsr+copy %.3f (sum %.3f)\n", timed(ldsr, cpy, true), tr + tc);
    const float tf = timed(ldsr_f, ldsr_f, false), tl2 = timed(ldsr_l, ldsr_l, false), tfl = timed(ldsr_fl, ldsr_fl, false);
    auto cpy3 = [&](hipStream_t s) { cpy(s); cpy(s); cpy(s); };
    const float tc3 = timed(cpy3, cpy3, false);
    printf("alone: ldsr+flush %.3f  ldsr+list %.3f  ldsr+flush+list %.3f  copy3 %.3f\n", tf, tl2, tfl, tc3);
    printf("ldsr      +copy3 %.3f (sum %.3f)\n", timed(ldsr, cpy3, true), tr + tc3);
    printf("ldsr flush+copy3 %.3f (sum %.3f)\n", timed(ldsr_f, cpy3, true), tf + tc3);
    printf("ldsr list +copy3 %.3f (sum %.3f)\n", timed(ldsr_l, cpy3, true), tl2 + tc3);
    printf("ldsr f+l  +copy3 %.3f (sum %.3f)\n", timed(ldsr_fl, cpy3, true), tfl + tc3);
    printf("lds+alu   %.3f (sum %.3f)\n", timed(lds, alu, true), tl + ta);
    printf("ldsr+alu  %.3f (sum %.3f)\n", timed(ldsr, alu, true), tr + ta);
    return 0;
}
