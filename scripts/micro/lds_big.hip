// LDS atomic rate of one 1024-thread workgroup per CU as a function of the histogram size
// (bins beyond 64 KiB of LDS?).  build: hipcc --offload-arch=gfx950 -O3 -o lds_big lds_big.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, unsigned nbins)
{
    extern __shared__ unsigned hist[];
    for (unsigned i = threadIdx.x; i < nbins; i += 1024) hist[i] = 0;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        atomicAdd(&hist[(unsigned)(((unsigned long long)(x >> 4) * nbins) >> 28)], 1u);
    }
    __syncthreads();
    if (hist[threadIdx.x] == 0xffffffffu) out[0] = 1;
}

int main()
{
    unsigned* o; CK(hipMalloc(&o, 4096));
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned sizes[] = {4096, 8192, 16383, 16384, 16385, 20000, 23173, 30000, 38000};
    for (unsigned nb : sizes) {
        const int iters = 20000;
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), nb * 4, 0, o, 100, nb);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), nb * 4, 0, o, iters, nb);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("bins %6u (%6.1f KiB): %.3f ms  %.2f lane-atomics/clk/CU (2.1 GHz)\n", nb, nb * 4 / 1024.0, ms,
               1024.0 * iters / (ms * 1e-3 * 2.1e9));
    }
    return 0;
}
