// ds_add_u32 on fresh random addresses every instruction (xorshift per lane, 3 vector ops), as a function of
// the number of bins the addresses spread over, and of how many lanes of an instruction share an address.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_rand lds_rand.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0: uniform over nbins; MODE 1: uniform but lanes 2k and 2k+1 forced to different banks parity (no pair can share an address)
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, unsigned nbins_mask, int dup_lanes)
{
    extern __shared__ unsigned hist[];
    for (unsigned i = threadIdx.x; i <= nbins_mask; i += 1024) hist[i] = 0;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const unsigned lane = threadIdx.x & 63;
    for (int i = 0; i < iters; i++) {
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        unsigned a = x;
        if (MODE == 1) a = __shfl(x, (int)(lane / dup_lanes * dup_lanes));  // groups of dup_lanes lanes share an address
        atomicAdd(&hist[a & nbins_mask], 1u);
    }
    __syncthreads();
    if (hist[threadIdx.x] == 0xffffffffu) out[0] = 1;
}

template <int MODE>
static float run(unsigned* o, unsigned nbins, int dup, int wgs)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(1024), nbins * 4, 0, o, 100, nbins - 1, dup);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(1024), nbins * 4, 0, o, iters, nbins - 1, dup);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3f * 2.4e9f / (16.0f * iters * (wgs / 256));
}

int main()
{
    unsigned* o; CK(hipMalloc(&o, 4096));
    CK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (unsigned nb : {64u, 256u, 1024u, 4096u, 16384u})
        printf("uniform over %5u bins: %.2f clk per wave instruction (16 waves per CU), %.2f (32 waves)\n", nb, run<0>(o, nb, 1, 256),
               nb <= 16384 ? run<0>(o, nb, 1, 512) : 0.f);
    for (int d : {1, 2, 3, 4, 8})
        printf("16384 bins, groups of %d neighbouring lanes share an address: %.2f clk\n", d, run<1>(o, 16384, d, 256));
    return 0;
}
