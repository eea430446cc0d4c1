// Cost of ds_add_u32 when groups of lanes share an address: G distinct addresses per wave
// instruction (in G different banks, or all in one bank), 64 / G lanes on each.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_same lds_same.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, int groups, int same_bank)
{
    __shared__ unsigned hist[16384];
    for (unsigned i = threadIdx.x; i < 16384; i += 1024) hist[i] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned g = lane % (unsigned)groups;
    // group g of wave w: its own address; same_bank: addresses 32 words apart
    unsigned idx = (wave * 64 + g * (same_bank ? 32u : 1u)) & 16383u;
    for (int i = 0; i < iters; i++) {
        atomicAdd(&hist[idx], 1u);
        idx = (idx + 2048u) & 16383u;  // another address of the same bank next time
    }
    __syncthreads();
    if (hist[threadIdx.x] == 0xffffffffu) out[0] = 1;
}

int main()
{
    unsigned* o; CK(hipMalloc(&o, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int sb = 0; sb < 2; sb++)
        for (int g : {1, 2, 4, 8, 16, 32, 64}) {
            const int iters = 20000;
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, o, 100, g, sb);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, o, iters, g, sb);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // 16 waves per CU, iters instructions each
            printf("%s, %2d addresses x %2d lanes: %.1f clk per wave instruction\n", sb ? "one bank  " : "own banks ", g, 64 / g,
                   ms * 1e-3 * 2.1e9 / (16.0 * iters));
        }
    return 0;
}
