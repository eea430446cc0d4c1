// Micro-benchmark: throughput of LDS atomic adds (no return) per CU under different address patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// mode 0: lane -> consecutive words (conflict free); 1: stride 2; 2: stride 4; 3: same address;
// 4: pseudo random (changes every iteration); 5: pairs share an address; 6: stride 32 (one bank)
template <int MODE, bool RET>
__global__ __launch_bounds__(512) void k(int iters, uint32_t* out, int words)
{
    extern __shared__ uint32_t h[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t a = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint32_t idx;
            if (MODE == 0) idx = lane + u * 64 + it;
            else if (MODE == 1) idx = lane * 2 + u + it;
            else if (MODE == 2) idx = lane * 4 + u + it;
            else if (MODE == 3) idx = u + it;
            else if (MODE == 4) { a = a * 1664525u + 1013904223u; idx = a >> 12; }
            else if (MODE == 5) idx = (lane >> 1) + u * 64 + it;
            else idx = lane * 32 + u + it;
            idx &= (uint32_t)(words - 1);
            if (RET) acc += atomicAdd(&h[idx], 1u); else atomicAdd(&h[idx], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = h[1] + acc;
}

template <int MODE, bool RET>
void run(const char* name, int blocks, int threads, int words)
{
    uint32_t* out;
    CHECK(hipMalloc(&out, blocks * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int iters = 2000;
    CHECK(hipFuncSetAttribute((const void*)k<MODE, RET>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipLaunchKernelGGL((k<MODE, RET>), dim3(blocks), dim3(threads), words * 4, 0, 10, out, words);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, RET>), dim3(blocks), dim3(threads), words * 4, 0, iters, out, words);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double votes = (double)blocks * threads * iters * 8;
    printf("%-28s ret=%d blocks=%d thr=%d words=%d: %.3f ms  %.1f Gvotes/s  (%.2f votes/clk/CU @2.4GHz,256CU)\n", name, (int)RET, blocks, threads, words, ms,
           votes / ms / 1e6, votes / (ms * 1e-3) / 256 / 2.4e9);
    CHECK(hipFree(out));
}

int main()
{
    const int words = 16384;  // 64 KiB
    for (int wg = 1; wg <= 2; wg++) {
        int blocks = 256 * wg;
        run<0, false>("consecutive", blocks, 512, words);
        run<1, false>("stride2", blocks, 512, words);
        run<2, false>("stride4", blocks, 512, words);
        run<5, false>("pairs same address", blocks, 512, words);
        run<3, false>("all same address", blocks, 512, words);
        run<6, false>("stride32 (one bank)", blocks, 512, words);
        run<4, false>("random", blocks, 512, words);
        run<0, true>("consecutive", blocks, 512, words);
        run<4, true>("random", blocks, 512, words);
    }
    return 0;
}
