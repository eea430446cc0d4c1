// Raw rate of ds_add_u32 (no return) with the addresses held in registers, so that no vector ALU work
// stands between two atomics: eight addresses per lane, issued round robin.
//   pattern 0: consecutive words (lane i -> base + i): no bank conflicts, no shared addresses
//   pattern 1: random words
//   pattern 2: sorted with shared addresses, lane i -> base + floor(i * c)
//   pattern 3: pattern 2 with only the first lane of each address active (adds the group's count)
//   pattern 4: consecutive words, only lanes < active
// build: hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(unsigned* out, int iters, int pattern, float c, int active)
{
    __shared__ unsigned hist[16384];
    for (unsigned i = threadIdx.x; i < 16384; i += 64 * WAVES) hist[i] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned a[8];
    bool on = true;
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
    for (int j = 0; j < 8; j++) {
        const unsigned base = wave * 977u + j * 2053u;
        unsigned off = lane;
        if (pattern == 1) {
            x = x * 1664525u + 1013904223u;
            off = x >> 10;
        } else if (pattern == 2 || pattern == 3) {
            off = (unsigned)floorf((float)lane * c);
        }
        a[j] = (base + off) & 16383u;
    }
    if (pattern == 3) on = lane == 0 || (unsigned)floorf((float)lane * c) != (unsigned)floorf((float)(lane - 1) * c);
    if (pattern == 4) on = (int)lane < active;
    if (on) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) atomicAdd(&hist[a[j]], 1u);
        }
    }
    __syncthreads();
    if (hist[threadIdx.x] == 0xffffffffu) out[0] = 1;
}

template <int WAVES>
static float run(unsigned* o, int pattern, float c, int active)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    hipLaunchKernelGGL(k<WAVES>, dim3(256), dim3(64 * WAVES), 0, 0, o, 50, pattern, c, active);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<WAVES>, dim3(256), dim3(64 * WAVES), 0, 0, o, iters, pattern, c, active);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3f * 2.4e9f / ((float)WAVES * iters * 8);  // clk per wave instruction and CU at 2.4 GHz
}

int main()
{
    unsigned* o; CK(hipMalloc(&o, 4096));
    printf("waves per CU: 16 / 8 / 4\n");
    printf("consecutive: %.2f / %.2f / %.2f clk per wave instruction\n", run<16>(o, 0, 0, 0), run<8>(o, 0, 0, 0), run<4>(o, 0, 0, 0));
    printf("random:      %.2f / %.2f / %.2f\n", run<16>(o, 1, 0, 0), run<8>(o, 1, 0, 0), run<4>(o, 1, 0, 0));
    for (float c : {0.95f, 0.8f, 0.7f, 0.5f, 0.4f, 0.3f, 0.2f, 0.1f, 0.0f})
        printf("sorted floor(i * %.2f): all lanes %.2f clk; first lane of each address only %.2f clk\n", c, run<16>(o, 2, c, 0),
               run<16>(o, 3, c, 0));
    for (float c : {1.5f, 2.0f, 2.83f, 4.0f, 8.0f})
        printf("strided floor(i * %.2f): %.2f clk\n", c, run<16>(o, 2, c, 0));
    for (int a : {64, 48, 32, 16, 8, 1})
        printf("consecutive, %2d active lanes: %.2f clk\n", a, run<16>(o, 4, 0, a));
    return 0;
}
