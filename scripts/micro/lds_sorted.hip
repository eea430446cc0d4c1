// Cost of ds_add_u32 for the address patterns a run-ordered Hough vote would produce:
// lane i -> base + floor(i * c) (sorted, groups of lanes share an address), strided patterns,
// partially active waves with distinct consecutive addresses, and random addresses for reference.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_sorted lds_sorted.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// mode 0: offset = floor(lane * c); mode 1: only lanes < active vote, offset = lane;
// mode 2: random; mode 3: offset = floor(lane * c) but lanes that repeat the previous lane's address stay idle
// mode 4: like 3 but with a count (same instruction count, fewer active lanes)
template <int mode>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, float c, int active)
{
    __shared__ unsigned hist[16384];
    for (unsigned i = threadIdx.x; i < 16384; i += 1024) hist[i] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned off = (unsigned)floorf((float)lane * c);
    const unsigned prev = lane ? (unsigned)floorf((float)(lane - 1) * c) : 0xffffffffu;
    unsigned base = wave * 977u + blockIdx.x * 131u;
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
#pragma unroll 4
    for (int i = 0; i < iters; i++) {
        if (mode == 0) {
            atomicAdd(&hist[(base + off) & 16383u], 1u);
        } else if (mode == 1) {
            if ((int)lane < active) atomicAdd(&hist[(base + lane) & 16383u], 1u);
        } else if (mode == 2) {
            x = x * 1664525u + 1013904223u;
            atomicAdd(&hist[(x >> 10) & 16383u], 1u);
        } else {
            if (off != prev) atomicAdd(&hist[(base + off) & 16383u], 3u);
        }
        base += 2053u;
    }
    __syncthreads();
    if (hist[threadIdx.x] == 0xffffffffu) out[0] = 1;
}

static float run(unsigned* o, int mode, float c, int active)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
#define LAUNCH(n) do { switch (mode) { \
        case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, o, n, c, active); break; \
        case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, o, n, c, active); break; \
        case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, o, n, c, active); break; \
        default: hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, o, n, c, active); } } while (0)
    LAUNCH(100);
    CK(hipEventRecord(e0, 0));
    LAUNCH(iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3f * 2.4e9f / (16.0f * iters);  // clk per wave instruction at 2.4 GHz, 16 waves per CU
}

int main()
{
    unsigned* o; CK(hipMalloc(&o, 4096));
    printf("random addresses: %.2f clk per wave instruction\n", run(o, 2, 0, 0));
    for (float c : {1.0f, 0.95f, 0.9f, 0.8f, 0.7f, 0.6f, 0.5f, 0.4f, 0.3f, 0.2f, 0.1f, 0.05f, 0.0f})
        printf("sorted, lane i -> floor(i * %.2f): all lanes %.2f clk; first lane of each address only %.2f clk\n", c, run(o, 0, c, 0),
               run(o, 3, c, 0));
    for (float c : {1.5f, 2.0f, 2.83f, 3.0f, 4.0f, 5.0f, 8.0f, 16.0f, 17.0f, 32.0f, 33.0f})
        printf("strided, lane i -> floor(i * %.2f): %.2f clk\n", c, run(o, 0, c, 0));
    for (int a : {64, 48, 32, 16, 8, 1})
        printf("consecutive addresses, %2d active lanes: %.2f clk\n", a, run(o, 1, 0, a));
    return 0;
}
