// Micro-benchmark: reading interleaved BGR rows (48 bytes per 16 pixels) with lane-strided
// 16-byte loads versus contiguous loads, writing 2 x 16 bytes per lane like the gray kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int W = 3840, H = 2160, B = 256, R = 4;

template <int MODE>  // 0: lane-strided 48 B; 1: three contiguous 1 KiB loads per wave row (data not redistributed)
__global__ __launch_bounds__(256) void k(const uint8_t* src, uint8_t* g, uint8_t* t)
{
    const int cx = blockIdx.x * 64 + threadIdx.x;
    const int b = blockIdx.z;
    const int r0 = (blockIdx.y * 4 + threadIdx.y) * R;
    if (cx * 16 >= W) return;
    uint4 a[R][3];
#pragma unroll
    for (int k2 = 0; k2 < R; k2++) {
        const uint8_t* row = src + ((size_t)b * H + min(r0 + k2, H - 1)) * (W * 3);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (MODE == 0) a[k2][q] = *(const uint4*)(row + (size_t)cx * 48 + q * 16);
            else a[k2][q] = *(const uint4*)(row + (size_t)blockIdx.x * 3072 + q * 1024 + threadIdx.x * 16);
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < R; k2++) {
        if (r0 + k2 >= H) break;
        uint4 s;
        s.x = a[k2][0].x ^ a[k2][1].x ^ a[k2][2].x; s.y = a[k2][0].y ^ a[k2][1].y ^ a[k2][2].y;
        s.z = a[k2][0].z ^ a[k2][1].z ^ a[k2][2].z; s.w = a[k2][0].w ^ a[k2][1].w ^ a[k2][2].w;
        const size_t o = ((size_t)b * H + r0 + k2) * W + cx * 16;
        *(uint4*)(g + o) = s;
        *(uint4*)(t + o) = s;
    }
}

int main()
{
    uint8_t *src, *g, *t;
    CHECK(hipMalloc(&src, (size_t)W * H * 3 * B)); CHECK(hipMalloc(&g, (size_t)W * H * B)); CHECK(hipMalloc(&t, (size_t)W * H * B));
    CHECK(hipMemset(src, 3, (size_t)W * H * 3 * B));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(4, (H + 4 * R - 1) / (4 * R), B), dim3(64, 4), 0, 0, src, g, t);
            else hipLaunchKernelGGL(k<1>, dim3(4, (H + 4 * R - 1) / (4 * R), B), dim3(64, 4), 0, 0, src, g, t);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%s: %.3f ms  %.2f TB/s (5 B/px)\n", mode ? "contiguous 1 KiB loads" : "lane-strided 48 B loads", ms, 5.0 * W * H * B / ms / 1e9);
        }
    return 0;
}
