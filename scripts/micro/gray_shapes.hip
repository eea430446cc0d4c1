// Micro-benchmark: work shapes for a 3-bytes-in / 2-bytes-out per pixel stream (the gray kernel's traffic:
// 48 bytes of BGR per 16 pixels in, 16 bytes to each of two planes out).
//   mode 0: the gray kernel's shape: workgroup = 4 waves x 4 rows x 1024 pixels (16 row pieces of 3 KiB)
//   mode 1: flat, one 16-pixel chunk per lane, consecutive lanes = consecutive chunks of the whole batch
//   mode 2: flat, four chunks per lane 1024 pixels apart (a wave = 4096 consecutive pixels), all loads first
//   mode 3: flat, grid-stride over chunks with 2048 workgroups, four chunks in flight per lane
//   mode 4: workgroup = 4 waves side by side on the same RR rows (RR whole rows of 3840 pixels), RR = 4, 2, 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int W = 3840, H = 2160, B = 256, R = 4;
constexpr size_t NCH = (size_t)W * H * B / 16;  // 16-pixel chunks

__device__ __forceinline__ uint4 mix(uint4 a, uint4 b, uint4 c)
{
    return make_uint4(a.x ^ b.x ^ c.x, a.y ^ b.y ^ c.y, a.z ^ b.z ^ c.z, a.w ^ b.w ^ c.w);
}

__global__ __launch_bounds__(256) void k_rows(const uint8_t* src, uint8_t* g, uint8_t* t)
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
        for (int q = 0; q < 3; q++) a[k2][q] = *(const uint4*)(row + (size_t)cx * 48 + q * 16);
    }
#pragma unroll
    for (int k2 = 0; k2 < R; k2++) {
        if (r0 + k2 >= H) break;
        const uint4 s = mix(a[k2][0], a[k2][1], a[k2][2]);
        const size_t o = ((size_t)b * H + r0 + k2) * W + cx * 16;
        *(uint4*)(g + o) = s;
        *(uint4*)(t + o) = s;
    }
}

template <int RR>  // the 4 waves of a workgroup side by side: RR whole image rows per workgroup
__global__ __launch_bounds__(256) void k_wide(const uint8_t* src, uint8_t* g, uint8_t* t)
{
    const int cx = threadIdx.y * 64 + threadIdx.x;  // chunk within the row
    const size_t r0 = (size_t)blockIdx.x * RR;      // row of the whole batch
    if (cx * 16 >= W) return;
    uint4 a[RR][3];
#pragma unroll
    for (int k2 = 0; k2 < RR; k2++) {
        const uint8_t* row = src + (r0 + k2) * (W * 3);
#pragma unroll
        for (int q = 0; q < 3; q++) a[k2][q] = *(const uint4*)(row + (size_t)cx * 48 + q * 16);
    }
#pragma unroll
    for (int k2 = 0; k2 < RR; k2++) {
        const uint4 s = mix(a[k2][0], a[k2][1], a[k2][2]);
        const size_t o = (r0 + k2) * W + cx * 16;
        *(uint4*)(g + o) = s;
        *(uint4*)(t + o) = s;
    }
}

template <int PER>  // chunks per lane, 64 chunks (1024 pixels) apart
__global__ __launch_bounds__(256) void k_flat(const uint8_t* src, uint8_t* g, uint8_t* t)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t c0 = wave * 64 * PER + (threadIdx.x & 63);
    uint4 a[PER][3];
#pragma unroll
    for (int k2 = 0; k2 < PER; k2++) {
        const size_t c = c0 + (size_t)k2 * 64;
        if (c < NCH) {
#pragma unroll
            for (int q = 0; q < 3; q++) a[k2][q] = *(const uint4*)(src + c * 48 + q * 16);
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < PER; k2++) {
        const size_t c = c0 + (size_t)k2 * 64;
        if (c < NCH) {
            const uint4 s = mix(a[k2][0], a[k2][1], a[k2][2]);
            *(uint4*)(g + c * 16) = s;
            *(uint4*)(t + c * 16) = s;
        }
    }
}

__global__ __launch_bounds__(256) void k_stride(const uint8_t* src, uint8_t* g, uint8_t* t)
{
    const size_t nthreads = (size_t)gridDim.x * 256;
    for (size_t c0 = (size_t)blockIdx.x * 256 + threadIdx.x; c0 < NCH; c0 += 4 * nthreads) {
        uint4 a[4][3];
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            const size_t c = c0 + k2 * nthreads;
            if (c < NCH) {
#pragma unroll
                for (int q = 0; q < 3; q++) a[k2][q] = *(const uint4*)(src + c * 48 + q * 16);
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            const size_t c = c0 + k2 * nthreads;
            if (c < NCH) {
                const uint4 s = mix(a[k2][0], a[k2][1], a[k2][2]);
                *(uint4*)(g + c * 16) = s;
                *(uint4*)(t + c * 16) = s;
            }
        }
    }
}

int main()
{
    uint8_t *src, *g, *t;
    CHECK(hipMalloc(&src, (size_t)W * H * 3 * B)); CHECK(hipMalloc(&g, (size_t)W * H * B)); CHECK(hipMalloc(&t, (size_t)W * H * B));
    CHECK(hipMemset(src, 3, (size_t)W * H * 3 * B));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[7] = {"rows 4x4x1024 (gray kernel)", "flat, 1 chunk per lane", "flat, 4 chunks per lane", "flat, grid-stride 2048 wgs",
                            "4 whole rows per workgroup", "2 whole rows per workgroup", "1 whole row per workgroup"};
    for (int mode = 0; mode < 7; mode++)
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_rows, dim3(4, (H + 4 * R - 1) / (4 * R), B), dim3(64, 4), 0, 0, src, g, t);
            else if (mode == 1) hipLaunchKernelGGL(k_flat<1>, dim3((unsigned)((NCH + 255) / 256)), dim3(256), 0, 0, src, g, t);
            else if (mode == 2) hipLaunchKernelGGL(k_flat<4>, dim3((unsigned)((NCH + 1023) / 1024)), dim3(256), 0, 0, src, g, t);
            else if (mode == 3) hipLaunchKernelGGL(k_stride, dim3(2048), dim3(256), 0, 0, src, g, t);
            else if (mode == 4) hipLaunchKernelGGL(k_wide<4>, dim3((unsigned)((size_t)H * B / 4)), dim3(64, 4), 0, 0, src, g, t);
            else if (mode == 5) hipLaunchKernelGGL(k_wide<2>, dim3((unsigned)((size_t)H * B / 2)), dim3(64, 4), 0, 0, src, g, t);
            else hipLaunchKernelGGL(k_wide<1>, dim3((unsigned)((size_t)H * B)), dim3(64, 4), 0, 0, src, g, t);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("%-30s %.3f ms  %.2f TB/s (5 B/px)\n", names[mode], ms, 5.0 * W * H * B / ms / 1e9);
        }
    return 0;
}
