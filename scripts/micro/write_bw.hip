// Micro-benchmark: write-only bandwidth for the store patterns of the plane-producing kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int W = 3840, H = 2160, B = 256;

__global__ __launch_bounds__(256) void k_flat(uint4* dst, size_t n16, uint4 v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = v;
}
// thread (tx,ty): 16-byte chunk column 64*bx+tx, rows (4*by+ty)*R .. +R
template <int R>
__global__ __launch_bounds__(256) void k_rows(uint8_t* dst, uint4 v)
{
    const int cx = blockIdx.x * 64 + threadIdx.x;
    if (cx * 16 >= W) return;
    uint8_t* img = dst + (size_t)blockIdx.z * W * H;
    const int r0 = (blockIdx.y * 4 + threadIdx.y) * R;
#pragma unroll
    for (int k = 0; k < R; k++)
        if (r0 + k < H) *(uint4*)(img + (size_t)(r0 + k) * W + cx * 16) = v;
}
// one wave = 64 lanes x BYTES bytes wide, walks ROWS rows downwards
template <int ROWS, typename T>
__global__ __launch_bounds__(256) void k_walk(uint8_t* dst, T v)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int strips = (W + 64 * (int)sizeof(T) - 1) / (64 * (int)sizeof(T));
    const int bands = (H + ROWS - 1) / ROWS;
    const long long wave = (long long)blockIdx.x * 4 + wv;
    if (wave >= (long long)strips * bands * B) return;
    const int sx = wave % strips;
    const int band = (wave / strips) % bands, b = wave / strips / bands;
    const int x = sx * 64 * (int)sizeof(T) + lane * (int)sizeof(T);
    if (x >= W) return;
    uint8_t* img = dst + (size_t)b * W * H;
    for (int y = band * ROWS; y < min(H, band * ROWS + ROWS); y++) *(T*)(img + (size_t)y * W + x) = v;
}

template <typename F>
void timeit(const char* name, F f)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-46s %.3f ms  %.2f TB/s\n", name, ms, (double)W * H * B / ms / 1e9);
}

int main()
{
    uint8_t* dst;
    const size_t n = (size_t)W * H * B;
    CHECK(hipMalloc(&dst, n));
    const uint4 v = make_uint4(1, 2, 3, 4);
    timeit("hipMemsetAsync", [&] { CHECK(hipMemsetAsync(dst, 1, n, 0)); });
    timeit("flat grid-stride 16 B/lane, 4096 blocks", [&] { hipLaunchKernelGGL(k_flat, dim3(4096), dim3(256), 0, 0, (uint4*)dst, n / 16, v); });
    timeit("flat 16 B/lane, one pass (n/4096 blocks)", [&] { hipLaunchKernelGGL(k_flat, dim3((unsigned)(n / 16 / 256)), dim3(256), 0, 0, (uint4*)dst, n / 16, v); });
    timeit("2-D rows, 4 rows/lane", [&] { hipLaunchKernelGGL(k_rows<4>, dim3(4, (H + 15) / 16, B), dim3(64, 4), 0, 0, dst, v); });
    timeit("2-D rows, 8 rows/lane", [&] { hipLaunchKernelGGL(k_rows<8>, dim3(4, (H + 31) / 32, B), dim3(64, 4), 0, 0, dst, v); });
    timeit("2-D rows, 1 row/lane", [&] { hipLaunchKernelGGL(k_rows<1>, dim3(4, (H + 3) / 4, B), dim3(64, 4), 0, 0, dst, v); });
    {
        const int strips = (W + 1023) / 1024;
        timeit("walk 128 rows, 16 B/lane", [&] { hipLaunchKernelGGL((k_walk<128, uint4>), dim3((unsigned)(((long long)strips * ((H + 127) / 128) * B + 3) / 4)), dim3(256), 0, 0, dst, v); });
        timeit("walk 32 rows, 16 B/lane", [&] { hipLaunchKernelGGL((k_walk<32, uint4>), dim3((unsigned)(((long long)strips * ((H + 31) / 32) * B + 3) / 4)), dim3(256), 0, 0, dst, v); });
    }
    {
        const int strips = (W + 255) / 256;
        timeit("walk 64 rows, 4 B/lane", [&] { hipLaunchKernelGGL((k_walk<64, uint32_t>), dim3((unsigned)(((long long)strips * ((H + 63) / 64) * B + 3) / 4)), dim3(256), 0, 0, dst, 7u); });
    }
    return 0;
}
