// Micro-benchmark: row-by-row strip copy with the access pattern of the streaming Canny kernel
// (one wave = a strip of 64 lanes x BYTES bytes, walks ROWS rows downwards, 3 rows prefetched).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <typename T, int ROWS, bool WRITE>
__global__ __launch_bounds__(256) void k(const uint8_t* src, uint8_t* dst, int w, int h, int pitch, size_t stride, int strips_x, int bands, long long nwaves)
{
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long wave_id = (long long)blockIdx.x * 4 + wv;
    if (wave_id >= nwaves) return;
    const int lane = threadIdx.x & 63;
    const int sx = (int)(wave_id % strips_x);
    long long t = wave_id / strips_x;
    const int band = (int)(t % bands), b = (int)(t / bands);
    const int x = min(sx * 64 * (int)sizeof(T) + lane * (int)sizeof(T), w - (int)sizeof(T));
    const int y0 = band * ROWS;
    const uint8_t* s = src + (size_t)b * stride;
    uint8_t* d = dst + (size_t)b * stride;
    T q[3];
    T acc = {};
#pragma unroll
    for (int k = 0; k < 3; k++) q[k] = *(const T*)(s + (uint32_t)(min(y0 + k, h - 1) * pitch + x));
#pragma unroll 3
    for (int it = 0; it < ROWS; it++) {
        T g = q[it % 3];
        q[it % 3] = *(const T*)(s + (uint32_t)(min(y0 + it + 3, h - 1) * pitch + x));
        if (WRITE) { if (y0 + it < h) *(T*)(d + (uint32_t)((y0 + it) * pitch + x)) = g; }
        else { const uint32_t* gp = (const uint32_t*)&g; uint32_t* ap = (uint32_t*)&acc; for (unsigned i = 0; i < sizeof(T) / 4; i++) ap[i] ^= gp[i]; }
    }
    if (!WRITE) { const uint32_t* ap = (const uint32_t*)&acc; uint32_t v = 0; for (unsigned i = 0; i < sizeof(T) / 4; i++) v ^= ap[i]; if (v == 0x12345678u) *(uint32_t*)d = v; }
}

template <typename T, int ROWS, bool WRITE>
void run(const char* name, const uint8_t* src, uint8_t* dst, int w, int h, int batch)
{
    const int pitch = w;
    const size_t stride = (size_t)w * h;
    const int strips_x = (w + 64 * (int)sizeof(T) - 1) / (64 * (int)sizeof(T)), bands = (h + ROWS - 1) / ROWS;
    const long long nwaves = (long long)strips_x * bands * batch;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k<T, ROWS, WRITE>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, 0, src, dst, w, h, pitch, stride, strips_x, bands, nwaves);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double bytes = (double)stride * batch * (WRITE ? 2 : 1);
    printf("%-34s rows/strip=%3d: %.3f ms  %.2f TB/s (%s)\n", name, ROWS, ms, bytes / ms / 1e9, WRITE ? "read+write" : "read only");
}

int main()
{
    const int w = 3840, h = 2160, batch = 256;
    uint8_t *src, *dst;
    CHECK(hipMalloc(&src, (size_t)w * h * batch)); CHECK(hipMalloc(&dst, (size_t)w * h * batch));
    CHECK(hipMemset(src, 1, (size_t)w * h * batch)); CHECK(hipMemset(dst, 0, (size_t)w * h * batch));
    run<uint32_t, 64, true>("4 B/lane (256 B per wave row)", src, dst, w, h, batch);
    run<uint2, 64, true>("8 B/lane (512 B per wave row)", src, dst, w, h, batch);
    run<uint4, 64, true>("16 B/lane (1 KiB per wave row)", src, dst, w, h, batch);
    run<uint32_t, 128, true>("4 B/lane", src, dst, w, h, batch);
    run<uint32_t, 32, true>("4 B/lane", src, dst, w, h, batch);
    run<uint32_t, 64, false>("4 B/lane", src, dst, w, h, batch);
    run<uint4, 64, false>("16 B/lane", src, dst, w, h, batch);
    CHECK(hipMemcpy(dst, src, (size_t)w * h * batch, hipMemcpyDeviceToDevice));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a)); CHECK(hipMemcpyAsync(dst, src, (size_t)w * h * batch, hipMemcpyDeviceToDevice, 0)); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemcpy D2D: %.3f ms %.2f TB/s (read+write)\n", ms, 2.0 * w * h * batch / ms / 1e9);
    return 0;
}
