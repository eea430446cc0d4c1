// The inner loop of the Hough vote on synthetic random points, with parts switched off, to see what
// bounds it: VARIANT 0 = float math + ds_add (the real loop), 1 = float math only (bins folded into a
// register), 2 = ds_add only (address = two integer ops on the loaded word), 3 = as 0 but the four
// ds_add of an angle issued behind the math of the NEXT angle (software pipelined).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o vote_loop vote_loop.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u32;
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int TG = 4, SPAN = 4416;

struct Trig { float c[TG], s[TG]; int k4[TG]; float m[TG]; };

template <int VARIANT, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k(const uint4* __restrict__ pts4, u32 trips, Trig T, u32* out)
{
    extern __shared__ __align__(16) u32 hist[];
    const int tid = threadIdx.x;
    for (int i = tid; i < TG * SPAN; i += NT) hist[i] = 0;
    __syncthreads();
    char* const hb = (char*)hist;
    const f2 magic = {12582912.0f, 12582912.0f};
    u32 fold = 0;
    uint4 buf[2];
    buf[0] = pts4[tid];
    buf[1] = pts4[NT + tid];
    for (u32 i = 0; i + 2 <= trips; i += 2) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint4 cur = buf[k];
            u32 nx = i + 2 + k;
            nx = nx < trips ? nx : trips - 1;
            buf[k] = pts4[(size_t)nx * NT + tid];
            if (VARIANT == 2) {
#pragma unroll
                for (int g = 0; g < TG; g++) {
                    atomicAdd((u32*)(hb + (((cur.x >> g) & 0x3ffcu) + g * SPAN * 4)), 1u);
                    atomicAdd((u32*)(hb + (((cur.y >> g) & 0x3ffcu) + g * SPAN * 4)), 1u);
                    atomicAdd((u32*)(hb + (((cur.z >> g) & 0x3ffcu) + g * SPAN * 4)), 1u);
                    atomicAdd((u32*)(hb + (((cur.w >> g) & 0x3ffcu) + g * SPAN * 4)), 1u);
                }
                continue;
            }
            if (VARIANT >= 4) {
                // scalar two-operand instructions only; VARIANT 5: the bin's byte address by one 24-bit multiply
                const float xs[4] = {(float)(cur.x & 0xffffu), (float)(cur.y & 0xffffu), (float)(cur.z & 0xffffu), (float)(cur.w & 0xffffu)};
                const float ys[4] = {(float)(cur.x >> 16), (float)(cur.y >> 16), (float)(cur.z >> 16), (float)(cur.w >> 16)};
#pragma unroll
                for (int g = 0; g < TG; g++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float r;
                        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "s"(T.c[g]), "v"(xs[q]));
                        float t;
                        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "s"(T.s[g]), "v"(ys[q]));
                        asm volatile("v_add_f32 %0, %1, %0" : "+v"(r) : "v"(t));
                        if (VARIANT == 4) {
                            asm volatile("v_add_f32 %0, %1, %0" : "+v"(r) : "s"(12582912.0f));
                            atomicAdd((u32*)(hb + (((u32)__float_as_int(r) << 2) + (u32)T.k4[g])), 1u);
                        } else {
                            asm volatile("v_add_f32 %0, %1, %0" : "+v"(r) : "s"(T.m[g]));
                            u32 a;
                            asm volatile("v_mul_u32_u24 %0, 4, %1" : "=v"(a) : "v"(r));
                            atomicAdd((u32*)(hb + a), 1u);
                        }
                    }
                }
                continue;
            }
            const f2 x01 = {(float)(cur.x & 0xffffu), (float)(cur.y & 0xffffu)};
            const f2 x23 = {(float)(cur.z & 0xffffu), (float)(cur.w & 0xffffu)};
            const f2 y01 = {(float)(cur.x >> 16), (float)(cur.y >> 16)};
            const f2 y23 = {(float)(cur.z >> 16), (float)(cur.w >> 16)};
            u32 pa[4] = {0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < TG; g++) {
                const f2 r01 = (x01 * T.c[g] + y01 * T.s[g]) + magic;
                const f2 r23 = (x23 * T.c[g] + y23 * T.s[g]) + magic;
                const u32 a0 = ((u32)__float_as_int(r01.x) << 2) + (u32)T.k4[g], a1 = ((u32)__float_as_int(r01.y) << 2) + (u32)T.k4[g];
                const u32 a2 = ((u32)__float_as_int(r23.x) << 2) + (u32)T.k4[g], a3 = ((u32)__float_as_int(r23.y) << 2) + (u32)T.k4[g];
                if (VARIANT == 1) {
                    fold ^= a0 ^ a1 ^ a2 ^ a3;
                } else if (VARIANT == 0) {
                    atomicAdd((u32*)(hb + a0), 1u);
                    atomicAdd((u32*)(hb + a1), 1u);
                    atomicAdd((u32*)(hb + a2), 1u);
                    atomicAdd((u32*)(hb + a3), 1u);
                } else {
                    if (g > 0) {
                        atomicAdd((u32*)(hb + pa[0]), 1u);
                        atomicAdd((u32*)(hb + pa[1]), 1u);
                        atomicAdd((u32*)(hb + pa[2]), 1u);
                        atomicAdd((u32*)(hb + pa[3]), 1u);
                    }
                    pa[0] = a0; pa[1] = a1; pa[2] = a2; pa[3] = a3;
                }
            }
            if (VARIANT == 3) {
                atomicAdd((u32*)(hb + pa[0]), 1u);
                atomicAdd((u32*)(hb + pa[1]), 1u);
                atomicAdd((u32*)(hb + pa[2]), 1u);
                atomicAdd((u32*)(hb + pa[3]), 1u);
            }
        }
    }
    __syncthreads();
    u32 sum = fold;
    for (int i = tid; i < TG * SPAN; i += NT) sum += hist[i];
    if (sum == 0x12345678u) out[0] = sum;
}

template <int VARIANT, int NT>
static void run(const char* name, const uint4* pts, u32 npts, Trig T, u32* out, int wgs)
{
    CK(hipFuncSetAttribute((const void*)k<VARIANT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u32 trips = npts / 4 / NT;
    const size_t lds = TG * SPAN * 4;
    hipLaunchKernelGGL((k<VARIANT, NT>), dim3(wgs), dim3(NT), lds, 0, pts, trips, T, out);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<VARIANT, NT>), dim3(wgs), dim3(NT), lds, 0, pts, trips, T, out);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double vote_instr_per_cu = (double)wgs / 256.0 * trips * (NT / 64) * 4 * TG;
    printf("%-44s %7.3f ms  %.2f clk per vote instruction and CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / vote_instr_per_cu);
}

int main()
{
    const u32 npts = 1 << 20;
    u32* h = (u32*)malloc(npts * 4);
    srand(1);
    for (u32 i = 0; i < npts; i++) h[i] = ((u32)(rand() % 2160) << 16) | (u32)(rand() % 3840);
    u32 *d, *o; CK(hipMalloc(&d, npts * 4)); CK(hipMalloc(&o, 4096));
    CK(hipMemcpy(d, h, npts * 4, hipMemcpyHostToDevice));
    Trig T;
    for (int g = 0; g < TG; g++) {
        const float th = (40 + g) * (float)(M_PI / 180);
        T.c[g] = cosf(th); T.s[g] = sinf(th);
        T.k4[g] = (int)((unsigned)(0 - 0x4B400000 + g * SPAN) * 4u);  // rho >= 0 for these angles
        T.m[g] = 8388608.0f + (float)(g * SPAN);
    }
    run<0, 1024>("math + ds_add, 2 x 1024 threads per CU", (const uint4*)d, npts, T, o, 512);
    run<1, 1024>("math only", (const uint4*)d, npts, T, o, 512);
    run<2, 1024>("ds_add only (2 int ops per vote)", (const uint4*)d, npts, T, o, 512);
    run<3, 1024>("math + ds_add, software pipelined by angle", (const uint4*)d, npts, T, o, 512);
    run<4, 1024>("two-operand scalar math + ds_add", (const uint4*)d, npts, T, o, 512);
    run<5, 1024>("same, address by v_mul_u32_u24", (const uint4*)d, npts, T, o, 512);
    run<5, 512>("same, 2 x 512 threads per CU", (const uint4*)d, npts, T, o, 512);
    run<5, 256>("same, 2 x 256 threads per CU", (const uint4*)d, npts, T, o, 512);
    run<0, 1024>("math + ds_add, 1 x 1024 threads per CU", (const uint4*)d, npts, T, o, 256);
    run<0, 512>("math + ds_add, 2 x 512 threads per CU", (const uint4*)d, npts, T, o, 512);
    run<0, 512>("math + ds_add, 4 x 512 threads per CU (LDS 2 fit)", (const uint4*)d, npts, T, o, 1024);
    run<0, 256>("math + ds_add, 2 x 256 threads per CU", (const uint4*)d, npts, T, o, 512);
    return 0;
}
