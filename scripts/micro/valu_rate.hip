// Issue rate of the vector instructions the Hough vote is made of (wave64, 16 waves per CU,
// eight independent chains per lane).  build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float c)
{
    f2 a[8];
    for (int j = 0; j < 8; j++) a[j] = f2{(float)threadIdx.x + j, (float)j + 0.5f};
    const f2 cc = {c, c * 1.0001f};
    unsigned u[8];
    for (int j = 0; j < 8; j++) u[j] = threadIdx.x + j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[j].x) : "v"(cc.x));
            if (OP == 1) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[j]) : "v"(cc));
            if (OP == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[j]) : "v"(cc));
            if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(a[j]) : "v"(cc));
            if (OP == 4) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(a[j].x) : "v"(cc.x));
            if (OP == 5) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            if (OP == 6) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[j].x) : "v"(u[j]));
            if (OP == 7) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[j].x) : "v"(cc.x));
            if (OP == 8) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[j]) : "s"(cc));
            if (OP == 9) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            if (OP == 10) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            if (OP == 11) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            if (OP == 12) asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(a[j].x) : "v"(u[j]));
        }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += a[j].x + a[j].y + (float)u[j];
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
static void run(const char* name, float* o)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, o, 100, 1.0000001f);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, o, iters, 1.0000001f);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // 16 waves per CU = 4 per SIMD; clk per wave instruction on one SIMD
    printf("%-28s %.2f clk per wave instruction and SIMD (2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / (4.0 * iters * 8));
}

int main()
{
    float* o; CK(hipMalloc(&o, 4096));
    run<0>("v_mul_f32", o); run<7>("v_add_f32", o); run<4>("v_fma_f32", o);
    run<1>("v_pk_mul_f32", o); run<8>("v_pk_mul_f32 (sgpr pair)", o); run<2>("v_pk_add_f32", o); run<3>("v_pk_fma_f32", o);
    run<5>("v_lshl_add_u32", o); run<9>("v_add_u32", o); run<6>("v_cvt_f32_u32 sdwa", o);
    run<10>("v_pk_add_u16", o); run<11>("v_mad_u32_u24", o); run<12>("v_dot2_f32_f16", o);
    return 0;
}
