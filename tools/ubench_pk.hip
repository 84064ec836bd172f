// Does v_pk_fma_f32 retire two FMAs in the time of one v_fma_f32 on gfx950?  16 independent accumulators per lane,
// 4 waves/SIMD on every CU; A: 16 x v_fma_f32, B: 8 x v_pk_fma_f32 (same flops), C: 16 x v_add_u32, D: sad+alignbit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 4096
typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, float seed)
{
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = seed + threadIdx.x + i;
    float w = 1.0001f, x = 0.5f;
    unsigned ua[16];
#pragma unroll
    for (int i = 0; i < 16; i++) ua[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < N_IT; it++) {
        if (OP == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w), "v"(x));
        }
        if (OP == 1) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float2v a = {acc[i], acc[i + 1]}, ww = {w, w}, xx = {x, x};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(ww), "v"(xx));
                acc[i] = a.x;
                acc[i + 1] = a.y;
            }
        }
        if (OP == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ua[i]) : "v"(ua[(i + 1) & 15]));
        }
        if (OP == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_sad_u8 %0, %0, %1, 0" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]));
        }
        if (OP == 4) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_alignbit_b32 %0, %0, %1, 25" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]));
        }
        if (OP == 5) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(acc[i]) : "v"(ua[i]));
        }
        if (OP == 6) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mad_u32_u24 %0, %0, %2, %1" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]), "s"(128u));
        }
        if (OP == 7) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(w));
        }
        if (OP == 8) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_lshl_add_u32 %0, %0, 7, %1" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]));
        }
        if (OP == 9) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(w));
        }
        if (OP == 10) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]));
        }
        if (OP == 11) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]), "v"(ua[(i + 9) & 15]));
        }
        if (OP == 12) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]));
        }
        if (OP == 13) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]));
        }
        if (OP == 14) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]), "v"(ua[(i + 9) & 15]));
        }
        if (OP == 15) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(ua[i]));
        }
        if (OP == 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(ua[i]) : "v"(acc[i]));
        }
        if (OP == 17) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(ua[i]));
        }
        if (OP == 18) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_min_u32 %0, %0, %1" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]));
        }
        if (OP == 19) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]));
        }
        if (OP == 20) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(ua[i]) : "v"(ua[(i + 5) & 15]));
        }
    }
    float s = 0;
    unsigned u = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        s += acc[i];
        u ^= ua[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)u;
}
template <int OP> void run(const char* name, float* d, double instr_per_it, double flops_per_it)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, 2.f);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    double per = ms * 1e6 / (4.0 * N_IT * instr_per_it);
    printf("%-22s %7.3f ms  %.3f ns per wave-instr per SIMD  (%.2f clk @2.4GHz)  %.1f T lane-results/s\n", name, ms, per, per * 2.4,
           256.0 * 1024 * N_IT * flops_per_it / (ms * 1e-3) / 1e12);
}
int main()
{
    float* d;
    (void)hipMalloc(&d, 256 * 1024 * 4);
    run<0>("16 x v_fma_f32", d, 16, 16);
    run<1>("8 x v_pk_fma_f32", d, 8, 16);
    run<2>("16 x v_add_u32", d, 16, 16);
    run<3>("16 x v_sad_u8", d, 16, 16);
    run<4>("16 x v_alignbit_b32", d, 16, 16);
    run<5>("16 x v_cvt_f32_ubyte1", d, 16, 16);
    run<6>("16 x v_mad_u32_u24", d, 16, 16);
    run<7>("16 x v_mul_f32", d, 16, 16);
    run<8>("16 x v_lshl_add_u32", d, 16, 16);
    run<9>("16 x v_add_f32", d, 16, 16);
    run<10>("16 x v_pk_add_u16", d, 16, 32);
    run<11>("16 x v_pk_mad_u16", d, 16, 32);
    run<12>("16 x v_pk_max_i16", d, 16, 32);
    run<13>("16 x v_and_b32", d, 16, 16);
    run<14>("16 x v_perm_b32", d, 16, 16);
    run<15>("16 x v_lshrrev_b32", d, 16, 16);
    run<16>("16 x v_cvt_pk_u8_f32", d, 16, 16);
    run<17>("16 x v_bfe_u32", d, 16, 16);
    run<18>("16 x v_min_u32", d, 16, 16);
    run<19>("16 x v_pk_mul_lo_u16", d, 16, 32);
    run<20>("16 x v_pk_sub_i16", d, 16, 32);
    return 0;
}
