// Micro-benchmark: issue cost of the VALU ops the bilateral kernel is made of, at 4 waves/SIMD on every CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 4096
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned* out, unsigned seed)
{
    unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77u, a3 = a1 + 1234567u;
    float f0 = (float)(a0 & 255), f1 = (float)(a1 & 255), f2 = 1.0001f, f3 = 0.9999f;
    for (int i = 0; i < N_IT; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) { a0 = __builtin_amdgcn_sad_u8(a0, a1, a2); a1 = __builtin_amdgcn_sad_u8(a1, a2, a3); a2 = __builtin_amdgcn_sad_u8(a2, a3, a0); a3 = __builtin_amdgcn_sad_u8(a3, a0, a1); }
            if (OP == 1) { a0 = (a0 << 7) + a1; a1 = (a1 << 7) + a2; a2 = (a2 << 7) + a3; a3 = (a3 << 7) + a0; }
            if (OP == 2) { f0 = f0 * f2; f1 = f1 * f3; f2 = f2 * f3; f3 = f3 * f0; }
            if (OP == 3) { f0 = f0 + f2; f1 = f1 + f3; f2 = f2 + f3; f3 = f3 + f0; }
            if (OP == 4) { f0 = (float)(a0 & 255); f1 = (float)((a1 >> 8) & 255); f2 = (float)((a2 >> 16) & 255); f3 = (float)(a3 >> 24);
                           a0 += __float_as_uint(f1); a1 += __float_as_uint(f2); a2 += __float_as_uint(f3); a3 += __float_as_uint(f0); }
            if (OP == 5) { a0 = a0 + a1; a1 = a1 + a2; a2 = a2 + a3; a3 = a3 + a0; }
            if (OP == 7) { a0 = (a0 << 7); a1 = (a1 << 5); a2 = a2 >> 3; a3 = a3 >> 9; a0 += 0x1234567u; a1 += 0x7654321u; a2 += 0xfedcba9u; a3 += 0x89abcdeu; }
            if (OP == 8) { a0 = (a0 << 7) | a1; a1 = (a1 << 7) | a2; a2 = (a2 << 7) | a3; a3 = (a3 << 7) | a0; }
            if (OP == 9) { a0 = __builtin_amdgcn_ubfe(a1, 8, 8) + a0; a1 = __builtin_amdgcn_ubfe(a2, 16, 8) + a1; a2 = __builtin_amdgcn_ubfe(a3, 8, 8) + a2; a3 = __builtin_amdgcn_ubfe(a0, 16, 8) + a3; }
            if (OP == 10) { a0 = (a0 & 0xFFFFFF) * 128u + a1; a1 = (a1 & 0xFFFFFF) * 128u + a2; a2 = (a2 & 0xFFFFFF) * 128u + a3; a3 = (a3 & 0xFFFFFF) * 128u + a0; }
            if (OP == 11) { f0 = (float)a0; f1 = (float)a1; f2 = (float)a2; f3 = (float)a3; a0 += __float_as_uint(f1); a1 += __float_as_uint(f2); a2 += __float_as_uint(f3); a3 += __float_as_uint(f0); }
            if (OP == 12) { a0 = min(a0, a1) + 3u; a1 = max(a1, a2) + 5u; a2 = min(a2, a3) + 7u; a3 = max(a3, a0) + 9u; }
            if (OP == 13) { a0 = a0 * a1; a1 = a1 * a2; a2 = a2 * a3; a3 = a3 * a0; }
            if (OP == 14) { a0 = (a0 & 255u) ^ a1; a1 = (a1 & 255u) ^ a2; a2 = (a2 & 255u) ^ a3; a3 = (a3 & 255u) ^ a0; }
            if (OP == 15) { a0 = __builtin_amdgcn_perm(a0, a1, 0x07060302u); a1 = __builtin_amdgcn_perm(a1, a2, 0x07060302u); a2 = __builtin_amdgcn_perm(a2, a3, 0x07060302u); a3 = __builtin_amdgcn_perm(a3, a0, 0x07060302u); }
            if (OP == 16) { a0 = __builtin_amdgcn_alignbit(a0, a1, 25); a1 = __builtin_amdgcn_alignbit(a1, a2, 25); a2 = __builtin_amdgcn_alignbit(a2, a3, 25); a3 = __builtin_amdgcn_alignbit(a3, a0, 25); }
            if (OP == 17) { asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(f0) : "v"(a0), "v"(f2));
                            asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(f1) : "v"(a1), "v"(f3));
                            asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(f2) : "v"(a2), "v"(f0));
                            asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(f3) : "v"(a3), "v"(f1)); }
            if (OP == 18) { asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f0) : "v"(a0)); asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f1) : "v"(a1));
                            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f2) : "v"(a2)); asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f3) : "v"(a3));
                            a0 += __float_as_uint(f1); a1 += __float_as_uint(f2); a2 += __float_as_uint(f3); a3 += __float_as_uint(f0); }
            if (OP == 19) { asm volatile("v_lshlrev_b32 %0, 7, %1" : "=v"(a0) : "v"(a1)); asm volatile("v_lshlrev_b32 %0, 7, %1" : "=v"(a1) : "v"(a2));
                            asm volatile("v_lshlrev_b32 %0, 7, %1" : "=v"(a2) : "v"(a3)); asm volatile("v_lshlrev_b32 %0, 7, %1" : "=v"(a3) : "v"(a0)); }
            if (OP == 20) { asm volatile("v_or_b32 %0, %1, %2" : "=v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_or_b32 %0, %1, %2" : "=v"(a1) : "v"(a2), "v"(a3));
                            asm volatile("v_or_b32 %0, %1, %2" : "=v"(a2) : "v"(a3), "v"(a0)); asm volatile("v_or_b32 %0, %1, %2" : "=v"(a3) : "v"(a0), "v"(a1)); }
            if (OP == 21) { asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f0) : "v"(a0)); asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(f1) : "v"(a1));
                            asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f2) : "v"(a2)); asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(f3) : "v"(a3)); }
            if (OP == 22) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double*)&f0) : "v"(*(double*)&f0), "v"(*(double*)&f2));
                            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double*)&f2) : "v"(*(double*)&f2), "v"(*(double*)&f0)); }
            if (OP == 6) { f0 = __builtin_fmaf(f0, f2, f1); f1 = __builtin_fmaf(f1, f3, f2); f2 = __builtin_fmaf(f2, f3, f0); f3 = __builtin_fmaf(f3, f0, f1); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ __float_as_uint(f0 + f1 + f2 + f3);
}
template <int OP> void run(const char* name, unsigned* d, int ops_per_iter)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, 2u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = 4.0 /*waves*/ * N_IT * 8.0 * ops_per_iter;
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
    unsigned* d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>("v_sad_u8", d, 4);
    run<1>("v_lshl_add_u32", d, 4);
    run<2>("v_mul_f32", d, 4);
    run<3>("v_add_f32", d, 4);
    run<4>("cvt_f32_ubyte + v_add_u32", d, 8);
    run<5>("v_add_u32", d, 4);
    run<6>("v_fma_f32", d, 4);
    run<7>("v_lshl/lshr VOP2 + v_add_u32", d, 8);
    run<8>("v_lshl_or_b32", d, 4);
    run<9>("v_bfe_u32 + v_add_u32", d, 8);
    run<10>("v_and + v_mad_u32_u24", d, 8);
    run<11>("v_cvt_f32_u32 + v_add_u32", d, 8);
    run<12>("v_min/max_u32 + v_add_u32", d, 8);
    run<13>("v_mul_lo_u32", d, 4);
    run<14>("v_and + v_xor (or v_and_or)", d, 8);
    run<15>("v_perm_b32", d, 4);
    run<16>("v_alignbit_b32", d, 4);
    run<17>("v_fma_mix_f32", d, 4);
    run<18>("v_cvt_f32_f16 + v_add_u32", d, 8);
    run<19>("v_lshlrev_b32 (asm)", d, 4);
    run<20>("v_or_b32 (asm)", d, 4);
    run<21>("v_cvt_f32_ubyteN (asm)", d, 4);
    run<22>("v_pk_mul_f32 (asm)", d, 2);
    return 0;
}
