// Issue cost of single vector instructions on gfx950: 16 independent chains per lane, 4 waves per SIMD on every CU
// (256 workgroups x 1024 lanes), N_IT iterations; prints clocks per wave64 instruction per SIMD at 2.4 GHz.
// The numbers feed the instruction-mix models in DESIGN.md / bench.py (profiles/r02/issue_costs.txt).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_ops.hip -o tools/bin/ubench_ops && tools/bin/ubench_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 2048

#define OPS(X)                                                                                     \
    X(0, "v_fma_f32", "v_fma_f32 %0, %2, %3, %0", F)                                               \
    X(1, "v_add_f32", "v_add_f32 %0, %0, %2", F)                                                   \
    X(2, "v_mul_f32", "v_mul_f32 %0, %0, %2", F)                                                   \
    X(3, "v_fma_mix_f32 (f16 src0)", "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]", FU)        \
    X(4, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %1", FU)                                              \
    X(5, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %1", FU)                                        \
    X(6, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %1", FU)                                              \
    X(7, "v_cvt_u32_f32", "v_cvt_u32_f32 %1, %0", UF)                                              \
    X(8, "v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 %1, %0, 1, %1", UF)                                   \
    X(9, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %1, %0, %2", UF)                              \
    X(10, "v_rcp_f32", "v_rcp_f32 %0, %0", F)                                                      \
    X(11, "v_add_u32", "v_add_u32 %1, %1, %4", U)                                                  \
    X(12, "v_lshlrev_b32", "v_lshlrev_b32 %1, 2, %1", U)                                           \
    X(13, "v_lshrrev_b32", "v_lshrrev_b32 %1, 8, %1", U)                                           \
    X(14, "v_and_b32", "v_and_b32 %1, %1, %4", U)                                                  \
    X(15, "v_bfe_u32", "v_bfe_u32 %1, %1, 8, 8", U)                                                \
    X(16, "v_sad_u8 (vgpr acc)", "v_sad_u8 %1, %1, %4, %5", U)                                     \
    X(17, "v_sad_u8 (sgpr acc)", "v_sad_u8 %1, %1, %4, %6", U)                                     \
    X(18, "v_alignbit_b32", "v_alignbit_b32 %1, %1, %4, 25", U)                                    \
    X(19, "v_perm_b32", "v_perm_b32 %1, %1, %4, %5", U)                                            \
    X(20, "v_lshl_add_u32", "v_lshl_add_u32 %1, %1, 2, %4", U)                                     \
    X(21, "v_lshl_or_b32", "v_lshl_or_b32 %1, %1, 8, %4", U)                                       \
    X(22, "v_and_or_b32", "v_and_or_b32 %1, %1, %4, %5", U)                                        \
    X(23, "v_add3_u32", "v_add3_u32 %1, %1, %4, %5", U)                                            \
    X(24, "v_bfi_b32", "v_bfi_b32 %1, %1, %4, %5", U)                                              \
    X(25, "v_mul_u32_u24", "v_mul_u32_u24 %1, %1, %4", U)                                          \
    X(26, "v_mad_u32_u24", "v_mad_u32_u24 %1, %1, %4, %5", U)                                      \
    X(27, "v_mad_i32_i24", "v_mad_i32_i24 %1, %1, %4, %5", U)                                      \
    X(28, "v_mul_lo_u32", "v_mul_lo_u32 %1, %1, %4", U)                                            \
    X(29, "v_mul_hi_u32", "v_mul_hi_u32 %1, %1, %4", U)                                            \
    X(30, "v_min_u32", "v_min_u32 %1, %1, %4", U)                                                  \
    X(31, "v_med3_i32", "v_med3_i32 %1, %1, %4, %5", U)                                            \
    X(32, "v_max3_u32", "v_max3_u32 %1, %1, %4, %5", U)                                            \
    X(33, "v_cndmask_b32 (vcc)", "v_cndmask_b32 %1, %1, %4, vcc", U)                               \
    X(34, "v_mov_b32", "v_mov_b32 %1, %4", U)                                                      \
    X(35, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf", U) \
    X(36, "v_add_u32 sdwa byte", "v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", U) \
    X(37, "v_pk_add_u16", "v_pk_add_u16 %1, %1, %4", U)                                            \
    X(38, "v_pk_mad_i16", "v_pk_mad_i16 %1, %1, %4, %5", U)                                        \
    X(39, "v_pk_max_i16", "v_pk_max_i16 %1, %1, %4", U)                                            \
    X(40, "v_pk_min_u16", "v_pk_min_u16 %1, %1, %4", U)                                            \
    X(41, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %1, %1, %4", U)                                      \
    X(42, "v_pk_lshrrev_b16", "v_pk_lshrrev_b16 %1, 4, %1", U)                                     \
    X(43, "v_dot4_u32_u8", "v_dot4_u32_u8 %1, %1, %4, %5", U)                                      \
    X(44, "v_dot2_f32_f16", "v_dot2_f32_f16 %0, %1, %4, %0", FU)                                   \
    X(45, "v_pk_fma_f16", "v_pk_fma_f16 %1, %1, %4, %5", U)                                        \
    X(46, "v_fma_f32 (sgpr src)", "v_fma_f32 %0, %0, %7, %3", F)                                   \
    X(47, "v_add_f32 (2 x 8 chains)", "v_add_f32 %0, %0, %2", F)

enum { F, U, FU, UF };

template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, float seed, unsigned useed)
{
    float acc[16];
    unsigned ua[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        acc[i] = seed + threadIdx.x * 0.001f + i;
        ua[i] = (threadIdx.x * 2654435761u + i * 40503u) ^ useed;
    }
    const float w = 1.0001f, x = 0.5f;
    const unsigned sacc = __builtin_amdgcn_readfirstlane(useed & 1023u);
    const float sf = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed)));
    for (int it = 0; it < N_IT; it++) {
#define X(id, name, text, kind)                                                                                      \
    if (OP == id) {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 16; i++)                                                               \
            asm volatile(text : "+v"(acc[i]), "+v"(ua[i]) : "v"(w), "v"(x), "v"(ua[(i + 5) & 15]), "v"(ua[(i + 9) & 15]), "s"(sacc), "s"(sf)); \
    }
        OPS(X)
#undef X
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

// packed f32 (64-bit register pairs): 8 independent pair chains per lane
typedef float f2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(1024) void k_pk(float* out, float seed)
{
    f2v acc[8], w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        acc[i] = f2v{seed + threadIdx.x * 0.001f + i, seed - i};
        w[i] = f2v{1.0001f + i * 1e-6f, 0.9999f};
    }
    f2v x = {0.5f, 0.25f};
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w[i]), "v"(x));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(w[(i + 3) & 7]), "v"(x));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(w[i]));
                if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(w[i]));
                if (OP == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(w[(i + 3) & 7]), "v"(x));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS read throughput: every lane reads its own bank (conflict-free), 8 reads in flight
template <int BYTES>
__global__ __launch_bounds__(1024) void k_lds(float* out)
{
    __shared__ __attribute__((aligned(16))) unsigned buf[16384];
    for (int i = threadIdx.x; i < 16384; i += 1024) buf[i] = i;
    __syncthreads();
    unsigned a = 0;
    unsigned addr = (threadIdx.x & 63) * BYTES;
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (BYTES == 4) {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(q * 256));
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                a += v;
            } else if (BYTES == 1) {
                unsigned v;
                asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr * 4), "n"(q * 256));
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                a += v;
            } else {
                uint4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(q * 1024));
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                a += v.x + v.w;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)a;
}

template <typename Fn> float time_ms(Fn fn)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    fn();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 3; r++) fn();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

int main()
{
    float* d;
    (void)hipMalloc(&d, 256 * 1024 * 4);
    printf("%-30s %9s %9s\n", "instruction", "ms", "clk/wave-instr/SIMD @2.4GHz (4 waves/SIMD)");
#define X(id, name, text, kind)                                                                           \
    {                                                                                                     \
        const float ms = time_ms([&] { hipLaunchKernelGGL(k<id>, dim3(256), dim3(1024), 0, 0, d, 1.f, 77u); }); \
        printf("%-30s %9.3f %9.2f\n", name, ms, ms * 1e6 / (4.0 * N_IT * 16) * 2.4);                      \
    }
    OPS(X)
#undef X
#define PK(id, name)                                                                                         \
    {                                                                                                         \
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_pk<id>, dim3(256), dim3(1024), 0, 0, d, 1.f); }); \
        printf("%-30s %9.3f %9.2f  (two results per instruction)\n", name, ms, ms * 1e6 / (4.0 * N_IT * 16) * 2.4); \
    }
    PK(0, "v_pk_fma_f32")
    PK(1, "v_pk_fma_f32 op_sel hi,hi")
    PK(4, "v_pk_fma_f32 op_sel lo,lo")
    PK(2, "v_pk_add_f32")
    PK(3, "v_pk_mul_f32")
    {
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_lds<4>, dim3(256), dim3(1024), 0, 0, d); });
        printf("%-30s %9.3f %9.2f  (clk per wave-instr per CU: 16 waves share one LDS)\n", "ds_read_b32 conflict-free", ms, ms * 1e6 / (16.0 * N_IT * 8) * 2.4);
    }
    {
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_lds<1>, dim3(256), dim3(1024), 0, 0, d); });
        printf("%-30s %9.3f %9.2f  (per CU)\n", "ds_read_u8 conflict-free", ms, ms * 1e6 / (16.0 * N_IT * 8) * 2.4);
    }
    {
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_lds<16>, dim3(256), dim3(1024), 0, 0, d); });
        printf("%-30s %9.3f %9.2f  (per CU)\n", "ds_read_b128", ms, ms * 1e6 / (16.0 * N_IT * 8) * 2.4);
    }
    return 0;
}
