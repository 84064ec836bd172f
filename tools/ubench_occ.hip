// Issue cost of a vector instruction against the number of waves per SIMD (1..8) on gfx950, and the shader clock the
// chip really runs at under that load (clock64 = s_memtime ticks against wall_clock64 = 100 MHz).
// 16 independent chains per lane; every CU gets the same number of waves.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_occ.hip -o tools/bin/ubench_occ && tools/bin/ubench_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 4096

template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* clk, float seed, unsigned useed)
{
    float acc[16];
    unsigned ua[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        acc[i] = seed + threadIdx.x * 0.001f + i;
        ua[i] = (threadIdx.x * 2654435761u + i * 40503u) ^ useed;
    }
    const float w = 1.0001f, x = 0.5f;
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %2, %3, %0" : "+v"(acc[i]), "+v"(ua[i]) : "v"(w), "v"(x), "v"(ua[(i + 5) & 15]));
            if (OP == 1) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "+v"(acc[i]), "+v"(ua[i]) : "v"(w), "v"(x), "v"(ua[(i + 5) & 15]));
            if (OP == 2) asm volatile("v_sad_u8 %1, %1, %4, %4" : "+v"(acc[i]), "+v"(ua[i]) : "v"(w), "v"(x), "v"(ua[(i + 5) & 15]));
            if (OP == 3) asm volatile("v_lshlrev_b32 %1, 2, %1" : "+v"(acc[i]), "+v"(ua[i]) : "v"(w), "v"(x), "v"(ua[(i + 5) & 15]));
            if (OP == 4) asm volatile("v_add_u32 %1, %1, %4" : "+v"(acc[i]), "+v"(ua[i]) : "v"(w), "v"(x), "v"(ua[(i + 5) & 15]));
            if (OP == 5) asm volatile("v_add_f32 %0, %0, %2" : "+v"(acc[i]), "+v"(ua[i]) : "v"(w), "v"(x), "v"(ua[(i + 5) & 15]));
        }
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
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

typedef float f2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k_pk(float* out, unsigned long long* clk, float seed)
{
    f2v acc[8], w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        acc[i] = f2v{seed + threadIdx.x * 0.001f + i, seed - i};
        w[i] = f2v{1.0001f + i * 1e-6f, 0.9999f};
    }
    f2v x = {0.5f, 0.25f};
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w[i]), "v"(x));
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
    unsigned long long* c;
    (void)hipMalloc(&d, 512 * 1024 * 4);
    (void)hipMalloc(&c, 16);
    const char* names[7] = {"v_fma_f32", "v_cvt_f32_ubyte1", "v_sad_u8", "v_lshlrev_b32", "v_add_u32", "v_add_f32", "v_pk_fma_f32"};
    const int Ws[6] = {1, 2, 3, 4, 6, 8};
    printf("%-18s %5s %9s %22s %12s %14s\n", "instruction", "waves", "ms", "ns/wave-instr/SIMD", "s_memtime/us", "clk@2.4GHz");
    for (int op = 0; op < 7; op++)
        for (int wi = 0; wi < 6; wi++) {
            const int W = Ws[wi];
            // W waves per SIMD on every CU: W <= 4 -> one workgroup of 256 W lanes per CU; 6 / 8 -> two of 768 / 1024
            const int per_cu = W <= 4 ? 1 : 2, threads = 256 * (W / per_cu);
            auto fn = [&] {
                switch (op) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256 * per_cu), dim3(threads), 0, 0, d, c, 1.f, 77u); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256 * per_cu), dim3(threads), 0, 0, d, c, 1.f, 77u); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256 * per_cu), dim3(threads), 0, 0, d, c, 1.f, 77u); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256 * per_cu), dim3(threads), 0, 0, d, c, 1.f, 77u); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256 * per_cu), dim3(threads), 0, 0, d, c, 1.f, 77u); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(256 * per_cu), dim3(threads), 0, 0, d, c, 1.f, 77u); break;
                default: hipLaunchKernelGGL(k_pk, dim3(256 * per_cu), dim3(threads), 0, 0, d, c, 1.f); break;
                }
            };
            const float ms = time_ms(fn);
            unsigned long long h[2];
            (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
            const double ns = ms * 1e6 / ((double)W * N_IT * 16);
            printf("%-18s %5d %9.3f %22.3f %12.1f %14.2f\n", names[op], W, ms, ns, h[1] ? (double)h[0] / (h[1] / 100.0) : 0.0, ns * 2.4);
        }
    return 0;
}
