// Micro-benchmark: the bilateral tap body (sad, alignbit, 4 mul, 4 add per tap-output) as a pure VALU stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 512
template <int WAVES_X64>
__global__ __launch_bounds__(WAVES_X64 * 64) void k(unsigned* out, unsigned seed, float spw)
{
    unsigned p[12], ctr[4];
    float fb[12], fg[12], fr[12];
    float sb[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0}, sr[4] = {0, 0, 0, 0}, sw[4] = {0, 0, 0, 0};
    unsigned h = threadIdx.x * 2654435761u + seed;
    for (int j = 0; j < 12; j++) { h = h * 1664525u + 1013904223u; p[j] = h & 0xFFFFFF; fb[j] = (float)(p[j] & 255); fg[j] = (float)((p[j] >> 8) & 255); fr[j] = (float)((p[j] >> 16) & 255); }
    for (int o = 0; o < 4; o++) ctr[o] = p[4 + o];
    const unsigned lane_hi = (threadIdx.x & 31) << 27;
    for (int i = 0; i < N_IT; i++) {
#pragma unroll
        for (int dx = -4; dx <= 4; dx++) {
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int j = o + 4 + dx;
                const unsigned sad = __builtin_amdgcn_sad_u8(p[j], ctr[o], 0u);
                const unsigned addr = __builtin_amdgcn_alignbit(sad, lane_hi, 25);
                const float wgt = spw * __uint_as_float((addr & 0xFFFFu) | 0x3f000000u);
                const float tb = fb[j] * wgt, tg = fg[j] * wgt, tr = fr[j] * wgt;
                sb[o] = sb[o] + tb; sg[o] = sg[o] + tg; sr[o] = sr[o] + tr; sw[o] = sw[o] + wgt;
            }
        }
        ctr[i & 3] += 0x010101u; // keep the loop from being hoisted
    }
    float r = 0;
    for (int o = 0; o < 4; o++) r += sb[o] + sg[o] + sr[o] + sw[o];
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(r);
}
template <int W> void run(unsigned* d, int blocks)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(W * 64), 0, 0, d, 1u, 0.99f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(W * 64), 0, 0, d, 2u, 0.98f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double waves_per_simd = (double)blocks * W / 1024.0;
    double instr = waves_per_simd * N_IT * 36.0 * 12.0; // 12 VALU per tap-output (sad, alignbit, and_or, 4 mul, 4 add + 1)
    printf("%2d waves/WG x %4d WGs (%.1f waves/SIMD): %8.3f ms, %.2f cycles per VALU instr at 2.4 GHz, %.1f cycles per tap-output\n", W, blocks,
           waves_per_simd, ms, ms * 1e6 * 2.4 / instr, ms * 1e6 * 2.4 / (waves_per_simd * N_IT * 36.0));
}
int main()
{
    unsigned* d; (void)hipMalloc(&d, 2048 * 1024 * 4);
    run<16>(d, 256);
    run<4>(d, 1024);
    run<4>(d, 2048);
    run<8>(d, 1024);
    run<4>(d, 512);
    run<4>(d, 256);
    return 0;
}
