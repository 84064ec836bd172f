// Is v_mfma_f32_4x4x1_16B_f32 an exact fused multiply-add per element (one IEEE round-to-nearest-even rounding of
// a * b + c), i.e. bit-identical to fmaf?  Every lane supplies one A and one B value; block b = lanes 4b..4b+3;
// D[b][m][n] = A[4b + m] * B[4b + n] + C[b][m][n], lane 4b + n holds column n in its four result registers (row m).
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_exact.hip -o tools/bin/probe_mfma_exact && tools/bin/probe_mfma_exact
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, const float* c, float* d, int steps)
{
    const int lane = threadIdx.x;
    v4f acc;
    for (int m = 0; m < 4; m++) acc[m] = c[lane * 4 + m];
    for (int s = 0; s < steps; s++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s * 64 + lane], b[s * 64 + lane], acc, 0, 0, 0);
    for (int m = 0; m < 4; m++) d[lane * 4 + m] = acc[m];
}
int main()
{
    const int steps = 49, trials = 2000;
    float *ha = (float*)malloc(steps * 64 * 4), *hb = (float*)malloc(steps * 64 * 4), hc[256], hd[256];
    float *da, *db, *dc, *dd;
    hipMalloc(&da, steps * 64 * 4); hipMalloc(&db, steps * 64 * 4); hipMalloc(&dc, 1024); hipMalloc(&dd, 1024);
    srand(1);
    long bad = 0, total = 0;
    for (int t = 0; t < trials; t++) {
        for (int i = 0; i < steps * 64; i++) {
            // weights like the filter's (exp(-x) products down to 1e-23, some exactly 0) and byte-valued pixels (channel 3 = 1.0)
            double e = -(rand() % 5300) / 100.0;
            ha[i] = (rand() % 11 == 0) ? 0.f : (float)exp(e) * (1.f + (rand() % 1000) / 7919.f);
            hb[i] = (i % 4 == 3) ? 1.f : (float)(rand() % 256);
        }
        for (int i = 0; i < 256; i++) hc[i] = (t % 2) ? 0.f : (float)(rand() % 100000) / 3.f;
        hipMemcpy(da, ha, steps * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, steps * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd, steps);
        hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        for (int lane = 0; lane < 64; lane++)
            for (int m = 0; m < 4; m++) {
                const int blk = lane / 4, n = lane % 4;
                float r = hc[lane * 4 + m];
                for (int s = 0; s < steps; s++) r = fmaf(ha[s * 64 + blk * 4 + m], hb[s * 64 + blk * 4 + n], r);
                total++;
                if (memcmp(&r, &hd[lane * 4 + m], 4)) { if (bad < 5) printf("mismatch lane %d m %d: mfma %.9g fmaf %.9g\n", lane, m, hd[lane * 4 + m], r); bad++; }
            }
    }
    printf("v_mfma_f32_4x4x1f32 chains of %d against fmaf chains: %ld of %ld results differ\n", steps, bad, total);
    return 0;
}
