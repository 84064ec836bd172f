// Is y = rcp(x); e = fma(-x, y, 1); y = fma(e, y, y) the correctly rounded 1.0f / x for EVERY float x in [1, 128)?
// (the bilateral divides by its weight sum, which lies in [1, 49]: the centre tap's weight is exactly 1 and there are 49
// taps of weight <= 1).  Exhaustive: 7 binades x 2^23 values against the IEEE division.
//   hipcc --offload-arch=gfx950 -O2 -fno-fast-math -ffp-contract=off tools/probe_rcp_exact.hip -o tools/bin/probe_rcp_exact
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
// the refined reciprocal of 2^24 consecutive floats, for the comparison with the HOST's division
__global__ void k_out(unsigned base, float* out)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(base + i);
    float y = __builtin_amdgcn_rcpf(x);
    const float e = __fmaf_rn(-x, y, 1.0f);
    out[i] = __fmaf_rn(e, y, y);
}
__global__ void k(unsigned base, unsigned long long* bad, unsigned* first_bad)
{
    const unsigned bits = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(bits);
    const float want = __fdiv_rn(1.0f, x);
    float y = __builtin_amdgcn_rcpf(x);
    const float e = __fmaf_rn(-x, y, 1.0f);
    y = __fmaf_rn(e, y, y);
    if (__float_as_uint(y) != __float_as_uint(want)) {
        atomicAdd(bad, 1ull);
        atomicMin(first_bad, bits);
    }
}
int main()
{
    unsigned long long* bad; unsigned* fb;
    hipMalloc(&bad, 8); hipMalloc(&fb, 4);
    hipMemset(bad, 0, 8); hipMemset(fb, 0xff, 4);
    const unsigned lo = 0x3F800000u /* 1.0 */, hi = 0x43000000u /* 128.0 */;
    for (unsigned b = lo; b < hi; b += 1u << 24) hipLaunchKernelGGL(k, dim3((1u << 24) / 256), dim3(256), 0, 0, b, bad, fb);
    unsigned long long hb; unsigned hf;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, fb, 4, hipMemcpyDeviceToHost);
    printf("rcp + one fma refinement against the DEVICE's 1.0f / x over every float in [1, 128): %llu of %u differ (first 0x%08x)\n", hb, hi - lo, hf);
    // and against the host's IEEE division (the device's own expansion of `/` is not taken on trust)
    float* dout; hipMalloc(&dout, (size_t)4 << 24);
    float* hout = (float*)malloc((size_t)4 << 24);
    unsigned long long hbad = 0;
    for (unsigned b = lo; b < hi; b += 1u << 24) {
        hipLaunchKernelGGL(k_out, dim3((1u << 24) / 256), dim3(256), 0, 0, b, dout);
        hipMemcpy(hout, dout, (size_t)4 << 24, hipMemcpyDeviceToHost);
        for (unsigned i = 0; i < (1u << 24); i++) {
            unsigned bits = b + i; float x; memcpy(&x, &bits, 4);
            volatile float w = 1.0f / x;
            float wv = w;
            if (memcmp(&wv, &hout[i], 4)) hbad++;
        }
    }
    printf("the same against the HOST's 1.0f / x: %llu differ\n", hbad);
    return 0;
}
