#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, unsigned* out, int n) { int i = threadIdx.x; if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u); }
int main() {
    float h[] = {0.5f, 1.5f, 2.5f, 3.5f, 254.5f, 255.5f, -0.5f, 0.49999997f, 0.50000006f, 300.f, -3.f, 127.5f, 128.5f, 1.4999999f, 2.5000002f, 254.49998f};
    int n = sizeof(h) / 4; float* d; unsigned* o; hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice); hipLaunchKernelGGL(k, 1, 64, 0, 0, d, o, n);
    unsigned r[32]; hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%.8g -> %u\n", h[i], r[i]);
    return 0;
}
