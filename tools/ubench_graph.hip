// Does a hipGraph close the gaps of a class-API call?  upload (pageable host -> device), a small kernel, download
// (device -> pageable host), as three stream calls + one synchronise against ONE graph launch + synchronise, with the
// host pointers of the copy nodes re-set before every launch (hipGraphExecMemcpyNodeSetParams1D), 1080p frame in / board out.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_graph.hip -o tools/bin/ubench_graph && tools/bin/ubench_graph
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double med(F f, int n = 100) { std::vector<double> t; for (int i = 0; i < 10; i++) f(); for (int i = 0; i < n; i++) { double a = now_us(); f(); t.push_back(now_us() - a); } std::sort(t.begin(), t.end()); return t[t.size() / 2]; }
__global__ void k_touch(const unsigned char* in, unsigned char* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = in[i] + 1; }
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t frame = 5324800, board = 1153200;
    unsigned char *din, *dout; CK(hipMalloc(&din, 8 << 20)); CK(hipMalloc(&dout, 2 << 20));
    unsigned char* hin[2] = {(unsigned char*)malloc(frame), (unsigned char*)malloc(frame)};
    unsigned char* hout[2] = {(unsigned char*)malloc(board), (unsigned char*)malloc(board)};
    memset(hin[0], 1, frame); memset(hin[1], 2, frame); memset(hout[0], 0, board); memset(hout[1], 0, board);
    int flip = 0;
    double t_stream = med([&] {
        flip ^= 1;
        CK(hipMemcpyAsync(din, hin[flip], frame, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_touch, dim3((board + 255) / 256), dim3(256), 0, st, din, dout, (int)board);
        CK(hipMemcpyAsync(hout[flip], dout, board, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
    });
    printf("stream: H2D 5.3 MB + kernel + D2H 1.15 MB + sync      %8.1f us   (out[0] = %d)\n", t_stream, hout[flip][0]);
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    hipGraphNode_t n_in, n_k, n_out;
    CK(hipGraphAddMemcpyNode1D(&n_in, g, nullptr, 0, din, hin[0], frame, hipMemcpyHostToDevice));
    hipKernelNodeParams kp; memset(&kp, 0, sizeof(kp));
    int nb = (int)board; void* args[3] = {&din, &dout, &nb};
    kp.func = (void*)k_touch; kp.gridDim = dim3((board + 255) / 256); kp.blockDim = dim3(256); kp.kernelParams = args;
    CK(hipGraphAddKernelNode(&n_k, g, &n_in, 1, &kp));
    CK(hipGraphAddMemcpyNode1D(&n_out, g, &n_k, 1, hout[0], dout, board, hipMemcpyDeviceToHost));
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double t_graph = med([&] {
        flip ^= 1;
        CK(hipGraphExecMemcpyNodeSetParams1D(ge, n_in, din, hin[flip], frame, hipMemcpyHostToDevice));
        CK(hipGraphExecMemcpyNodeSetParams1D(ge, n_out, hout[flip], dout, board, hipMemcpyDeviceToHost));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
    });
    printf("graph:  the same three nodes, pointers re-set per launch  %8.1f us   (out[0] = %d, expect %d)\n", t_graph, hout[flip][0], hin[flip][0] + 1);
    return 0;
}
