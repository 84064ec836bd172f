// Host <-> device copy costs that shape the class-API (one frame per call) entry points: a 1080p frame (6.2 MB), the
// bounding rectangle of the calibration quad inside it (1013 x 917 px), the warped board (620 x 620 px) and the small
// result records, from / to pageable and pinned host memory, as 1-D and 2-D copies.  Prints microseconds per copy
// (copy + stream synchronise, median of 200).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_copy.hip -o tools/bin/ubench_copy && tools/bin/ubench_copy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                       \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F>
static double med(F f, int n = 200)
{
    std::vector<double> t;
    for (int i = 0; i < 10; i++) f();
    for (int i = 0; i < n; i++) {
        double a = now_us();
        f();
        t.push_back(now_us() - a);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int W = 1920, H = 1080, S = 620;
    const size_t frame = (size_t)W * H * 3, board = (size_t)S * S * 3;
    const int rx = 550, ry = 108, rw = 1013, rh = 917;
    unsigned char *dev, *pin, *pag;
    CK(hipMalloc(&dev, frame + 4096));
    CK(hipHostMalloc(&pin, frame, hipHostMallocDefault));
    pag = (unsigned char*)malloc(frame);
    memset(pag, 1, frame);
    memset(pin, 2, frame);
    struct {
        const char* name;
        unsigned char* h;
    } kinds[2] = {{"pageable", pag}, {"pinned", pin}};
    for (auto& k : kinds) {
        unsigned char* h = k.h;
        printf("%-9s H2D frame 1-D            %8.1f us\n", k.name, med([&] { CK(hipMemcpyAsync(dev, h, frame, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }));
        printf("%-9s H2D frame 2-D (tight)    %8.1f us\n", k.name, med([&] { CK(hipMemcpy2DAsync(dev, W * 3, h, W * 3, W * 3, H, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }));
        printf("%-9s H2D quad rows 1-D        %8.1f us\n", k.name, med([&] { CK(hipMemcpyAsync(dev + (size_t)ry * W * 3, h + (size_t)ry * W * 3, (size_t)rh * W * 3, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }));
        printf("%-9s H2D quad rect 2-D        %8.1f us\n", k.name, med([&] { CK(hipMemcpy2DAsync(dev + (size_t)ry * W * 3 + rx * 3, W * 3, h + (size_t)ry * W * 3 + rx * 3, W * 3, rw * 3, rh, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }));
        printf("%-9s H2D board 1-D            %8.1f us\n", k.name, med([&] { CK(hipMemcpyAsync(dev, h, board, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }));
        printf("%-9s H2D board 2-D strided    %8.1f us\n", k.name, med([&] { CK(hipMemcpy2DAsync(dev, S * 3, h, W * 3, S * 3, S, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }));
        printf("%-9s D2H frame 1-D            %8.1f us\n", k.name, med([&] { CK(hipMemcpyAsync(h, dev, frame, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("%-9s D2H board 1-D            %8.1f us\n", k.name, med([&] { CK(hipMemcpyAsync(h, dev, board, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("%-9s D2H 16 KB                %8.1f us\n", k.name, med([&] { CK(hipMemcpyAsync(h, dev, 16384, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("%-9s H2D 2 KB                 %8.1f us\n", k.name, med([&] { CK(hipMemcpyAsync(dev, h, 2048, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }));
    }
    // the sequences a class-API call issues: upload, (kernel), download, ONE synchronise
    {
        unsigned char* pag2 = (unsigned char*)malloc(frame);
        memset(pag2, 3, frame);
        unsigned char* pin2;
        CK(hipHostMalloc(&pin2, frame, hipHostMallocDefault));
        const size_t rows = (size_t)rh * W * 3;
        printf("seq pageable: H2D rows 1-D + D2H board 1-D, one sync        %8.1f us\n", med([&] { CK(hipMemcpyAsync(dev, pag, rows, hipMemcpyHostToDevice, st)); CK(hipMemcpyAsync(pag2, dev, board, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("seq pageable: H2D rows 1-D + D2H board 2-D tight, one sync  %8.1f us\n", med([&] { CK(hipMemcpyAsync(dev, pag, rows, hipMemcpyHostToDevice, st)); CK(hipMemcpy2DAsync(pag2, S * 3, dev, S * 3, S * 3, S, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("seq pageable: H2D rows, sync, D2H board 1-D, sync            %8.1f us\n", med([&] { CK(hipMemcpyAsync(dev, pag, rows, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); CK(hipMemcpyAsync(pag2, dev, board, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("seq pageable in, PINNED out: H2D rows + D2H board, one sync  %8.1f us\n", med([&] { CK(hipMemcpyAsync(dev, pag, rows, hipMemcpyHostToDevice, st)); CK(hipMemcpyAsync(pin2, dev, board, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("seq pinned: H2D rows + D2H board, one sync                   %8.1f us\n", med([&] { CK(hipMemcpyAsync(dev, pin, rows, hipMemcpyHostToDevice, st)); CK(hipMemcpyAsync(pin2, dev, board, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("seq pageable: D2H frame 2-D tight                            %8.1f us\n", med([&] { CK(hipMemcpy2DAsync(pag2, W * 3, dev, W * 3, W * 3, H, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("seq pageable: H2D frame + D2H frame (1-D), one sync          %8.1f us\n", med([&] { CK(hipMemcpyAsync(dev, pag, frame, hipMemcpyHostToDevice, st)); CK(hipMemcpyAsync(pag2, dev, frame, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("seq pinned: H2D frame + D2H frame (1-D), one sync            %8.1f us\n", med([&] { CK(hipMemcpyAsync(dev, pin, frame, hipMemcpyHostToDevice, st)); CK(hipMemcpyAsync(pin2, dev, frame, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
        printf("sync hipMemcpy pageable: H2D frame                           %8.1f us\n", med([&] { CK(hipMemcpy(dev, pag, frame, hipMemcpyHostToDevice)); }));
        printf("sync hipMemcpy pageable: D2H frame                           %8.1f us\n", med([&] { CK(hipMemcpy(pag2, dev, frame, hipMemcpyDeviceToHost)); }));
        // a fresh pageable buffer per copy (what a camera loop hands over): no pinning can be cached
        printf("fresh malloc per copy: H2D frame 1-D                         %8.1f us\n", med([&] { unsigned char* q = (unsigned char*)malloc(frame); memset(q, 1, 4096); CK(hipMemcpyAsync(dev, q, frame, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); free(q); }, 50));
        printf("fresh malloc per copy: D2H board 1-D                         %8.1f us\n", med([&] { unsigned char* q = (unsigned char*)malloc(board); CK(hipMemcpyAsync(q, dev, board, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); free(q); }, 50));
    }
    // pageable -> pinned staging on the host, then pinned DMA (what a caller-side staging buffer would cost)
    printf("host memcpy frame pageable -> pinned   %8.1f us\n", med([&] { memcpy(pin, pag, frame); }));
    printf("host memcpy board pageable -> pinned   %8.1f us\n", med([&] { memcpy(pin, pag, board); }));
    printf("empty stream synchronise               %8.1f us\n", med([&] { CK(hipStreamSynchronize(st)); }));
    return 0;
}
