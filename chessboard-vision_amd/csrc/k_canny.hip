// cv2.Canny(gray, t1, t2) (aperture 3, L1 gradient) on a whole image, for SmartGridExtractor.refine_grid
// (grid_extractor.py:66-121; one shot at calibration).  Same algorithm as the Canny inside
// k_hough: Sobel 3x3 with replicated borders -> |dx| + |dy| -> non-maximum suppression along the quantised
// gradient direction (TG22 fixed point) -> hysteresis.  The image does not fit LDS, so hysteresis is tile
// local (64 x 64 tiles flooded to a fixed point in LDS) and repeated over the image until no tile changes.
#include "cbv_device.h"

#define CN_T 64 // tile edge of the hysteresis kernel

// gray (from BGR when cn == 3) -> magnitude (u16, padded by one zero pixel) + direction class (u8, tight)
__global__ __launch_bounds__(256) void k_canny_grad(const u8* __restrict__ src, int w, int h, int stride, int cn,
                                                     u16* __restrict__ mag, u8* __restrict__ dir)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    int v[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int yy = min(max(y + j - 1, 0), h - 1);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int xx = min(max(x + i - 1, 0), w - 1);
            const u8* p = src + (size_t)yy * stride + (size_t)xx * cn;
            v[j][i] = cn == 3 ? d_gray(p[0], p[1], p[2]) : p[0];
        }
    }
    const int dx = (v[0][2] - v[0][0]) + 2 * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0]);
    const int dy = (v[2][0] - v[0][0]) + 2 * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2]);
    const int ax = abs(dx), ay = abs(dy) << 15;
    const int tg22x = ax * 13573, tg67x = tg22x + (ax << 16);
    mag[(size_t)(y + 1) * (w + 2) + x + 1] = (u16)(ax + abs(dy));
    dir[(size_t)y * w + x] = (u8)(ay < tg22x ? 0 : (ay > tg67x ? 1 : (((dx ^ dy) < 0) ? 3 : 2)));
}

// map (padded by one "not an edge" pixel): 0 weak candidate, 1 not an edge, 2 edge
__global__ __launch_bounds__(256) void k_canny_nms(const u16* __restrict__ mag, const u8* __restrict__ dir, int w, int h,
                                                    int low, int high, u8* __restrict__ map)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int pw = w + 2;
    const u16* m = mag + (size_t)(y + 1) * pw + x + 1;
    const int v = m[0];
    u8 code = 1;
    if (v > low) {
        const int d = dir[(size_t)y * w + x];
        bool keep;
        if (d == 0) keep = v > m[-1] && v >= m[1];
        else if (d == 1) keep = v > m[-pw] && v >= m[pw];
        else {
            const int s = d == 3 ? -1 : 1;
            keep = v > m[-pw - s] && v > m[pw + s];
        }
        if (keep) code = v > high ? 2 : 0;
    }
    map[(size_t)(y + 1) * pw + x + 1] = code;
}

__global__ __launch_bounds__(256) void k_canny_hyst(u8* __restrict__ map, int w, int h, int* __restrict__ changed)
{
    __shared__ u8 t[(CN_T + 2) * (CN_T + 2)];
    const int pw = w + 2, x0 = blockIdx.x * CN_T, y0 = blockIdx.y * CN_T;
    const int tw = min(CN_T, w - x0), th = min(CN_T, h - y0), lw = tw + 2;
    for (int i = threadIdx.x; i < (th + 2) * lw; i += 256) {
        const int r = i / lw, c = i - r * lw;
        t[r * (CN_T + 2) + c] = map[(size_t)(y0 + r) * pw + x0 + c];
    }
    __syncthreads();
    int any = 0;
    for (;;) {
        int ch = 0;
        for (int i = threadIdx.x; i < th * tw; i += 256) {
            const int r = i / tw, c = i - r * tw;
            u8* p = &t[(r + 1) * (CN_T + 2) + c + 1];
            if (*p != 0) continue;
            const int L = CN_T + 2;
            if ((p[-L - 1] | p[-L] | p[-L + 1] | p[-1] | p[1] | p[L - 1] | p[L] | p[L + 1]) & 2) {
                *p = 2;
                ch = 1;
            }
        }
        if (!__syncthreads_or(ch)) break;
        any = 1;
    }
    if (!any) return;
    for (int i = threadIdx.x; i < th * tw; i += 256) {
        const int r = i / tw, c = i - r * tw;
        map[(size_t)(y0 + r + 1) * pw + x0 + c + 1] = t[(r + 1) * (CN_T + 2) + c + 1];
    }
    if (threadIdx.x == 0) atomicOr(changed, 1);
}

__global__ __launch_bounds__(256) void k_canny_out(const u8* __restrict__ map, int w, int h, u8* __restrict__ edges)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    edges[(size_t)y * w + x] = map[(size_t)(y + 1) * (w + 2) + x + 1] == 2 ? 255 : 0;
}

// src: device image (tight rows of w * cn bytes at `stride`); edges: device, tight w x h.  Synchronises.
int launch_canny(cbv_ctx* ctx, const u8* src, int w, int h, int stride, int cn, int low, int high, u8* edges, DevBuf* scratch)
{
    const size_t pw = (size_t)w + 2, ph = (size_t)h + 2;
    const size_t mag_b = (pw * ph * 2 + 255) & ~(size_t)255, dir_b = ((size_t)w * h + 255) & ~(size_t)255;
    const size_t map_b = (pw * ph + 255) & ~(size_t)255;
    if (int rc = dev_ensure(ctx, scratch, mag_b + dir_b + map_b + 256)) return rc;
    u8* base = (u8*)scratch->p;
    u16* mag = (u16*)base;
    u8* dir = base + mag_b;
    u8* map = dir + dir_b;
    int* flag = (int*)(map + map_b);
    CBV_HIP(ctx, hipMemsetAsync(mag, 0, mag_b, ctx->stream));
    CBV_HIP(ctx, hipMemsetAsync(map, 1, map_b, ctx->stream));
    const dim3 grid((w + 63) / 64, (h + 3) / 4), blk(256);
    hipLaunchKernelGGL(k_canny_grad, grid, blk, 0, ctx->stream, src, w, h, stride, cn, mag, dir);
    hipLaunchKernelGGL(k_canny_nms, grid, blk, 0, ctx->stream, (const u16*)mag, (const u8*)dir, w, h, low, high, map);
    const dim3 tgrid((w + CN_T - 1) / CN_T, (h + CN_T - 1) / CN_T);
    for (int it = 0; it < 4096; it++) { // a chain crosses at most tiles-many tile borders; bounded anyway
        int host_flag = 0;
        CBV_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_canny_hyst, tgrid, blk, 0, ctx->stream, map, w, h, flag);
        CBV_HIP(ctx, hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (!host_flag) break;
    }
    hipLaunchKernelGGL(k_canny_out, grid, blk, 0, ctx->stream, (const u8*)map, w, h, edges);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// find_chessboard_corners' pixel stages (board_detection.py:9-14): BGR2GRAY + GaussianBlur((k, k), sigma) in
// OpenCV's 8.8 fixed point (REFLECT_101), and cv2.dilate with a (2r+1)^2 rectangle (pixels outside the image do
// not take part; three 5x5 dilations = one 13x13).  32 x 32 output tiles, separable through LDS.
// ---------------------------------------------------------------------------
#define GB_T 32
#define GB_RMAX 7

__global__ __launch_bounds__(256) void k_gray_gauss(const u8* __restrict__ src, int w, int h, int stride, int cn,
                                                     const int* __restrict__ coef, int k, u8* __restrict__ dst)
{
    __shared__ u8 g[(GB_T + 2 * GB_RMAX) * (GB_T + 2 * GB_RMAX)];
    __shared__ u16 hb[(GB_T + 2 * GB_RMAX) * GB_T];
    __shared__ int cf[2 * GB_RMAX + 1];
    const int r = k >> 1, L = GB_T + 2 * r;
    const int x0 = blockIdx.x * GB_T, y0 = blockIdx.y * GB_T;
    if (threadIdx.x < k) cf[threadIdx.x] = coef[threadIdx.x];
    for (int i = threadIdx.x; i < L * L; i += 256) {
        const int ly = i / L, lx = i - ly * L;
        const int sy = d_reflect101(y0 - r + ly, h), sx = d_reflect101(x0 - r + lx, w);
        const u8* p = src + (size_t)sy * stride + (size_t)sx * cn;
        g[ly * L + lx] = (u8)(cn == 3 ? d_gray(p[0], p[1], p[2]) : p[0]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * GB_T; i += 256) {
        const int ly = i / GB_T, lx = i - ly * GB_T;
        int acc = 0;
        for (int t = 0; t < k; t++) acc += cf[t] * g[ly * L + lx + t];
        hb[ly * GB_T + lx] = (u16)acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < GB_T * GB_T; i += 256) {
        const int ly = i / GB_T, lx = i - ly * GB_T;
        if (x0 + lx >= w || y0 + ly >= h) continue;
        u32 acc = 0;
        for (int t = 0; t < k; t++) acc += (u32)cf[t] * hb[(ly + t) * GB_T + lx];
        dst[(size_t)(y0 + ly) * w + x0 + lx] = (u8)((acc + (1u << 15)) >> 16);
    }
}

__global__ __launch_bounds__(256) void k_dilate_rect(const u8* __restrict__ src, int w, int h, int r, u8* __restrict__ dst)
{
    __shared__ u8 g[(GB_T + 2 * GB_RMAX) * (GB_T + 2 * GB_RMAX)];
    __shared__ u8 hb[(GB_T + 2 * GB_RMAX) * GB_T];
    const int L = GB_T + 2 * r;
    const int x0 = blockIdx.x * GB_T, y0 = blockIdx.y * GB_T;
    for (int i = threadIdx.x; i < L * L; i += 256) {
        const int ly = i / L, lx = i - ly * L;
        const int sy = y0 - r + ly, sx = x0 - r + lx;
        g[ly * L + lx] = (sx >= 0 && sx < w && sy >= 0 && sy < h) ? src[(size_t)sy * w + sx] : 0; // outside never wins a max
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * GB_T; i += 256) {
        const int ly = i / GB_T, lx = i - ly * GB_T;
        int m = 0;
        for (int t = 0; t <= 2 * r; t++) m = max(m, (int)g[ly * L + lx + t]);
        hb[ly * GB_T + lx] = (u8)m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < GB_T * GB_T; i += 256) {
        const int ly = i / GB_T, lx = i - ly * GB_T;
        if (x0 + lx >= w || y0 + ly >= h) continue;
        int m = 0;
        for (int t = 0; t <= 2 * r; t++) m = max(m, (int)hb[(ly + t) * GB_T + lx]);
        dst[(size_t)(y0 + ly) * w + x0 + lx] = (u8)m;
    }
}

int launch_gray_gauss(cbv_ctx* ctx, const u8* src, int w, int h, int stride, int cn, const int* coef_dev, int k, u8* dst)
{
    if (k < 1 || k > 2 * GB_RMAX + 1 || !(k & 1)) return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "Gaussian kernel size %d is not supported (odd, <= 15)", k);
    hipLaunchKernelGGL(k_gray_gauss, dim3((w + GB_T - 1) / GB_T, (h + GB_T - 1) / GB_T), dim3(256), 0, ctx->stream, src, w, h, stride, cn,
                       coef_dev, k, dst);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

int launch_dilate_rect(cbv_ctx* ctx, const u8* src, int w, int h, int r, u8* dst)
{
    if (r < 0 || r > GB_RMAX) return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "dilation radius %d is not supported (<= 7)", r);
    hipLaunchKernelGGL(k_dilate_rect, dim3((w + GB_T - 1) / GB_T, (h + GB_T - 1) / GB_T), dim3(256), 0, ctx->stream, src, w, h, r, dst);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}
