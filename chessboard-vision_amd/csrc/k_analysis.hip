// ImageEnhancer.prepare_analysis (frame_enhancer.py:148-159):
//   BGR2GRAY -> GaussianBlur((5,5),0) -> Otsu threshold -> THRESH_BINARY.
// k_gray_blur_hist: one pass over the BGR frame producing the (unblurred) gray
// image the method returns, the blurred image and its 256-bin histogram;
// k_otsu: the double-precision between-class-variance sweep;
// k_threshold: blurred > t ? 255 : 0.
#include "cbv_device.h"

#define AN_TW 64
#define AN_TH 16

__global__ __launch_bounds__(256) void k_gray_blur_hist(const u8* __restrict__ src, u8* __restrict__ gray,
                                                         u8* __restrict__ blur, u32* __restrict__ aux,
                                                         int tiles_total, Geom g, int tiles_xn, int tiles_n)
{
    __shared__ u8 gt[(AN_TH + 4) * (AN_TW + 4)];
    __shared__ u16 hb[(AN_TH + 4) * AN_TW];
    __shared__ u32 hist[256 * 4];
    const int tid = xcd_remap(blockIdx.x, tiles_n);
    const int tyi = tid / tiles_xn, txi = tid - tyi * tiles_xn;
    const int x0 = txi * AN_TW, y0 = tyi * AN_TH;
    const size_t fo = (size_t)blockIdx.z * g.frame_stride;
    const size_t go = (size_t)blockIdx.z * g.w * g.h;
    const u8* sf = src + fo;
    for (int i = threadIdx.x; i < 256 * 4; i += blockDim.x) hist[i] = 0;
    for (int i = threadIdx.x; i < (AN_TH + 4) * (AN_TW + 4); i += blockDim.x) {
        const int r = i / (AN_TW + 4), c = i - r * (AN_TW + 4);
        const int yy = y0 - 2 + r, xx = x0 - 2 + c;
        const int sy = d_reflect101(yy, g.h), sx = d_reflect101(xx, g.w);
        const u8* p = sf + (size_t)sy * g.stride + (size_t)sx * 3;
        const int gv = d_gray(p[0], p[1], p[2]);
        gt[i] = (u8)gv;
        if (r >= 2 && r < AN_TH + 2 && c >= 2 && c < AN_TW + 2 && yy < g.h && xx < g.w)
            gray[go + (size_t)yy * g.w + xx] = (u8)gv;
    }
    __syncthreads();
    // horizontal 1-4-6-4-1 in 8.8 fixed point (coefficients 16,64,96,64,16)
    for (int i = threadIdx.x; i < (AN_TH + 4) * AN_TW; i += blockDim.x) {
        const int r = i / AN_TW, c = i - r * AN_TW;
        const u8* p = &gt[r * (AN_TW + 4) + c];
        hb[i] = (u16)(16 * p[0] + 64 * p[1] + 96 * p[2] + 64 * p[3] + 16 * p[4]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < AN_TH * AN_TW; i += blockDim.x) {
        const int r = i / AN_TW, c = i - r * AN_TW;
        const int yy = y0 + r, xx = x0 + c;
        if (yy < g.h && xx < g.w) {
            const u16* p = &hb[r * AN_TW + c];
            u32 acc = 16u * p[0] + 64u * p[AN_TW] + 96u * p[2 * AN_TW] + 64u * p[3 * AN_TW] + 16u * p[4 * AN_TW];
            u32 v = (acc + (1u << 15)) >> 16;
            blur[go + (size_t)yy * g.w + xx] = (u8)v;
            atomicAdd(&hist[v * 4 + (threadIdx.x & 3)], 1u);
        }
    }
    __syncthreads();
    u32* oh = aux + (size_t)blockIdx.z * aux_words(tiles_total) + (size_t)tiles_total * 256 + 2;
    {
        const int bin = threadIdx.x;
        u32 s = hist[bin * 4] + hist[bin * 4 + 1] + hist[bin * 4 + 2] + hist[bin * 4 + 3];
        if (s) atomicAdd(&oh[bin], s);
    }
}

int launch_gray_blur_hist(cbv_ctx* ctx, const u8* src, u8* gray, u8* blur, u32* aux, int tiles, Geom g, int batch)
{
    int txn = (g.w + AN_TW - 1) / AN_TW, tyn = (g.h + AN_TH - 1) / AN_TH;
    prof_begin(ctx, CBV_K_GRAY_BLUR);
    hipLaunchKernelGGL(k_gray_blur_hist, dim3(txn * tyn, 1, batch), dim3(256), 0, ctx->stream, src, gray, blur, aux,
                       tiles, g, txn, txn * tyn);
    prof_end(ctx, CBV_K_GRAY_BLUR);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// getThreshVal_Otsu_8u: 256 dependent double steps; one lane per frame.
__global__ void k_otsu(u32* __restrict__ aux, int tiles_total, int total)
{
    u32* base = aux + (size_t)blockIdx.x * aux_words(tiles_total) + (size_t)tiles_total * 256 + 2;
    if (threadIdx.x != 0) return;
    const u32* h = base;
    double mu = 0, scale = 1. / (double)total;
    for (int i = 0; i < 256; i++) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; i++) {
        double p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        double q2 = 1. - q1;
        if (fmin(q1, q2) < 1.1920928955078125e-07 || fmax(q1, q2) > 1. - 1.1920928955078125e-07) continue;
        mu1 = (mu1 + i * p_i) / q1;
        double mu2 = (mu - q1 * mu1) / q2;
        double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    base[256] = (u32)(int)max_val;
}

int launch_otsu(cbv_ctx* ctx, u32* aux, int tiles, int total, int batch)
{
    prof_begin(ctx, CBV_K_OTSU);
    hipLaunchKernelGGL(k_otsu, dim3(batch), dim3(64), 0, ctx->stream, aux, tiles, total);
    prof_end(ctx, CBV_K_OTSU);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

__global__ __launch_bounds__(256) void k_threshold(const u8* __restrict__ blur, u8* __restrict__ binary,
                                                    const u32* __restrict__ aux, int tiles_total, size_t n)
{
    const u32 t = aux[(size_t)blockIdx.z * aux_words(tiles_total) + (size_t)tiles_total * 256 + 2 + 256];
    const u8* s = blur + (size_t)blockIdx.z * n;
    u8* d = binary + (size_t)blockIdx.z * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d[i] = s[i] > t ? 255 : 0;
}

int launch_threshold(cbv_ctx* ctx, const u8* blur, u8* binary, const u32* aux, int tiles, int w, int h, int batch)
{
    size_t n = (size_t)w * h;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    prof_begin(ctx, CBV_K_THRESHOLD);
    hipLaunchKernelGGL(k_threshold, dim3(blocks, 1, batch), dim3(256), 0, ctx->stream, blur, binary, aux, tiles, n);
    prof_end(ctx, CBV_K_THRESHOLD);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// cv2.CLAHE.apply on a single-channel image (the `clahe` attribute of ImageEnhancer, frame_enhancer.py:36,114):
// per-tile histograms of the image extended by REFLECT_101 to a multiple of the tile grid, k_clahe_lut (shared with
// correct_lighting), then the same float bilinear interpolation between the four neighbouring tile LUTs as
// k_clahe_apply.  Small and simple on purpose: the hot path applies CLAHE inside k_color_lab_hist / k_clahe_apply.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_clahe_gray_hist(const u8* __restrict__ src, int w, int h, int stride, ClaheGeom cg,
                                                          u32* __restrict__ aux)
{
    __shared__ u32 hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int tile = blockIdx.x, tx = tile % cg.tiles_x, ty = tile / cg.tiles_x;
    const int n = cg.tw * cg.th;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i / cg.tw, c = i - r * cg.tw;
        const int sy = d_reflect101(ty * cg.th + r, h), sx = d_reflect101(tx * cg.tw + c, w);
        atomicAdd(&hist[src[(size_t)sy * stride + sx]], 1u);
    }
    __syncthreads();
    aux[(size_t)tile * 256 + threadIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_clahe_gray_apply(const u8* __restrict__ src, u8* __restrict__ dst, int w, int h, int stride,
                                                           ClaheGeom cg, const u8* __restrict__ luts)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const float inv_th = 1.0f / cg.th, inv_tw = 1.0f / cg.tw;
    const float tyf = (float)y * inv_th - 0.5f, txf = (float)x * inv_tw - 0.5f;
    const int ty1 = d_floor_f(tyf), tx1 = d_floor_f(txf);
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya, xa = txf - (float)tx1, xa1 = 1.0f - xa;
    const int r1 = max(ty1, 0), r2 = min(ty1 + 1, cg.tiles_y - 1), c1 = max(tx1, 0), c2 = min(tx1 + 1, cg.tiles_x - 1);
    const int v = src[(size_t)y * stride + x];
    const u8* l1 = luts + (size_t)r1 * cg.tiles_x * 256;
    const u8* l2 = luts + (size_t)r2 * cg.tiles_x * 256;
    const float ra = (float)l1[c1 * 256 + v] * xa1 + (float)l1[c2 * 256 + v] * xa;
    const float rb = (float)l2[c1 * 256 + v] * xa1 + (float)l2[c2 * 256 + v] * xa;
    dst[(size_t)y * w + x] = (u8)__builtin_amdgcn_cvt_pk_u8_f32(ra * ya1 + rb * ya, 0, 0u); // round-half-even + saturate
}

int launch_clahe_gray(cbv_ctx* ctx, const u8* src, int w, int h, int stride, ClaheGeom cg, u32* aux, u8* luts, u8* dst)
{
    const int tiles = cg.tiles_x * cg.tiles_y;
    hipLaunchKernelGGL(k_clahe_gray_hist, dim3(tiles), dim3(256), 0, ctx->stream, src, w, h, stride, cg, aux);
    CBV_HIP(ctx, hipGetLastError());
    const int rc = launch_clahe_lut(ctx, aux, luts, cg, 1, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(k_clahe_gray_apply, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, src, dst, w, h, stride, cg, luts);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}
