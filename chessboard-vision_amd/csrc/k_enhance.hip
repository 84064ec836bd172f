// Enhancement kernels except the bilateral filter:
//   k_color_lab_hist  apply_color_profile + BGR2LAB + per-tile L histograms
//   k_clahe_lut       clip / redistribute / cumulative LUT per tile
//   k_clahe_apply     bilinear LUT interpolation on L + LAB2BGR
//   k_sharpen         3x3 filter2D (REFLECT_101) + global min/max
//   k_norm_lut        NORM_MINMAX scale/shift -> 256-entry byte map
//   k_normalize       byte map
// All memory-bound: pixels move as 12-byte groups (4 BGR pixels = 3 dwords
// per lane, 768 contiguous bytes per wave); tables sit in LDS.
#include "cbv_device.h"

// ---------------------------------------------------------------------------
__global__ void k_reset_aux(u32* aux, int tiles, int batch)
{
    size_t per = aux_words(tiles);
    size_t total = per * batch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t k = i % per;
        aux[i] = (k == (size_t)tiles * 256) ? 255u : 0u; // [min] starts at 255, everything else 0
    }
}

int launch_reset_aux(cbv_ctx* ctx, u32* aux, int tiles, int batch)
{
    size_t total = aux_words(tiles) * batch;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    prof_begin(ctx, CBV_K_RESET);
    hipLaunchKernelGGL(k_reset_aux, dim3(blocks), dim3(256), 0, ctx->stream, aux, tiles, batch);
    prof_end(ctx, CBV_K_RESET);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// colour profile + BGR2LAB + tile histograms
// ---------------------------------------------------------------------------
// acc + c * x with a 24-bit multiply, c uniform (a scalar register): one v_mad_i32_i24, kept as one by the inline assembly
__device__ __forceinline__ int d_mad24s(int c, int x, int acc)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "s"(c), "v"(x), "v"(acc));
    return r;
}

struct ColorLds {
    int sdiv[256];
    int hdiv[256];
    u16 cbrt[LAB_CBRT_TAB_SIZE_B];
    u16 gamma[256];
    ProfileTabs pt;
    u32 hist[256 * 8]; // 8 bank-interleaved copies: hist[bin * 8 + (lane & 7)]
};

// v_perm_b32 helpers (byte k of the result = byte sel[k] of {hi, lo}: 0-3 from lo, 4-7 from hi, 0x0c = 0)
__device__ __forceinline__ u32 d_pack3(int x, int y, int z) // x | y << 8 | z << 16 of three values already in [0,255]
{
    const u32 t = __builtin_amdgcn_perm((u32)y, (u32)x, 0x0c0c0400u);
    return __builtin_amdgcn_perm((u32)z, t, 0x0c040100u);
}

// apply_color_profile for one pixel: byte maps, integer RGB2HSV_b, float HSV2RGB_b whose per-byte
// front end (sector, fraction, s/255, v/255) is tabulated.  Returns b | g << 8 | r << 16.
__device__ __forceinline__ u32 d_profile_px(const ColorLds& L, int b, int g, int r, bool radical)
{
    b = L.pt.csa[b];
    g = L.pt.csa[g];
    r = L.pt.csa[r];
    const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int s = (diff * L.sdiv[v] + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * L.hdiv[diff] + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0; // 0 <= h <= 180: saturate_cast<uchar> is the identity
    const int m = radical ? L.pt.hmask[h] : 0;
    const float fs = L.pt.s_f[m][s & 255];
    const float fv = L.pt.v_f[v];
    const float f = L.pt.hfr[h], f1 = 1.f - f;
    // tab = {v, v(1-s), v(1-s f), v(1-s(1-f))}; with s == 0 every entry equals v, as OpenCV's branch returns
    const float t1 = fv * (1.f - fs);
    const float t2 = fv * (1.f - fs * f);
    const float t3 = fv * (1.f - fs * f1);
    u32 T = __builtin_amdgcn_cvt_pk_u8_f32(fv * 255.0f, 0, 0u); // round-half-even + saturate + pack
    T = __builtin_amdgcn_cvt_pk_u8_f32(t1 * 255.0f, 1, T);
    T = __builtin_amdgcn_cvt_pk_u8_f32(t2 * 255.0f, 2, T);
    T = __builtin_amdgcn_cvt_pk_u8_f32(t3 * 255.0f, 3, T);
    return __builtin_amdgcn_perm(T, T, L.pt.hsel[h]);
}

// RGB2Lab_b (integer).  Returns L | a << 8 | b << 16; oL receives L.
// fw: RGB2Lab_b's nine coefficients, uniform (scalar registers)
__device__ __forceinline__ u32 d_bgr2lab_px(const ColorLds& L, const int (&fw)[9], int b, int g, int r, int& oL)
{
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);
    // LDS gathers are the scarce resource of this kernel (32 lanes/clk/CU, bank conflicts on
    // data-dependent addresses): three 16-bit gamma reads + nine 24-bit multiplies beat three
    // 16-byte reads of premultiplied rows.
    const int R = L.gamma[b], G = L.gamma[g], B = L.gamma[r]; // positional naming as in OpenCV
    // (each row a chain of three multiply-adds from the rounding constant: 3 instructions, where hipcc left to itself
    // forms two products and a three-way add)
    const int half = 1 << (LAB_SHIFT - 1);
    const int fX = L.cbrt[d_mad24s(fw[2], B, d_mad24s(fw[1], G, d_mad24s(fw[0], R, half))) >> LAB_SHIFT];
    const int fY = L.cbrt[d_mad24s(fw[5], B, d_mad24s(fw[4], G, d_mad24s(fw[3], R, half))) >> LAB_SHIFT];
    const int fZ = L.cbrt[d_mad24s(fw[8], B, d_mad24s(fw[7], G, d_mad24s(fw[6], R, half))) >> LAB_SHIFT];
    oL = d_sat8(D_DESCALE(Lscale * fY + Lshift, LAB_SHIFT2));
    const int oa = d_sat8(D_DESCALE(500 * (fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2));
    const int ob = d_sat8(D_DESCALE(200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2));
    return d_pack3(oL, oa, ob);
}

// grid: x = row slices of a tile, y = tile, z = frame.  One wave per row.
__global__ __launch_bounds__(256) void k_color_lab_hist(const u8* __restrict__ src, u8* __restrict__ dst,
                                                         u32* __restrict__ aux, Geom g, ClaheGeom cg,
                                                         const StaticTabs* __restrict__ st,
                                                         const ProfileTabs* __restrict__ pt, int rows_per_wg,
                                                         int do_profile, int do_lab, int tiles_total)
{
    __shared__ ColorLds L;
    lds_copy(L.sdiv, st->sdiv, sizeof(L.sdiv));
    lds_copy(L.hdiv, st->hdiv, sizeof(L.hdiv));
    lds_copy(L.cbrt, st->cbrt, sizeof(L.cbrt));
    lds_copy(L.gamma, st->gamma, sizeof(L.gamma));
    lds_copy(&L.pt, pt, sizeof(ProfileTabs));
    for (int i = threadIdx.x; i < 256 * 8; i += blockDim.x) L.hist[i] = 0;
    __syncthreads();

    const int tile = blockIdx.y;
    const int tx = tile % cg.tiles_x, ty = tile / cg.tiles_x;
    const size_t fo = (size_t)blockIdx.z * g.frame_stride;
    const u8* sf = src + fo;
    u8* df = dst + fo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_wg;
    const int r1 = min(r0 + rows_per_wg, cg.th);
    const int ex0 = tx * cg.tw;
    const bool aligned = ((cg.tw & 3) == 0) && ((g.stride & 3) == 0);
    const int groups = (cg.tw + 3) >> 2;
    const bool radical = L.pt.radical != 0;
    int fw[9];
#pragma unroll
    for (int i = 0; i < 9; i++) fw[i] = st->fwd[i];
    const u32 hist_lane = (u32)(lane & 7) << 29; // v_alignbit(L, hist_lane, 27) = (L << 5) | (lane & 7) << 2: byte address of the copy

    for (int rr = r0 + wave; rr < r1; rr += 4) {
        const int ey = ty * cg.th + rr;
        const int sy = d_reflect101(ey, g.h);
        const bool row_in = ey < g.h;
        // one pixel: colour profile, BGR2LAB, histogram.  Straight-line code (no per-pixel predicate), so the four
        // pixels of a group interleave and their table gathers overlap.
        auto do_px = [&](int b, int gg, int r) -> u32 {
            u32 v = (u32)b | ((u32)gg << 8) | ((u32)r << 16);
            if (do_profile) {
                v = d_profile_px(L, b, gg, r, radical);
                b = v & 255;
                gg = (v >> 8) & 255;
                r = (v >> 16) & 255;
            }
            if (do_lab) {
                int oL;
                v = d_bgr2lab_px(L, fw, b, gg, r, oL);
                atomicAdd((u32*)((u8*)L.hist + __builtin_amdgcn_alignbit((u32)oL, hist_lane, 27)), 1u);
            }
            return v;
        };
        for (int gi = lane; gi < groups; gi += 64) {
            const int ex = ex0 + gi * 4;
            const int npx = min(4, cg.tw - gi * 4);
            if (aligned && npx == 4 && ex + 3 < g.w) {
                const u32* p = (const u32*)(sf + (size_t)sy * g.stride + (size_t)ex * 3);
                const u32 d0 = p[0], d1 = p[1], d2 = p[2];
                // bytes: b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
                const u32 P0 = do_px(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u);
                const u32 P1 = do_px(d0 >> 24, d1 & 255u, (d1 >> 8) & 255u);
                const u32 P2 = do_px((d1 >> 16) & 255u, d1 >> 24, d2 & 255u);
                const u32 P3 = do_px((d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24);
                if (row_in) {
                    u32* q = (u32*)(df + (size_t)ey * g.stride + (size_t)ex * 3);
                    q[0] = __builtin_amdgcn_perm(P1, P0, 0x04020100u);
                    q[1] = __builtin_amdgcn_perm(P2, P1, 0x05040201u);
                    q[2] = __builtin_amdgcn_perm(P3, P2, 0x06050402u);
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) { // tile / row ends, unaligned rows: pixel by pixel (REFLECT_101 past the image)
                if (k < npx) {
                    const int sx = d_reflect101(ex + k, g.w);
                    const u8* p = sf + (size_t)sy * g.stride + (size_t)sx * 3;
                    const u32 P = do_px(p[0], p[1], p[2]);
                    if (row_in && ex + k < g.w) {
                        u8* q = df + (size_t)ey * g.stride + (size_t)(ex + k) * 3;
                        q[0] = (u8)(P & 255);
                        q[1] = (u8)((P >> 8) & 255);
                        q[2] = (u8)((P >> 16) & 255);
                    }
                }
            }
        }
    }
    if (do_lab) {
        __syncthreads();
        u32* hist = aux + (size_t)blockIdx.z * aux_words(tiles_total) + (size_t)tile * 256;
        for (int bin = threadIdx.x; bin < 256; bin += blockDim.x) {
            u32 s = 0;
#pragma unroll
            for (int c = 0; c < 8; c++) s += L.hist[bin * 8 + c];
            if (s) atomicAdd(&hist[bin], s);
        }
    }
}

int launch_color_lab_hist(cbv_ctx* ctx, const u8* src, u8* lab, u32* aux, Geom g, ClaheGeom cg, int batch,
                          int do_profile, int do_lab)
{
    // A workgroup stages ~35 KB of tables, so give it as many rows of its tile as the launch can
    // afford while still putting >= ~1024 workgroups on the chip.
    int slices = 1024 / (cg.tiles_x * cg.tiles_y * batch);
    if (slices < 1) slices = 1;
    if (slices > (cg.th + 3) / 4) slices = (cg.th + 3) / 4;
    const int rows_per_wg = (cg.th + slices - 1) / slices;
    dim3 grid((cg.th + rows_per_wg - 1) / rows_per_wg, cg.tiles_x * cg.tiles_y, batch);
    prof_begin(ctx, CBV_K_COLOR_LAB_HIST);
    hipLaunchKernelGGL(k_color_lab_hist, grid, dim3(256), 0, ctx->stream, src, lab, aux, g, cg, ctx->tabs, ctx->ptabs,
                       rows_per_wg, do_profile, do_lab, cg.tiles_x * cg.tiles_y);
    prof_end(ctx, CBV_K_COLOR_LAB_HIST);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// CLAHE LUT: one workgroup (256 threads = 256 bins) per tile
// ---------------------------------------------------------------------------
// `packed` (may be null): per frame [tiles_y + 1 bands][tiles_x + 1 column pairs][256] words holding, for L value v,
// the four LUT bytes k_clahe_apply interpolates between: byte 0 = (ty1, tx1), 1 = (ty1, tx2), 2 = (ty2, tx1),
// 3 = (ty2, tx2), where band b has ty1 = clamp(b - 1), ty2 = clamp(b) and pair p has tx1 = clamp(p - 1),
// tx2 = clamp(p).  A tile writes its byte into every word it is a corner of (one to nine of them), so the apply
// kernel does ONE gather per pixel instead of four.
__global__ __launch_bounds__(256) void k_clahe_lut(u32* __restrict__ aux, u8* __restrict__ luts, ClaheGeom cg,
                                                    int tiles_total, u8* __restrict__ packed, int self_clean)
{
    __shared__ int red[4];
    __shared__ int scan[256];
    const int tile = blockIdx.x, t = threadIdx.x;
    u32* hist = aux + (size_t)blockIdx.y * aux_words(tiles_total) + (size_t)tile * 256;
    int hv = (int)hist[t];
    if (self_clean) {
        // this kernel is the histograms' only reader, and the frame's [min, max] words are written later in the pass
        // (k_sharpen) and read after that: leaving both as k_reset_aux would spares the next pass that launch
        hist[t] = 0u;
        if (tile == 0 && t < 2) aux[(size_t)blockIdx.y * aux_words(tiles_total) + (size_t)tiles_total * 256 + t] = t == 0 ? 255u : 0u;
    }
    if (cg.clip > 0) {
        int excess = hv > cg.clip ? hv - cg.clip : 0;
        hv = min(hv, cg.clip);
        int ws = (int)wave_sum_u32((u32)excess);
        if ((t & 63) == 0) red[t >> 6] = ws;
        __syncthreads();
        int clipped = red[0] + red[1] + red[2] + red[3];
        int batch = clipped / 256;
        int residual = clipped - batch * 256;
        hv += batch;
        if (residual != 0) {
            int step = max(256 / residual, 1);
            if (t % step == 0 && t / step < residual) hv++;
        }
    }
    // inclusive scan over the 256 bins
    scan[t] = hv;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int v = t >= o ? scan[t - o] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    float f = (float)scan[t] * cg.lut_scale;
    const u8 lv = d_sat8_f(f);
    luts[((size_t)blockIdx.y * tiles_total + tile) * 256 + t] = lv;
    if (packed) {
        const int tx = tile % cg.tiles_x, ty = tile / cg.tiles_x;
        const int pairs = cg.tiles_x + 1, bands = cg.tiles_y + 1;
        u8* pf = packed + (size_t)blockIdx.y * bands * pairs * 1024;
        for (int b = ty; b <= ty + 1; b++) {
            // tile row ty is ty2 of band ty and ty1 of band ty + 1; clamping adds ty1 of band 0 / ty2 of the last band
            for (int yrole = 0; yrole < 2; yrole++) {
                const int want = yrole == 0 ? min(max(b - 1, 0), cg.tiles_y - 1) : min(b, cg.tiles_y - 1);
                if (want != ty) continue;
                for (int p = tx; p <= tx + 1; p++)
                    for (int xrole = 0; xrole < 2; xrole++) {
                        const int wantx = xrole == 0 ? min(max(p - 1, 0), cg.tiles_x - 1) : min(p, cg.tiles_x - 1);
                        if (wantx != tx) continue;
                        pf[(((size_t)b * pairs + p) * 256 + t) * 4 + yrole * 2 + xrole] = lv;
                    }
            }
        }
    }
}

int launch_clahe_lut(cbv_ctx* ctx, u32* aux, u8* luts, ClaheGeom cg, int batch, u32* packed, int self_clean)
{
    int tiles = cg.tiles_x * cg.tiles_y;
    prof_begin(ctx, CBV_K_CLAHE_LUT);
    hipLaunchKernelGGL(k_clahe_lut, dim3(tiles, batch), dim3(256), 0, ctx->stream, aux, luts, cg, tiles, (u8*)packed, self_clean);
    prof_end(ctx, CBV_K_CLAHE_LUT);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// CLAHE interpolation + LAB2BGR.  Rows are cut into bands that share the same
// pair of LUT rows (ty1, ty2), so a workgroup stages just two LUT rows.
// grid: x = chunk of `rows_per_wg` rows inside a band, y = band, z = frame.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int d_ab_to_xz(int i)
{
    if (i <= 3390) return i * 108 / 841 - LAB_BASE * 16 / 116 * 108 / 841;
    return i * i / LAB_BASE * i / LAB_BASE;
}

#define IFY_BIAS (128 * LAB_BASE / 200 - 1) // bdiv = ((b * 41943 + 16) >> 9) - IFY_BIAS
#define CLAHE_MAX_TILES_X 32
// RG = false: the whole-frame launch, with the region code compiled out
// Register budget: 48 VGPRs.  Three of its waves then fit beside the three 120-register waves a 768-lane bilateral
// workgroup keeps on every SIMD (the other lane of the pipeline runs the bilateral most of the time); a build that kept all
// four pixels of a group in flight took 60, only two waves fitted, and the path lost 1 % although the kernel alone was faster.
template <bool RG>
__global__ __launch_bounds__(256) void k_clahe_apply(const u8* __restrict__ lab, const u32* __restrict__ packed,
                                                      u8* __restrict__ dst, Geom g, ClaheGeom cg,
                                                      const StaticTabs* __restrict__ st, int rows_per_wg,
                                                      int tiles_total, PxRect cov, int invert, SatGate gate)
{
    // region-limited enhancement: only the pixels of `cov` (x in 4-pixel groups), or (invert) all the others, and in
    // that complement pass nothing for frames whose region already holds 0 and 255 (cbv_internal.h)
    if (RG && sat_gate_closed(gate, blockIdx.z)) return;
    // static LDS: table addresses become ds_read immediates (no per-lookup base add)
    __shared__ u8 inv_gamma[INV_GAMMA_TAB_SIZE]; // sRGBInvGammaTab_b as bytes (every entry is <= 255): the index IS the address
    __shared__ u32 lab_yf[256];          // y | (ify + IFY_BIAS) << 16, read back as two u16 at one address register
    __shared__ u16 adiv_t[256];          // (5 a 53687 + 2^7) >> 13
    __shared__ u16 bdiv_t[256];          // (b 41943 + 2^4) >> 9
    extern __shared__ __align__(16) u32 pk[]; // [tiles_x + 1][256]: this band's packed corner words (k_clahe_lut)

    // band b holds the rows whose unclamped ty1 is b - 1
    const int band = blockIdx.y;
    const float inv_th = 1.0f / cg.th, inv_tw = 1.0f / cg.tw;
    // first row of the band: smallest y with floor(y*inv_th - 0.5) >= band-1; search near the estimate
    int yb0, yb1;
    {
        int est = (int)((band - 0.5f) * cg.th);
        int y = min(max(est - 2, 0), g.h);
        while (y < g.h && d_floor_f((float)y * inv_th - 0.5f) < band - 1) y++;
        yb0 = y;
        int est1 = (int)((band + 0.5f) * cg.th);
        y = min(max(est1 - 2, yb0), g.h);
        while (y < g.h && d_floor_f((float)y * inv_th - 0.5f) < band) y++;
        yb1 = y;
    }
    const int y0 = yb0 + blockIdx.x * rows_per_wg;
    if (y0 >= yb1) return;
    const int y1 = min(y0 + rows_per_wg, yb1);
    if (RG && !invert && (y1 <= cov.y0 || y0 >= cov.y1)) return; // none of this workgroup's rows is in the region
    const int gx0 = cov.x0 >> 2, gx1 = (cov.x1 + 3) >> 2;

    for (int i = threadIdx.x; i < INV_GAMMA_TAB_SIZE; i += blockDim.x) inv_gamma[i] = (u8)st->inv_gamma[i];
    // u16 pairs (y, ify) as one word; ify is stored with the constant of `ify - bdiv` already added (it fits 16 bits:
    // ify <= LAB_BASE * 1.01), so that argument costs one subtraction per pixel and the other one three-way add
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        lab_yf[i] = (u32)st->lab_yf[2 * i] | ((u32)(st->lab_yf[2 * i + 1] + IFY_BIAS) << 16);
        adiv_t[i] = (u16)((5 * i * 53687 + (1 << 7)) >> 13); // <= 8356
        bdiv_t[i] = (u16)((i * 41943 + (1 << 4)) >> 9);      // <= 20889
    }
    // Lab2RGBinteger's nine coefficients: uniform loads, they stay in scalar registers
    int invc[9];
#pragma unroll
    for (int i = 0; i < 9; i++) invc[i] = st->inv[i];
    const int pairs = cg.tiles_x + 1;
    lds_copy(pk, packed + ((size_t)blockIdx.z * (cg.tiles_y + 1) + band) * pairs * 256, pairs * 1024);
    __syncthreads();

    const size_t fo = (size_t)blockIdx.z * g.frame_stride;
    const u8* sf = lab + fo;
    u8* df = dst + fo;
    const bool aligned = (g.stride & 3) == 0;
    const int groups = (g.w + 3) >> 2;
    const int shift = LAB_SHIFT + (LAB_BASE_SHIFT - INV_GAMMA_SHIFT);

    // x interpolation parameters of a 4-pixel column group (tile pair, xa)
    auto xparam = [&](int gi, int (&op)[4], float (&xa)[4]) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float txf = (float)(gi * 4 + k) * inv_tw - 0.5f;
            int tx1 = d_floor_f(txf);
            xa[k] = txf - (float)tx1;
            op[k] = min(max(tx1 + 1, 0), cg.tiles_x) * 1024; // byte offset of the pixel's column pair in pk
        }
    };
    // A pixel in three steps, so that the four pixels of a group run through each step together (straight-line code, no
    // per-pixel predicate: their gathers overlap).
    // 1: CLAHE's bilinear interpolation between the four corner LUTs on L, then the arguments of Lab2RGBinteger's two
    //    ab -> xz conversions
    //    (v4 = 4 L, a2 = 2 a, b2 = 2 b: byte offsets into the tables, formed by the caller with one shift and one mask
    //    each, both full-rate instructions; a bit-field extract and a shift are both half-rate)
    auto px_front = [&](u32 v4, u32 a2, u32 b2, int opk, float xak, float ya, float ya1, int& yv, int& ix, int& iz) {
        const float xa1k = 1.0f - xak; // (recomputed per pixel: one instruction against a register held across the rows)
        const u32 cw = *(const u32*)((const u8*)pk + opk + v4); // the four corner LUT values of L
        float ra = (float)(cw & 255u) * xa1k + (float)((cw >> 8) & 255u) * xak;
        float rb = (float)((cw >> 16) & 255u) * xa1k + (float)(cw >> 24) * xak;
        float res = ra * ya1 + rb * ya;
        const int LL = (int)__builtin_amdgcn_cvt_pk_u8_f32(res, 0, 0u); // round-half-even + saturate
        // LDS has room in this kernel (a third of the vector pipe's time): y and ify as two 16-bit reads, adiv / bdiv as
        // table reads instead of a multiply-add and a shift each
        const u16* yfp = (const u16*)((const u8*)lab_yf + LL * 4);
        yv = (int)yfp[0];
        const int ifyb = (int)yfp[1]; // ify + IFY_BIAS
        // adiv = adiv_t[a] - 128 LAB_BASE / 500, bdiv = bdiv_t[b] - IFY_BIAS
        ix = ifyb + (int)*(const u16*)((const u8*)adiv_t + a2) - (128 * LAB_BASE / 500 + IFY_BIAS);
        iz = ifyb - (int)*(const u16*)((const u8*)bdiv_t + b2);
    };
    // 2: d_ab_to_xz.  Its cubic branch (i > 3390, i.e. f(t) above 6/29: every pixel that is not nearly black in that
    //    coordinate) is evaluated for all lanes; the linear branch costs more instructions than the cubic one and is
    //    entered by the WAVE only when some lane needs it (px_fix_low behind a ballot).
    auto cubic = [](int i) { return (int)(((u32)__mul24((int)((u32)__mul24(i, i) >> LAB_BASE_SHIFT), i)) >> LAB_BASE_SHIFT); };
    auto px_fix_low = [](int i, int& v) {
        if (i <= 3390) v = i * 108 / 841 - LAB_BASE * 16 / 116 * 108 / 841;
    };
    // 3: the inverse matrix, the inverse gamma table, packing
    auto px_back = [&](int xv, int yv, int zv) -> u32 {
        // |x|,|y|,|z| < 2^17 and |coefficient| < 2^14: 24-bit multiplies are exact
        // (a chain of three multiply-adds that starts from the rounding constant: 3 instructions a channel; left to itself
        // hipcc forms two products and a three-way add, 4 instructions)
        const int half = 1 << (shift - 1);
        int ro = d_mad24s(invc[2], zv, d_mad24s(invc[1], yv, d_mad24s(invc[0], xv, half))) >> shift;
        int go = d_mad24s(invc[5], zv, d_mad24s(invc[4], yv, d_mad24s(invc[3], xv, half))) >> shift;
        int bo = d_mad24s(invc[8], zv, d_mad24s(invc[7], yv, d_mad24s(invc[6], xv, half))) >> shift;
        ro = min(max(ro, 0), INV_GAMMA_TAB_SIZE - 1);
        go = min(max(go, 0), INV_GAMMA_TAB_SIZE - 1);
        bo = min(max(bo, 0), INV_GAMMA_TAB_SIZE - 1);
        return d_pack3(inv_gamma[bo], inv_gamma[go], inv_gamma[ro]);
    };
    auto do_px = [&](int v, int aa, int bb, int opk, float xak, float ya, float ya1) -> u32 {
        int yv, ix, iz;
        px_front((u32)v * 4, (u32)aa * 2, (u32)bb * 2, opk, xak, ya, ya1, yv, ix, iz);
        return px_back(d_ab_to_xz(ix), yv, d_ab_to_xz(iz));
    };
    // one group of one row
    auto do_row = [&](int gi, int y, const int (&op)[4], const float (&xa)[4]) {
        const int x0 = gi * 4;
        const int npx = min(4, g.w - x0);
        const float tyf = (float)y * inv_th - 0.5f;
        const float ya = tyf - (float)d_floor_f(tyf), ya1 = 1.0f - ya;
        const u8* p = sf + (size_t)y * g.stride + (size_t)x0 * 3;
        u8* q = df + (size_t)y * g.stride + (size_t)x0 * 3;
        if (aligned && npx == 4) {
            const u32* pw = (const u32*)p;
            const u32 d0 = pw[0], d1 = pw[1], d2 = pw[2];
            // bytes: L0 a0 b0 L1 | a1 b1 L2 a2 | b2 L3 a3 b3
            // two pixels at a time: enough independent work to overlap the gathers, few enough live values for 48 VGPRs
            auto pair = [&](u32 v0, u32 a0, u32 b0, u32 v1, u32 a1, u32 b1, int k0, u32& Pa, u32& Pb) {
                int yv0, ix0, iz0, yv1, ix1, iz1;
                px_front(v0, a0, b0, op[k0], xa[k0], ya, ya1, yv0, ix0, iz0);
                px_front(v1, a1, b1, op[k0 + 1], xa[k0 + 1], ya, ya1, yv1, ix1, iz1);
                int xv0 = cubic(ix0), zv0 = cubic(iz0), xv1 = cubic(ix1), zv1 = cubic(iz1);
                if (__builtin_amdgcn_ballot_w64(min(min(ix0, iz0), min(ix1, iz1)) <= 3390)) { // wave-uniform: some lane is on the linear branch
                    px_fix_low(ix0, xv0);
                    px_fix_low(iz0, zv0);
                    px_fix_low(ix1, xv1);
                    px_fix_low(iz1, zv1);
                }
                Pa = px_back(xv0, yv0, zv0);
                Pb = px_back(xv1, yv1, zv1);
            };
            u32 P0, P1, P2, P3;
            pair((d0 << 2) & 0x3FCu, (d0 >> 7) & 0x1FEu, (d0 >> 15) & 0x1FEu, (d0 >> 22) & 0x3FCu, (d1 << 1) & 0x1FEu, (d1 >> 7) & 0x1FEu, 0, P0, P1);
            pair((d1 >> 14) & 0x3FCu, (d1 >> 23) & 0x1FEu, (d2 << 1) & 0x1FEu, (d2 >> 6) & 0x3FCu, (d2 >> 15) & 0x1FEu, (d2 >> 23) & 0x1FEu, 2, P2, P3);
            u32* qw = (u32*)q;
            qw[0] = __builtin_amdgcn_perm(P1, P0, 0x04020100u);
            qw[1] = __builtin_amdgcn_perm(P2, P1, 0x05040201u);
            qw[2] = __builtin_amdgcn_perm(P3, P2, 0x06050402u);
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { // ragged row end or unaligned rows: pixel by pixel
            if (k < npx) {
                const u32 P = do_px(p[3 * k], p[3 * k + 1], p[3 * k + 2], op[k], xa[k], ya, ya1);
                q[3 * k] = (u8)(P & 255);
                q[3 * k + 1] = (u8)((P >> 8) & 255);
                q[3 * k + 2] = (u8)((P >> 16) & 255);
            }
        }
    };
    if (RG && !invert && (gx0 > 0 || gx1 < groups)) {
        // region pass on a part of the columns: (row, group) items dealt out over the lanes, so that no lane idles on
        // the columns outside the region (x parameters per item: ~10 of its ~370 instructions)
        const int c0 = min(gx0, groups), ngr = min(gx1, groups) - c0;
        const int r0 = max(y0, cov.y0), nr = min(y1, cov.y1) - r0;
        for (int item = threadIdx.x; item < ngr * nr; item += blockDim.x) {
            const int ry = item / ngr, gi = c0 + (item - ry * ngr);
            int op[4];
            float xa[4];
            xparam(gi, op, xa);
            do_row(gi, r0 + ry, op, xa);
        }
        return;
    }
    // a lane keeps its 4-pixel column group and walks the chunk's rows: the x interpolation
    // parameters are computed once per group instead of once per pixel
    for (int gi = threadIdx.x; gi < groups; gi += blockDim.x) {
        const bool col_in = gi >= gx0 && gi < gx1;
        if (RG && !invert && !col_in) continue;
        int op[4];
        float xa[4];
        xparam(gi, op, xa);
        for (int y = y0; y < y1; y++) {
            const bool row_in = y >= cov.y0 && y < cov.y1;
            if (RG && (invert ? (row_in && col_in) : !row_in)) continue;
            do_row(gi, y, op, xa);
        }
    }
}

PxRect clahe_region_cover(Geom g, PxRect need)
{
    PxRect c = {need.x0 & ~3, need.y0, (need.x1 + 3) & ~3, need.y1};
    c.x1 = c.x1 > g.w ? g.w : c.x1;
    return c;
}

int launch_clahe_apply(cbv_ctx* ctx, const u8* lab, const u32* packed, u8* dst, Geom g, ClaheGeom cg, int batch, const EnhanceRegion* er)
{
    PxRect cov = {0, 0, g.w, g.h};
    int invert = 0;
    SatGate gate = {nullptr, 0};
    if (er) {
        cov = clahe_region_cover(g, er->px);
        invert = er->invert;
        gate = er->gate;
    }
    // 8 rows per workgroup amortise the 18 KB of tables it stages; a launch of one or two frames (the live-camera
    // case) would then put only ~150 workgroups on 256 CUs, so small batches take fewer rows, down to one
    int rows_per_wg = 8;
    while (rows_per_wg > 1 && (size_t)((cg.th + rows_per_wg) / rows_per_wg) * (cg.tiles_y + 1) * batch < 1024) rows_per_wg >>= 1;
    int tiles = cg.tiles_x * cg.tiles_y;
    if (cg.tiles_x > CLAHE_MAX_TILES_X) return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "clahe: tile grid wider than %d (%d)", CLAHE_MAX_TILES_X, cg.tiles_x);
    // a band is at most th rows (+1 for rounding)
    dim3 grid((cg.th + 1 + rows_per_wg - 1) / rows_per_wg, cg.tiles_y + 1, batch);
    prof_begin(ctx, CBV_K_CLAHE_APPLY);
    if (er)
        hipLaunchKernelGGL(k_clahe_apply<true>, grid, dim3(256), (size_t)(cg.tiles_x + 1) * 1024, ctx->stream, lab, packed, dst, g, cg, ctx->tabs,
                           rows_per_wg, tiles, cov, invert, gate);
    else
        hipLaunchKernelGGL(k_clahe_apply<false>, grid, dim3(256), (size_t)(cg.tiles_x + 1) * 1024, ctx->stream, lab, packed, dst, g, cg, ctx->tabs,
                           rows_per_wg, tiles, cov, invert, gate);
    prof_end(ctx, CBV_K_CLAHE_APPLY);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// 3x3 filter2D + global min/max.  Tile: 256 px x 16 rows; rows staged in LDS
// as raw bytes with a 1-pixel (3-byte) halo; each lane produces 4 px x 4 rows.
// ---------------------------------------------------------------------------
#define SH_TW 256
#define SH_TH 16
#define SH_PITCH (SH_TW * 3 + 8) // bytes per LDS row: 4 left (1 pad + 3 halo), 768, 3 halo + 1 pad

__global__ __launch_bounds__(256) void k_sharpen(const u8* __restrict__ src, u8* __restrict__ dst,
                                                  u32* __restrict__ aux, int tiles_total, Geom g, float k0, float k1,
                                                  float k2, float k3, float k4, float k5, float k6, float k7,
                                                  float k8, int tiles_xn, int tiles_n)
{
    __shared__ __attribute__((aligned(16))) u8 tile[(SH_TH + 2) * SH_PITCH];
    __shared__ int red[8];
    const int tid = xcd_remap(blockIdx.x, tiles_n);
    const int tyi = tid / tiles_xn, txi = tid - tyi * tiles_xn;
    const int x0 = txi * SH_TW, y0 = tyi * SH_TH;
    const size_t fo = (size_t)blockIdx.z * g.frame_stride;
    const u8* sf = src + fo;
    u8* df = dst + fo;
    const int wb = g.w * 3;
    // stage rows y0-1 .. y0+SH_TH, bytes [x0*3-4, x0*3+768+4)
    const int b0 = x0 * 3 - 4;
    const bool aligned = (g.stride & 3) == 0;
    for (int i = threadIdx.x; i < (SH_TH + 2) * (SH_PITCH / 4); i += blockDim.x) {
        const int r = i / (SH_PITCH / 4), c = i - r * (SH_PITCH / 4);
        const int sy = d_reflect101(y0 - 1 + r, g.h);
        const int bs = b0 + c * 4;
        u32 v;
        if (aligned && bs >= 0 && bs + 3 < wb) {
            v = *(const u32*)(sf + (size_t)sy * g.stride + bs);
        } else {
            v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int bx = bs + k;
                // byte bx belongs to pixel floor(bx/3) (may be -1 or w); reflect the pixel, keep the channel
                int pxl = bx >= 0 ? bx / 3 : -((2 - bx) / 3);
                int ch = bx - pxl * 3;
                int spx = d_reflect101(pxl, g.w);
                u32 byte = (pxl >= -1 && pxl <= g.w) ? sf[(size_t)sy * g.stride + spx * 3 + ch] : 0;
                v |= byte << (8 * k);
            }
        }
        *(u32*)&tile[r * SH_PITCH + c * 4] = v;
    }
    __syncthreads();

    const int lane_x = threadIdx.x & 63, rq = threadIdx.x >> 6; // 64 column groups x 4 row quads
    const int x = x0 + lane_x * 4;
    int mn = 255, mx = 0;
    if (x < g.w) {
        const int npx = min(4, g.w - x);
        // window of a lane: tile bytes [12*lane_x, 12*lane_x + 24); bytes 1..18 of it are
        // the 3-byte halo, the 12 output bytes and the right halo.
        float f[3][18];
        auto load_row = [&](int slot, int trow) {
            const u32* wr = (const u32*)&tile[trow * SH_PITCH + lane_x * 12];
            u32 wv[6];
#pragma unroll
            for (int k = 0; k < 6; k++) wv[k] = wr[k];
#pragma unroll
            for (int j = 0; j < 18; j++) f[slot][j] = (float)((wv[(j + 1) >> 2] >> (((j + 1) & 3) * 8)) & 255);
        };
        load_row(0, rq * 4 + 0);
        load_row(1, rq * 4 + 1);
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int ly = rq * 4 + rr;
            const int y = y0 + ly;
            load_row((rr + 2) % 3, ly + 2);
            const float* r0 = f[rr % 3];
            const float* r1 = f[(rr + 1) % 3];
            const float* r2 = f[(rr + 2) % 3];
            Px4 out;
            out.d[0] = out.d[1] = out.d[2] = 0;
#pragma unroll
            for (int j = 0; j < 12; j++) {
                float s = k0 * r0[j];
                s = s + k1 * r0[j + 3];
                s = s + k2 * r0[j + 6];
                s = s + k3 * r1[j];
                s = s + k4 * r1[j + 3];
                s = s + k5 * r1[j + 6];
                s = s + k6 * r2[j];
                s = s + k7 * r2[j + 3];
                s = s + k8 * r2[j + 6];
                int o = d_sat8_f(s);
                px_set(out, j, o);
                if (j < npx * 3 && y < g.h) {
                    mn = min(mn, o);
                    mx = max(mx, o);
                }
            }
            if (y < g.h) {
                u8* q = df + (size_t)y * g.stride + (size_t)x * 3;
                if (aligned && npx == 4) {
                    u32* qw = (u32*)q;
                    qw[0] = out.d[0];
                    qw[1] = out.d[1];
                    qw[2] = out.d[2];
                } else {
#pragma unroll
                    for (int j = 0; j < 12; j++)
                        if (j < npx * 3) q[j] = (u8)px_get(out, j);
                }
            }
        }
    }
    mn = wave_min_i32(mn);
    mx = wave_max_i32(mx);
    if ((threadIdx.x & 63) == 0) {
        red[rq] = mn;
        red[4 + rq] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32* mm = aux + (size_t)blockIdx.z * aux_words(tiles_total) + (size_t)tiles_total * 256;
        atomicMin(&mm[0], (u32)min(min(red[0], red[1]), min(red[2], red[3])));
        atomicMax(&mm[1], (u32)max(max(red[4], red[5]), max(red[6], red[7])));
    }
}

// ---------------------------------------------------------------------------
// Fast path for kernels  a * (3x3 box) + (c - a) * centre  with small integer a, c (the
// reference's [[-1,-1,-1],[-1,9,-1],[-1,-1,-1]] is a = -1, c = 9).  All sums are exact integers in
// f32, so the box sum is separated: per staged row 2 adds per byte (reused by three output rows),
// per output byte 2 adds + 1 mul + 1 fma + min/max + ONE v_cvt_pk_u8_f32 (convert, saturate, pack).
// The filter is byte-wise with +-3-byte neighbours, so tiles are cut in BYTES (1 KiB x 16 rows,
// 16-byte lanes, uint4 loads and stores), not in pixels.  f32 add/fma are the full-rate VALU ops on
// gfx950 (tools/ubench_valu.hip), hence floats rather than packed integer arithmetic.
// ---------------------------------------------------------------------------
#define SHB_TB 1024                 // tile width in bytes
#define SHB_TH 16                   // tile height in rows
#define SHB_PITCH (SHB_TB + 32)     // LDS row: 16 B halo | 1024 B | 16 B halo

#ifdef SH_TIMING
// diagnostic build: per-workgroup phase stamps of k_sharpen_box (tools/sharpen_timeline.py)
#define SH_MAX_STAMPS (1 << 16)
__device__ unsigned long long g_sh_stamps[SH_MAX_STAMPS * 5];
__device__ unsigned int g_sh_count;
#define SH_STAMP(k) do { if (threadIdx.x == 0) sh_t[k] = __builtin_readcyclecounter(); } while (0)
#else
#define SH_STAMP(k) do { } while (0)
#endif

typedef unsigned short shb_u2 __attribute__((ext_vector_type(2)));
typedef short shb_s2 __attribute__((ext_vector_type(2)));

// PACKED: the same arithmetic in packed 16-bit integers, two bytes per instruction, for kernels whose intermediate
// a * box + (c - a) * centre stays inside int16 (|a| * 2295 + |c - a| * 255 <= 32767; the reference's -1 / 9 kernel
// gives [-2295, 2550]).  Exact either way: the float form only ever holds small integers.  About 7 vector
// instructions per output byte instead of 20 (profiles/r02/sq_counters.txt).
template <bool PACKED>
__global__ __launch_bounds__(256) void k_sharpen_box(const u8* __restrict__ src, u8* __restrict__ dst,
                                                      u32* __restrict__ aux, int tiles_total, Geom g, float a, float ca,
                                                      TileSet ts, SatGate gate, ShRegion rg)
{
    if (sat_gate_closed(gate, blockIdx.z)) return; // complement pass of a frame whose region already holds 0 and 255
    __shared__ __attribute__((aligned(16))) u8 tile[(SHB_TH + 2) * SHB_PITCH];
    __shared__ int red[8];
#ifdef SH_TIMING
    unsigned long long sh_t[4] = {0, 0, 0, 0};
#endif
    SH_STAMP(0);
    int tyi, txi;
    tileset_at(ts, xcd_remap(blockIdx.x, ts.cum[ts.n]), txi, tyi);
    const int xb0 = txi * SHB_TB, y0 = tyi * SHB_TH;
    const size_t fo = (size_t)blockIdx.z * g.frame_stride;
    const u8* sf = src + fo;
    u8* df = dst + fo;
    const int wb = g.w * 3;
    const bool al16 = (g.stride & 15) == 0;
    // stage rows y0-1 .. y0+16, bytes [xb0-16, xb0+1040) as 66 uint4 per row.  All of a lane's loads are issued
    // before the first one is consumed: written as one loop (load, store to LDS, next) the compiler serialises
    // five global-memory latencies per workgroup (measured with SH_TIMING: 21 k of a workgroup's 30 k clocks).
    constexpr int SHB_VPR = SHB_PITCH / 16, SHB_NV = (SHB_TH + 2) * SHB_VPR, SHB_VPT = (SHB_NV + 255) / 256;
    uint4 stg[SHB_VPT];
    bool plain[SHB_VPT];
#pragma unroll
    for (int k = 0; k < SHB_VPT; k++) {
        const int i = threadIdx.x + k * 256;
        plain[k] = false;
        if (i < SHB_NV) {
            const int r = i / SHB_VPR, c = i - r * SHB_VPR;
            const int sy = d_reflect101(y0 - 1 + r, g.h);
            const int bs = xb0 - 16 + c * 16;
            plain[k] = al16 && bs >= 0 && bs + 16 <= wb;
            if (plain[k]) stg[k] = *(const uint4*)(sf + (size_t)sy * g.stride + bs);
        }
    }
#pragma unroll
    for (int k = 0; k < SHB_VPT; k++) {
        const int i = threadIdx.x + k * 256;
        if (i >= SHB_NV) continue;
        // vectors that are not whole inside the row are zero here; the few that straddle a row end are filled below
        *(uint4*)&tile[i * 16] = plain[k] ? stg[k] : make_uint4(0, 0, 0, 0); // i = r * SHB_VPR + c and SHB_PITCH = 16 * SHB_VPR
    }
    // Row ends (REFLECT_101 on pixels, channel kept): at most three vectors per staged row hold bytes within one pixel
    // of the row without lying whole inside it.  They get their own pass, one lane per vector, so the byte loop runs in
    // ONE wave of an edge tile; inside the loop above it ran, one lane active, in nearly every wave of such a tile (the
    // straddling vectors sit 66 lanes apart) and made up half of the kernel's vector instructions.
    if (!al16 || xb0 == 0 || xb0 + SHB_TB + 16 > wb) { // workgroup-uniform: the tile touches a row end (or rows are unaligned)
        __syncthreads(); // the zero fill above
        const int nspecial = al16 ? (SHB_TH + 2) * 3 : SHB_NV; // unaligned rows: every vector takes the byte path
        for (int t = threadIdx.x; t < nspecial; t += 256) {
            int r, c;
            if (al16) {
                r = t / 3;
                const int side = t - r * 3;
                // side 0: the vector holding bytes -3 .. -1 (c = 0, first tile only); sides 1, 2: the vector holding
                // byte wb and the one after it (bytes wb .. wb + 2 may spill into it)
                c = side == 0 ? 0 : (wb - (xb0 - 16)) / 16 + (side - 1);
                if (side == 0 && xb0 != 0) continue;
                if (c < 0 || c >= SHB_VPR) continue;
            } else {
                r = t / SHB_VPR;
                c = t - r * SHB_VPR;
            }
            const int sy = d_reflect101(y0 - 1 + r, g.h);
            const int bs = xb0 - 16 + c * 16;
            if (al16 && bs >= 0 && bs + 16 <= wb) continue; // whole inside: already staged
            u32 w4[4] = {0, 0, 0, 0};
            if (bs + 16 > -3 && bs < wb + 3) { // only bytes within one pixel of the row are ever used
                for (int kk = 0; kk < 16; kk++) {
                    const int bx = bs + kk;
                    // byte bx belongs to pixel floor(bx / 3); reflect the pixel (REFLECT_101), keep the channel
                    const int pxl = bx >= 0 ? bx / 3 : -((2 - bx) / 3);
                    const int ch = bx - pxl * 3;
                    if (pxl >= -1 && pxl <= g.w) w4[kk >> 2] |= (u32)sf[(size_t)sy * g.stride + d_reflect101(pxl, g.w) * 3 + ch] << ((kk & 3) * 8);
                }
            }
            *(uint4*)&tile[(r * SHB_VPR + c) * 16] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
    }
    SH_STAMP(1);
    __syncthreads();
    SH_STAMP(2);

    const int lc = threadIdx.x & 63, rq = threadIdx.x >> 6; // 64 chunks of 16 B x 4 row quads
    const int x0 = xb0 + lc * 16;
    int imin = 255, imax = 0;
    // region-limited launches cut tiles at 16-byte chunks: a lane produces its chunk (store, min / max) only when the
    // chunk is inside the launch's region (rows are cut at tile boundaries); the whole-frame launch has everything inside
    const bool in_rg = y0 >= rg.y0 && y0 < rg.y1 && x0 >= rg.b0 && x0 < rg.b1;
    const bool mine = rg.invert ? !in_rg : in_rg;
    if (PACKED && x0 < wb && mine) {
        const int nbytes = min(16, wb - x0);
        // wave-uniform (a scalar branch, not per-lane selects): every lane of the wave owns 16 output bytes; false only
        // for the wave that holds the row's last chunk
        const bool full = __builtin_amdgcn_ballot_w64(nbytes != 16) == 0;
        // the kernel's two integers in every 16-bit lane
        const int ai = (int)a, cai = (int)ca;
        const shb_s2 A2 = {(short)ai, (short)ai};
        const shb_u2 C2 = {(unsigned short)cai, (unsigned short)cai};
        shb_u2 h[3][8];  // horizontal 3-sums of the last three staged rows, output bytes (2m, 2m + 1) in one register
        shb_u2 cc[2][8]; // centre bytes of the last two staged rows
        shb_s2 vmin = {32767, 32767}, vmax = {-32768, -32768};
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const int ly = rq * 4 + r; // staged row index (tile row 0 = y0 - 1)
            const u8* tr = &tile[ly * SHB_PITCH + 16 + lc * 16];
            u32 d[6];
            d[0] = *(const u32*)(tr - 4);
            const uint4 m = *(const uint4*)tr;
            d[1] = m.x;
            d[2] = m.y;
            d[3] = m.z;
            d[4] = m.w;
            d[5] = *(const u32*)(tr + 16);
            // window element k (byte x0 - 3 + k) is byte k + 1 of d[].  P[p] = elements (2p, 2p + 1), Q[p] = elements
            // (2p + 1, 2p + 2), zero-extended to 16 bits by ONE v_perm_b32 each
            auto pair_at = [&](int e) { // elements (e, e + 1) -> bytes (e + 1, e + 2) of d
                const int b0 = e + 1, b1 = e + 2;
                const int w0 = b0 >> 2, w1 = b1 >> 2;
                // selector bytes: result byte 0 = first element, byte 2 = second element, bytes 1 and 3 = 0
                const u32 sel = (u32)((b0 & 3) | (w0 == w1 ? 0 : 0)) | (0x0cu << 8) | ((u32)(((b1 & 3) + (w1 != w0 ? 4 : 0))) << 16) | (0x0cu << 24);
                return __builtin_bit_cast(shb_u2, __builtin_amdgcn_perm(d[w1], d[w0], sel));
            };
#pragma unroll
            for (int m2 = 0; m2 < 8; m2++) {
                const int t = 2 * m2; // output bytes (t, t + 1): window elements t, t + 3, t + 6
                const shb_u2 left = pair_at(t), mid = pair_at(t + 3), right = pair_at(t + 6);
                h[r % 3][m2] = (left + mid) + right;
                cc[r & 1][m2] = mid;
            }
            if (r >= 2) {
                const int yo = y0 + ly - 2; // centre row of the three last staged rows
                if (yo < g.h) {
                    u32 o[4];
#pragma unroll
                    for (int m2 = 0; m2 < 8; m2++) {
                        const shb_u2 S = (h[0][m2] + h[1][m2]) + h[2][m2];
                        const shb_u2 t2 = cc[(r - 1) & 1][m2] * C2;
                        const shb_s2 v = __builtin_bit_cast(shb_s2, S) * A2 + __builtin_bit_cast(shb_s2, t2);
                        u32 pk;
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(pk) : "v"(v)); // clamp both lanes to [0, 255], bytes 0 and 1
                        if (m2 & 1) o[m2 >> 1] = __builtin_amdgcn_perm(pk, o[m2 >> 1], 0x05040100u); // bytes: lo.0 lo.1 hi.0 hi.1
                        else o[m2 >> 1] = pk;
                        if (full) { // all 16 bytes are outputs (every chunk but the row's last): no per-byte masks
                            vmin = __builtin_elementwise_min(vmin, v);
                            vmax = __builtin_elementwise_max(vmax, v);
                        }
                    }
                    u8* q = df + (size_t)yo * g.stride + x0;
                    if (full && al16) *(uint4*)q = make_uint4(o[0], o[1], o[2], o[3]);
                    else {
                        for (int k = 0; k < nbytes; k++) {
                            const int bv = (int)((o[k >> 2] >> ((k & 3) * 8)) & 255u);
                            q[k] = (u8)bv;
                            if (!full) {
                                imin = min(imin, bv);
                                imax = max(imax, bv);
                            }
                        }
                    }
                }
            }
        }
        // min / max of the SATURATED outputs = the clamped min / max of the unsaturated values
        if (full && vmin.x <= vmax.x) {
            imin = min(max((int)min(vmin.x, vmin.y), 0), 255);
            imax = min(max((int)max(vmax.x, vmax.y), 0), 255);
        }
    }
    if (!PACKED && x0 < wb && mine) {
        const int nbytes = min(16, wb - x0);
        float h[3][16]; // horizontal 3-sums of the last three staged rows
        float c[2][16]; // centre values of the last two staged rows
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const int ly = rq * 4 + r; // staged row index (tile row 0 = y0 - 1)
            const u8* tr = &tile[ly * SHB_PITCH + 16 + lc * 16];
            u32 d[6];
            d[0] = *(const u32*)(tr - 4);
            const uint4 m = *(const uint4*)tr;
            d[1] = m.x;
            d[2] = m.y;
            d[3] = m.z;
            d[4] = m.w;
            d[5] = *(const u32*)(tr + 16);
            float f[22]; // window bytes x0-3 .. x0+18
#pragma unroll
            for (int k = 0; k < 22; k++) f[k] = (float)((d[(k + 1) >> 2] >> (((k + 1) & 3) * 8)) & 255u);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                h[r % 3][k] = (f[k] + f[k + 3]) + f[k + 6];
                c[r & 1][k] = f[k + 3];
            }
            if (r >= 2) {
                const int yo = y0 + ly - 2; // centre row of the three last staged rows
                if (yo < g.h) {
                    u32 o[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const float S = (h[0][k] + h[1][k]) + h[2][k];
                        const float v = __fmaf_rn(ca, c[(r - 1) & 1][k], a * S);
                        o[k >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(v, k & 3, o[k >> 2]);
                    }
                    // min/max of the saturated bytes, in integers (byte operands are free with SDWA; f32
                    // min/max would pay a canonicalisation each)
                    if (nbytes == 16) {
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const int bv = (int)((o[k >> 2] >> ((k & 3) * 8)) & 255u);
                            imin = min(imin, bv);
                            imax = max(imax, bv);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; k++)
                            if (k < nbytes) {
                                const int bv = (int)((o[k >> 2] >> ((k & 3) * 8)) & 255u);
                                imin = min(imin, bv);
                                imax = max(imax, bv);
                            }
                    }
                    u8* q = df + (size_t)yo * g.stride + x0;
                    if (nbytes == 16 && al16) *(uint4*)q = make_uint4(o[0], o[1], o[2], o[3]);
                    else
                        for (int k = 0; k < nbytes; k++) q[k] = (u8)(o[k >> 2] >> ((k & 3) * 8));
                }
            }
        }
    }
    const int mn = wave_min_i32(imin), mx = wave_max_i32(imax);
    if ((threadIdx.x & 63) == 0) {
        red[rq] = mn;
        red[4 + rq] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32* mm = aux + (size_t)blockIdx.z * aux_words(tiles_total) + (size_t)tiles_total * 256;
        atomicMin(&mm[0], (u32)min(min(red[0], red[1]), min(red[2], red[3])));
        atomicMax(&mm[1], (u32)max(max(red[4], red[5]), max(red[6], red[7])));
    }
#ifdef SH_TIMING
    SH_STAMP(3);
    if (threadIdx.x == 0) {
        const unsigned int k = atomicAdd(&g_sh_count, 1u);
        if (k < SH_MAX_STAMPS) {
            unsigned int hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            g_sh_stamps[k * 5 + 0] = hw;
            for (int q = 0; q < 4; q++) g_sh_stamps[k * 5 + 1 + q] = sh_t[q];
        }
    }
#endif
}

#ifdef SH_TIMING
extern "C" __attribute__((visibility("default"))) int cbv_debug_sharpen_stamps(unsigned long long* out, int cap, int reset)
{
    unsigned int n = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_sh_count), sizeof(n));
    if (n > SH_MAX_STAMPS) n = SH_MAX_STAMPS;
    if ((int)n > cap) n = cap;
    if (out && n) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sh_stamps), (size_t)n * 5 * sizeof(unsigned long long));
    if (reset) {
        unsigned int z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sh_count), &z, sizeof(z));
    }
    return (int)n;
}
#endif

// a * box + (c - a) * centre with small integers?  (Only that form has a region-limited launch.)
bool sharpen_region_ok(const float* k)
{
    const float a = k[0], c = k[4];
    for (int i = 0; i < 9; i++)
        if (i != 4 && k[i] != a) return false;
    return !(a != (float)(int)a || c != (float)(int)c || fabsf(a) > 64.f || fabsf(c) > 64.f);
}

// the box kernel's region for the pixels of `need`: bytes cut at 16-byte chunks (a lane's unit), rows at tile rows
static ShRegion sharpen_region(PxRect need, int invert)
{
    ShRegion r;
    r.b0 = need.x0 * 3 / 16 * 16;
    r.b1 = (need.x1 * 3 + 15) / 16 * 16;
    r.y0 = need.y0 / SHB_TH * SHB_TH;
    r.y1 = (need.y1 + SHB_TH - 1) / SHB_TH * SHB_TH;
    r.invert = invert;
    return r;
}

PxRect sharpen_region_cover(Geom g, PxRect need)
{
    const ShRegion r = sharpen_region(need, 0);
    // whole pixels the chunks touch (a pixel cut by a chunk edge is written by both neighbours)
    PxRect c = {r.b0 / 3, r.y0, (r.b1 + 2) / 3, r.y1};
    c.x1 = c.x1 > g.w ? g.w : c.x1;
    c.y1 = c.y1 > g.h ? g.h : c.y1;
    return c;
}

int launch_sharpen(cbv_ctx* ctx, const u8* src, u8* dst, u32* aux, int tiles, Geom g, const float* k, int batch, const EnhanceRegion* er)
{
    const float a = k[0], c = k[4];
    const bool box = sharpen_region_ok(k);
    if (er && !box) return cbv_fail(ctx, CBV_ERR_STATE, "launch_sharpen: region-limited launch of a kernel that is not box-shaped");
    prof_begin(ctx, CBV_K_SHARPEN);
    if (box) {
        const int txn = (g.w * 3 + SHB_TB - 1) / SHB_TB, tyn = (g.h + SHB_TH - 1) / SHB_TH;
        TileSet ts = tileset_make(txn, tyn, 0, 0, txn, tyn, false);
        SatGate gate = {nullptr, 0};
        ShRegion rg = {0, 0x7fffffff, 0, 0x7fffffff, 0};
        if (er) {
            rg = sharpen_region(er->px, er->invert);
            // region pass: every tile the region touches; complement pass: every tile not wholly inside it (the tiles
            // the region's left and right edges cut are visited by both passes, each lane by exactly one)
            if (!er->invert)
                ts = tileset_make(txn, tyn, rg.b0 / SHB_TB, rg.y0 / SHB_TH, (rg.b1 + SHB_TB - 1) / SHB_TB, rg.y1 / SHB_TH, false);
            else
                ts = tileset_make(txn, tyn, (rg.b0 + SHB_TB - 1) / SHB_TB, rg.y0 / SHB_TH, rg.b1 / SHB_TB, rg.y1 / SHB_TH, true);
            gate = er->gate;
        }
        const int nt = ts.cum[ts.n];
        const bool packed = fabsf(a) * 2295.f + fabsf(c - a) * 255.f <= 32767.f && (c - a) >= 0.f; // centre weight as u16
        if (nt > 0 && packed)
            hipLaunchKernelGGL(k_sharpen_box<true>, dim3(nt, 1, batch), dim3(256), 0, ctx->stream, src, dst, aux, tiles, g, a,
                               c - a, ts, gate, rg);
        else if (nt > 0)
            hipLaunchKernelGGL(k_sharpen_box<false>, dim3(nt, 1, batch), dim3(256), 0, ctx->stream, src, dst, aux, tiles, g, a,
                               c - a, ts, gate, rg);
    } else {
        int txn = (g.w + SH_TW - 1) / SH_TW, tyn = (g.h + SH_TH - 1) / SH_TH;
        hipLaunchKernelGGL(k_sharpen, dim3(txn * tyn, 1, batch), dim3(256), 0, ctx->stream, src, dst, aux, tiles, g, k[0],
                           k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], txn, txn * tyn);
    }
    prof_end(ctx, CBV_K_SHARPEN);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// NORM_MINMAX as a byte map
// ---------------------------------------------------------------------------
__global__ void k_norm_lut(const u32* __restrict__ aux, int tiles_total, u8* __restrict__ norm_lut)
{
    const u32* mm = aux + (size_t)blockIdx.x * aux_words(tiles_total) + (size_t)tiles_total * 256;
    norm_lut[(size_t)blockIdx.x * 256 + threadIdx.x] = d_norm_lut_entry((int)mm[0], (int)mm[1], (int)threadIdx.x);
}

int launch_norm_lut(cbv_ctx* ctx, const u32* aux, int tiles, u8* norm_lut, int batch)
{
    prof_begin(ctx, CBV_K_NORM_LUT);
    hipLaunchKernelGGL(k_norm_lut, dim3(batch), dim3(256), 0, ctx->stream, aux, tiles, norm_lut);
    prof_end(ctx, CBV_K_NORM_LUT);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// rows are mapped 16 bytes per lane
__global__ __launch_bounds__(256) void k_normalize(const u8* __restrict__ src, u8* __restrict__ dst,
                                                    const u8* __restrict__ norm_lut, const u32* __restrict__ minmax, size_t mm_stride,
                                                    Geom g, int blocks_per_frame)
{
    __shared__ u8 lut[256];
    if (minmax) { // NormSrc::minmax: the table from the frame's extremes, worked out by every workgroup
        const u32* mm = minmax + (size_t)blockIdx.z * mm_stride;
        lut[threadIdx.x] = d_norm_lut_entry((int)mm[0], (int)mm[1], (int)threadIdx.x);
    } else lut[threadIdx.x] = norm_lut[(size_t)blockIdx.z * 256 + threadIdx.x];
    __syncthreads();
    const size_t fo = (size_t)blockIdx.z * g.frame_stride;
    const int wb = g.w * 3;
    const bool flat = (g.stride == wb) && ((g.frame_stride & 15) == 0);
    if (flat) {
        const size_t total = (size_t)wb * g.h;
        const size_t nvec = total / 16;
        const uint4* s = (const uint4*)(src + fo);
        uint4* d = (uint4*)(dst + fo);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)blocks_per_frame * blockDim.x) {
            uint4 v = s[i];
            u32 w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 x = w4[k];
                w4[k] = (u32)lut[x & 255] | ((u32)lut[(x >> 8) & 255] << 8) | ((u32)lut[(x >> 16) & 255] << 16) |
                        ((u32)lut[x >> 24] << 24);
            }
            d[i] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
        if (blockIdx.x == 0)
            for (size_t i = nvec * 16 + threadIdx.x; i < total; i += blockDim.x) dst[fo + i] = lut[src[fo + i]];
    } else {
        for (int y = blockIdx.x; y < g.h; y += blocks_per_frame)
            for (int x = threadIdx.x; x < wb; x += blockDim.x)
                dst[fo + (size_t)y * g.stride + x] = lut[src[fo + (size_t)y * g.stride + x]];
    }
}

int launch_normalize(cbv_ctx* ctx, const u8* src, u8* dst, NormSrc norm, Geom g, int batch)
{
    size_t nvec = (size_t)g.w * 3 * g.h / 16;
    int blocks = (int)((nvec + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    prof_begin(ctx, CBV_K_NORMALIZE);
    hipLaunchKernelGGL(k_normalize, dim3(blocks, 1, batch), dim3(256), 0, ctx->stream, src, dst, norm.lut, norm.minmax, norm.mm_stride, g, blocks);
    prof_end(ctx, CBV_K_NORMALIZE);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// stand-alone global min/max (normalize_intensity called on its own)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax(const u8* __restrict__ src, u32* __restrict__ aux, int tiles_total, Geom g)
{
    __shared__ int red[8];
    const size_t fo = (size_t)blockIdx.z * g.frame_stride;
    const int wb = g.w * 3;
    int mn = 255, mx = 0;
    for (int y = blockIdx.x; y < g.h; y += gridDim.x) {
        const u8* row = src + fo + (size_t)y * g.stride;
        for (int x = threadIdx.x; x < wb; x += blockDim.x) {
            const int v = row[x];
            mn = min(mn, v);
            mx = max(mx, v);
        }
    }
    mn = wave_min_i32(mn);
    mx = wave_max_i32(mx);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = mn;
        red[4 + (threadIdx.x >> 6)] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32* mm = aux + (size_t)blockIdx.z * aux_words(tiles_total) + (size_t)tiles_total * 256;
        atomicMin(&mm[0], (u32)min(min(red[0], red[1]), min(red[2], red[3])));
        atomicMax(&mm[1], (u32)max(max(red[4], red[5]), max(red[6], red[7])));
    }
}

int launch_minmax(cbv_ctx* ctx, const u8* src, u32* aux, int tiles, Geom g, int batch)
{
    int blocks = g.h < 512 ? g.h : 512;
    hipLaunchKernelGGL(k_minmax, dim3(blocks, 1, batch), dim3(256), 0, ctx->stream, src, aux, tiles, g);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}
