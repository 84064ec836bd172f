// cv2.bilateralFilter(d, sigmaColor, sigmaSpace) on 8UC3 (frame_enhancer.py:131).
//
// VALU-bound stencil (49 taps at d = 9), not an HBM-bound one.  Layout:
//   - a 64 x 32 pixel tile (+ radius halo, REFLECT_101 at the image border) is
//     staged in LDS as packed BGRx dwords, so |db|+|dg|+|dr| is ONE v_sad_u8;
//   - each lane owns a 4-pixel horizontal strip on two rows; for every tap row
//     it pulls 4+2R packed pixels with aligned ds_read_b128 and converts them
//     to float once for all four outputs;
//   - colour weights come from a 768-entry float LUT in LDS, space weights are
//     wave-uniform scalars;
//   - taps are accumulated per output in row-major (dy, dx) order with separate
//     multiply and add (no FMA contraction), which makes the result bit-equal to
//     the scalar definition of the filter.
#include "cbv_device.h"

#define BL_TW 64
#define BL_TH 32

template <int R>
struct BlCfg {
    static constexpr int HALO_X = 4;                       // halo rounded up to 4 px so b128 reads stay aligned
    static constexpr int PITCH = BL_TW + 2 * HALO_X;       // pixels (dwords) per LDS row
    static constexpr int ROWS = BL_TH + 2 * R;
    static_assert(R <= 4, "radius above 4 needs a wider halo");
};

template <int R>
__global__ __launch_bounds__(256) void k_bilateral(const u8* __restrict__ src, u8* __restrict__ dst, Geom g,
                                                    const BilateralTabs* __restrict__ bt, int tiles_xn, int tiles_n)
{
    using C = BlCfg<R>;
    __shared__ __attribute__((aligned(16))) u32 tile[C::ROWS * C::PITCH];
    __shared__ float cw[768];

    const int tid = xcd_remap(blockIdx.x, tiles_n);
    const int tyi = tid / tiles_xn, txi = tid - tyi * tiles_xn;
    const int x0 = txi * BL_TW, y0 = tyi * BL_TH;
    const size_t fo = (size_t)blockIdx.z * g.frame_stride;
    const u8* sf = src + fo;
    u8* df = dst + fo;

    for (int i = threadIdx.x; i < 768; i += blockDim.x) cw[i] = bt->color_w[i];

    // stage: groups of 4 pixels (12 source bytes -> 4 packed dwords)
    const bool aligned = (g.stride & 3) == 0;
    constexpr int GROUPS = C::PITCH / 4;
    for (int i = threadIdx.x; i < C::ROWS * GROUPS; i += blockDim.x) {
        const int r = i / GROUPS, gi = i - r * GROUPS;
        const int sy = d_reflect101(y0 - R + r, g.h);
        const int gx = x0 - C::HALO_X + gi * 4;
        u32 p0, p1, p2, p3;
        if (aligned && gx >= 0 && gx + 3 < g.w) {
            const u32* p = (const u32*)(sf + (size_t)sy * g.stride + (size_t)gx * 3);
            u32 d0 = p[0], d1 = p[1], d2 = p[2];
            p0 = d0 & 0xFFFFFFu;
            p1 = (d0 >> 24) | ((d1 & 0xFFFFu) << 8);
            p2 = (d1 >> 16) | ((d2 & 0xFFu) << 16);
            p3 = d2 >> 8;
        } else {
            u32 pp[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sx = d_reflect101(gx + k, g.w);
                const u8* p = sf + (size_t)sy * g.stride + (size_t)sx * 3;
                pp[k] = (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16);
            }
            p0 = pp[0];
            p1 = pp[1];
            p2 = pp[2];
            p3 = pp[3];
        }
        *(uint4*)&tile[r * C::PITCH + gi * 4] = make_uint4(p0, p1, p2, p3);
    }
    __syncthreads();

    const int sx = threadIdx.x & 15; // strip index: pixels 4*sx .. 4*sx+3
    const int sy = threadIdx.x >> 4; // rows sy and sy + 16
    const bool aligned_out = aligned;

#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        const int ly = sy + half * 16;
        const int y = y0 + ly;
        const int x = x0 + sx * 4;
        float sb[4], sg[4], sr[4], sw[4];
        u32 ctr[4];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            sb[o] = sg[o] = sr[o] = sw[o] = 0.f;
            ctr[o] = tile[(ly + R) * C::PITCH + C::HALO_X + sx * 4 + o];
        }
        int k = 0; // running tap index (row-major over the disc), compile-time after unrolling
#pragma unroll
        for (int dy = -R; dy <= R; dy++) {
            // packed pixels x-4 .. x+7 of tap row
            const uint4* rowp = (const uint4*)&tile[(ly + R + dy) * C::PITCH + sx * 4];
            const uint4 q0 = rowp[0], q1 = rowp[1], q2 = rowp[2];
            const u32 p[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
            float fb[12], fg[12], fr[12];
#pragma unroll
            for (int j = 0; j < 12; j++) {
                fb[j] = (float)(p[j] & 255u);
                fg[j] = (float)((p[j] >> 8) & 255u);
                fr[j] = (float)((p[j] >> 16) & 255u);
            }
            // taps of this row: |dx| <= floor(sqrt(R^2 - dy^2))  (r = sqrt(i*i + j*j) <= radius)
            int rx = 0;
            while ((rx + 1) * (rx + 1) + dy * dy <= R * R) rx++;
#pragma unroll
            for (int dx = -R; dx <= R; dx++) {
                if (dx < -rx || dx > rx) continue;
                const float spw = bt->space_w[k];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int j = o + 4 + dx;
                    const u32 sad = __builtin_amdgcn_sad_u8(p[j], ctr[o], 0u);
                    const float wgt = spw * cw[sad];
                    const float tb = fb[j] * wgt, tg = fg[j] * wgt, tr = fr[j] * wgt;
                    sb[o] = sb[o] + tb;
                    sg[o] = sg[o] + tg;
                    sr[o] = sr[o] + tr;
                    sw[o] = sw[o] + wgt;
                }
                k++;
            }
        }
        if (y < g.h && x < g.w) {
            Px4 out;
            out.d[0] = out.d[1] = out.d[2] = 0;
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const float inv = 1.f / sw[o];
                px_set(out, 3 * o, d_round_f(sb[o] * inv));
                px_set(out, 3 * o + 1, d_round_f(sg[o] * inv));
                px_set(out, 3 * o + 2, d_round_f(sr[o] * inv));
            }
            u8* q = df + (size_t)y * g.stride + (size_t)x * 3;
            const int npx = min(4, g.w - x);
            if (aligned_out && npx == 4) {
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

template <int R>
static int launch_bilateral_r(cbv_ctx* ctx, const u8* src, u8* dst, Geom g, int batch)
{
    int txn = (g.w + BL_TW - 1) / BL_TW, tyn = (g.h + BL_TH - 1) / BL_TH;
    prof_begin(ctx, CBV_K_BILATERAL);
    hipLaunchKernelGGL(k_bilateral<R>, dim3(txn * tyn, 1, batch), dim3(256), 0, ctx->stream, src, dst, g, ctx->btabs,
                       txn, txn * tyn);
    prof_end(ctx, CBV_K_BILATERAL);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

int launch_bilateral(cbv_ctx* ctx, const u8* src, u8* dst, Geom g, int batch)
{
    switch (ctx->btabs_host.radius) {
    case 1: return launch_bilateral_r<1>(ctx, src, dst, g, batch);
    case 2: return launch_bilateral_r<2>(ctx, src, dst, g, batch);
    case 3: return launch_bilateral_r<3>(ctx, src, dst, g, batch);
    case 4: return launch_bilateral_r<4>(ctx, src, dst, g, batch);
    default: return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "bilateral radius %d not supported (d <= 9)", ctx->btabs_host.radius);
    }
}
