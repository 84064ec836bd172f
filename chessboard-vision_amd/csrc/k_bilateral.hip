// cv2.bilateralFilter(d, sigmaColor, sigmaSpace) on 8UC3 (frame_enhancer.py:131).
//
// VALU/LDS-bound stencil (49 taps at d = 9), not an HBM-bound one.  Structure:
//   - persistent workgroups of 768 lanes (12 waves = 3 per SIMD, <= 128 VGPRs),
//     one per CU, each walking 128x48-pixel tiles; consecutive tiles of one XCD
//     are neighbours, so halo rows are re-read from that XCD's L2.  Three waves
//     per SIMD, not four, on purpose: see launch_bilateral;
//   - the tile (+4 px halo, REFLECT_101 at the image border) sits in LDS as
//     packed BGRx dwords: |db|+|dg|+|dr| is ONE v_sad_u8; the next tile is
//     prefetched into registers while the current one is filtered;
//   - the tap weight w = space * colour is ONE table lookup: taps with the same
//     dy^2 + dx^2 share their space weight, so there are only 10 distinct space
//     weights at d = 9, and folded[class][sad] (10 x 766 floats, 30 KB, built on
//     the host with the same single float multiply) replaces the colour-table
//     lookup AND the multiply.  The class offset rides in v_sad_u8's accumulator
//     operand (an SGPR), the byte address is one full-rate shift: per tap and
//     output 6 vector ops (sad, lshl, 3 fma, add) instead of 8.
//     The table is NOT replicated per bank any more (round 1 kept 32 copies of
//     the colour table, 98 KB, to make the gather conflict-free): lanes of a
//     half-wave read neighbouring pixels of one row at one tap, whose colour
//     distances are close, identical addresses broadcast, and addresses less
//     than 32 entries apart fall into different banks, so conflicts only arise
//     between lanes whose distances differ by a multiple of 32
//     (profiles/r02/sq_counters.txt has the measured conflict rate);
//   - each lane owns a 4-pixel strip on TWO adjacent rows; per tap row it reads
//     12 packed pixels with three aligned ds_read_b128 and converts them to
//     float once for all eight outputs (v_cvt_f32_ubyte is a half-rate op on
//     gfx950, like v_sad_u8; only f32 add/mul and simple integer ops are full
//     rate — tools/ubench_valu.hip);
//   - taps are accumulated per output in row-major (dy, dx) order exactly like
//     OpenCV's FMA3-dispatched body: w = space * colour (one rounding, here done
//     when the table is built), sum = fma(px, w, sum), wsum += w: bit-equal to
//     the oracle.
#include "cbv_device.h"

#define BL_TW 128        // tile width in pixels (32 strips of 4)
#define BL_HALO 4        // halo in pixels (radius <= 4; 4 keeps ds_read_b128 aligned)
#define BL_PITCH (BL_TW + 2 * BL_HALO)
#define BL_LUT_WORDS (CBV_BL_MAXCLS * 768)

__host__ __device__ constexpr int bl_row_reach(int R, int dy)
{
    int rx = 0;
    while ((rx + 1) * (rx + 1) + dy * dy <= R * R) rx++;
    return rx;
}

// NT lanes per workgroup (a multiple of 64): 32 strips x NT / 32 row pairs, i.e. tiles of 128 x NT / 16 pixels
template <int R, int NT>
__global__ __launch_bounds__(NT) void k_bilateral(const u8* __restrict__ src, u8* __restrict__ dst, Geom g,
                                                           const BilateralTabs* __restrict__ bt, TileSet ts, int batch,
                                                           SatGate gate)
{
    static_assert((2 * R + 1) * (2 * R + 1) < 128, "the epilogue's d_rcp_1_64 is verified for weight sums in [1, 128)");
    // static LDS (69 KB): compile-time addresses let the table gather use the ds_read immediate offset
#ifdef BL_TWO_COPIES
    // experiment (profiles/r03/bilateral_two_copies.txt): a second copy of the table 16 banks further on, used by the odd
    // strips, so that two lanes of a half-wave whose distances differ by a multiple of 32 no longer meet in one bank
    __shared__ __attribute__((aligned(16))) float fw[2 * BL_LUT_WORDS + 16];
#else
    __shared__ __attribute__((aligned(16))) float fw[BL_LUT_WORDS];              // [class][768] folded weights
#endif
    constexpr int BL_TH = NT / 16, BL_THREADS = NT;
    __shared__ __attribute__((aligned(16))) u32 tile[(BL_TH + 2 * R) * BL_PITCH];
    constexpr int ROWS = BL_TH + 2 * R;
    constexpr int GROUPS = BL_PITCH / 4;                        // 4-pixel groups per tile row
    constexpr int NG = ROWS * GROUPS;                           // groups per tile (<= 2448)
    constexpr int GPT = (NG + BL_THREADS - 1) / BL_THREADS;     // groups per thread (3)

    const int tid = threadIdx.x;
    for (int i = tid; i < bt->ncls * 768; i += BL_THREADS) fw[i] = (&bt->folded[0][0])[i];
#ifdef BL_TWO_COPIES
    for (int i = tid; i < bt->ncls * 768; i += BL_THREADS) fw[BL_LUT_WORDS + 16 + i] = (&bt->folded[0][0])[i];
    const u32 copy_off = (tid & 1) ? (u32)(BL_LUT_WORDS + 16) : 0u;
#endif

    const int tiles_per_frame = ts.cum[ts.n]; // the tiles of this launch's region (TileSet), not of the whole frame
    const int ntiles = tiles_per_frame * batch;
    const bool aligned = (g.stride & 3) == 0;

    // Interior 4-pixel groups are prefetched as 3 raw dwords; groups that touch the image border
    // (REFLECT_101) are rare and are fetched synchronously, byte-wise, when the tile is written.
    // a tile: frame and pixel origin (wave-uniform; worked out once per tile)
    struct TileAt {
        int f, px0, py0;
    };
    auto tile_at = [&](int t) {
        TileAt a;
        a.f = t / tiles_per_frame;
        int tyi, txi;
        tileset_at(ts, t - a.f * tiles_per_frame, txi, tyi);
        a.px0 = txi * BL_TW;
        a.py0 = tyi * BL_TH;
        return a;
    };
    auto group_origin = [&](const TileAt& ta, int gi, int& f, int& y, int& gx) {
        f = ta.f;
        const int r = gi / GROUPS, gc = gi - r * GROUPS;
        y = ta.py0 - R + r;
        gx = ta.px0 - BL_HALO + gc * 4;
    };
    auto prefetch = [&](const TileAt& t, int gi, u32* d) {
        int f, y, gx;
        group_origin(t, gi, f, y, gx);
        if (aligned && gx >= 0 && gx + 3 < g.w && y >= 0 && y < g.h) {
            const u32* q = (const u32*)(src + (size_t)f * g.frame_stride + (size_t)y * g.stride + (size_t)gx * 3);
            d[0] = q[0];
            d[1] = q[1];
            d[2] = q[2];
        }
    };
    auto commit = [&](const TileAt& t, int gi, const u32* d) {
        int f, y, gx;
        group_origin(t, gi, f, y, gx);
        u32 p0, p1, p2, p3;
        if (aligned && gx >= 0 && gx + 3 < g.w && y >= 0 && y < g.h) {
            p0 = d[0] & 0xFFFFFFu;
            p1 = (d[0] >> 24) | ((d[1] & 0xFFFFu) << 8);
            p2 = (d[1] >> 16) | ((d[2] & 0xFFu) << 16);
            p3 = d[2] >> 8;
        } else {
            const u8* sf = src + (size_t)f * g.frame_stride + (size_t)d_reflect101(y, g.h) * g.stride;
            u32 pp[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u8* q = sf + (size_t)d_reflect101(gx + k, g.w) * 3;
                pp[k] = (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16);
            }
            p0 = pp[0];
            p1 = pp[1];
            p2 = pp[2];
            p3 = pp[3];
        }
        const int r = gi / GROUPS, gc = gi - r * GROUPS;
        *(uint4*)&tile[r * BL_PITCH + gc * 4] = make_uint4(p0, p1, p2, p3);
    };

    // persistent walk: sequence number s = i * gridDim + block; XCD x owns a contiguous tile range.  Tiles of frames
    // whose gate is closed (complement pass of a frame whose region already holds 0 and 255) are stepped over.
    auto next_open = [&](int s) {
        while (s < ntiles && sat_gate_closed(gate, xcd_remap(s, ntiles) / tiles_per_frame)) s += gridDim.x;
        return s;
    };
    int s = next_open(blockIdx.x);
    u32 pre[GPT][3];
    TileAt cur = {0, 0, 0};
    if (s < ntiles) {
        cur = tile_at(xcd_remap(s, ntiles));
#pragma unroll
        for (int k = 0; k < GPT; k++) {
            const int gi = tid + k * BL_THREADS;
            if (gi < NG) prefetch(cur, gi, pre[k]);
        }
    }
    const int sx = tid & 31;  // strip: pixels 4*sx .. 4*sx+3
    const int ly = (tid >> 5) * 2; // first of the lane's two tile rows

    while (s < ntiles) {
        const TileAt here = cur;
        __syncthreads(); // previous tile fully consumed (and the LUT is written on the first pass)
#pragma unroll
        for (int k = 0; k < GPT; k++) {
            const int gi = tid + k * BL_THREADS;
            if (gi < NG) commit(here, gi, pre[k]);
        }
        __syncthreads();
        // prefetch the next tile while this one is filtered
        s = next_open(s + gridDim.x);
        if (s < ntiles) {
            cur = tile_at(xcd_remap(s, ntiles));
#pragma unroll
            for (int k = 0; k < GPT; k++) {
                const int gi = tid + k * BL_THREADS;
                if (gi < NG) prefetch(cur, gi, pre[k]);
            }
        }

        // outputs: [row a/b][pixel 0..3]
        float sb[2][4], sg[2][4], sr[2][4], sw[2][4];
        u32 ctr[2][4];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int o = 0; o < 4; o++) {
                sb[a][o] = sg[a][o] = sr[a][o] = sw[a][o] = 0.f;
                ctr[a][o] = tile[(ly + a + R) * BL_PITCH + BL_HALO + sx * 4 + o];
            }
        // One tile row per iteration, NOT unrolled (a fully unrolled body makes hipcc schedule ~250
        // live registers).  Tile row ly + i is tap row dy = i - R of output row a and dy = i - 1 - R of
        // output row b; every output still sees its taps in ascending (dy, dx) order.
        const u32* rowp0 = &tile[ly * BL_PITCH + sx * 4];
#pragma unroll 1
        for (int i = 0; i <= 2 * R + 1; i++) {
            const uint4* rowp = (const uint4*)(rowp0 + i * BL_PITCH);
            const uint4 q0 = rowp[0], q1 = rowp[1], q2 = rowp[2];
            // table offsets of this row's taps for both output rows: wave-uniform scalar loads (-1 = outside the disc)
            int off[2][2 * R + 1];
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const int dyi = min(max(i - a, 0), 2 * R);
#pragma unroll
                for (int d = 0; d <= 2 * R; d++) off[a][d] = bt->tap_off[dyi][d];
            }
            const u32 p[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
            float fb[12], fg[12], fr[12];
#pragma unroll
            for (int j = 0; j < 12; j++) {
                fb[j] = (float)(p[j] & 255u);
                fg[j] = (float)((p[j] >> 8) & 255u);
                fr[j] = (float)((p[j] >> 16) & 255u);
            }
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const int dyi = i - a; // tap-row index of output row a (0 .. 2R), wave-uniform
                if (dyi < 0 || dyi > 2 * R) continue;
#pragma unroll
                for (int dx = -R; dx <= R; dx++) {
                    const int toff = off[a][dx + R];
                    if (toff < 0) continue; // scalar branch
#ifdef BL_TWO_COPIES
                    const u32 tacc = (u32)toff + copy_off; // one vector add per tap, shared by the four outputs of the row
#else
                    const u32 tacc = (u32)toff;
#endif
#pragma unroll
                    for (int o = 0; o < 4; o++) {
                        const int j = o + 4 + dx;
                        // word index class * 768 + |db| + |dg| + |dr| in one v_sad_u8 (the class offset is its accumulator)
                        const u32 idx = __builtin_amdgcn_sad_u8(p[j], ctr[a][o], tacc);
                        const float wgt = *(const float*)((const u8*)fw + (idx << 2));
                        // v_muladd(v_cvt_f32(b), w, sum_b): fused, like OpenCV's FMA3-dispatched body
                        sb[a][o] = __fmaf_rn(fb[j], wgt, sb[a][o]);
                        sg[a][o] = __fmaf_rn(fg[j], wgt, sg[a][o]);
                        sr[a][o] = __fmaf_rn(fr[j], wgt, sr[a][o]);
                        sw[a][o] = sw[a][o] + wgt;
                    }
                }
            }
        }
        const int f = here.f;
        const int x = here.px0 + sx * 4;
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int y = here.py0 + ly + a;
            if (y < g.h && x < g.w) {
                Px4 out;
                out.d[0] = out.d[1] = out.d[2] = 0;
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    // the weight sum lies in [1, 49]: the centre tap's weight is exactly 1.0 (space 1 x colour 1) and no
                    // weight exceeds 1, so the exact reciprocal of that interval applies
                    const float inv = d_rcp_1_64(sw[a][o]);
                    // v_cvt_pk_u8_f32: cvRound (half to even) + pack; the value is a weighted mean of bytes, already in range
                    out.d[(3 * o) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(sb[a][o] * inv, (3 * o) & 3, out.d[(3 * o) >> 2]);
                    out.d[(3 * o + 1) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(sg[a][o] * inv, (3 * o + 1) & 3, out.d[(3 * o + 1) >> 2]);
                    out.d[(3 * o + 2) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(sr[a][o] * inv, (3 * o + 2) & 3, out.d[(3 * o + 2) >> 2]);
                }
                u8* q = dst + (size_t)f * g.frame_stride + (size_t)y * g.stride + (size_t)x * 3;
                const int npx = min(4, g.w - x);
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
}

// NT and workgroups per CU for a launch: 768 lanes = 3 waves per SIMD at <= 128 VGPRs: alone the kernel is 11 % slower
// than with 1024 lanes (4 waves hide the LDS gather better), but a fourth wave slot and 160 VGPRs per SIMD stay free, so
// the other lane's kernels run BESIDE it instead of waiting for its persistent workgroups to end; the whole path gains
// 2-3 % (gpurun sweep, 1080p x 512: 640 lanes 26.0 k frames/s, 768 27.8 k, 896 25.5 k, 1024 27.2 k, 2 x 512 25.9 k).
// One or two frames (the live-camera case) are too few 128 x 48 tiles for 256 persistent workgroups (a 1080p frame has
// 345: the second round is a third full).  128 x 32 tiles on TWO 512-lane workgroups per CU (2 x 52 KB of LDS, the same
// 4 waves per SIMD as one 1024-lane workgroup) put 510 tiles on 512 workgroups in one round.
static int bilateral_nt(cbv_ctx* ctx, Geom g, int batch)
{
    const long long t768 = (long long)((g.w + BL_TW - 1) / BL_TW) * ((g.h + 768 / 16 - 1) / (768 / 16)) * batch;
    return t768 < 2ll * ctx->num_cus ? 512 : 768;
}

// tiles of the launch: the whole frame, or (region-limited enhancement) the tiles covering er->px / all the others
static TileSet bilateral_tiles(Geom g, int nt, const EnhanceRegion* er)
{
    const int th = nt / 16;
    const int txn = (g.w + BL_TW - 1) / BL_TW, tyn = (g.h + th - 1) / th;
    if (!er) return tileset_make(txn, tyn, 0, 0, txn, tyn, false);
    return tileset_make(txn, tyn, er->px.x0 / BL_TW, er->px.y0 / th, (er->px.x1 + BL_TW - 1) / BL_TW, (er->px.y1 + th - 1) / th, er->invert != 0);
}

PxRect bilateral_region_cover(cbv_ctx* ctx, Geom g, int batch, PxRect need)
{
    const int th = bilateral_nt(ctx, g, batch) / 16;
    PxRect c = {need.x0 / BL_TW * BL_TW, need.y0 / th * th, (need.x1 + BL_TW - 1) / BL_TW * BL_TW, (need.y1 + th - 1) / th * th};
    c.x1 = c.x1 > g.w ? g.w : c.x1;
    c.y1 = c.y1 > g.h ? g.h : c.y1;
    return c;
}

template <int R, int NT>
static int launch_bilateral_r(cbv_ctx* ctx, const u8* src, u8* dst, Geom g, int batch, int wgs_per_cu, const EnhanceRegion* er)
{
    const TileSet ts = bilateral_tiles(g, NT, er);
    const long long ntiles = (long long)ts.cum[ts.n] * batch;
    if (ntiles == 0) return CBV_OK;
    long long grid = (long long)ctx->num_cus * wgs_per_cu < ntiles ? (long long)ctx->num_cus * wgs_per_cu : ntiles; // persistent workgroups
    grid = (grid + 7) & ~7ll;                                 // whole XCD groups
    if (grid > ntiles) grid = ntiles;
    const SatGate gate = er ? er->gate : SatGate{nullptr, 0};
    prof_begin(ctx, CBV_K_BILATERAL);
    hipLaunchKernelGGL((k_bilateral<R, NT>), dim3((unsigned)grid), dim3(NT), 0, ctx->stream, src, dst, g, ctx->btabs, ts, batch, gate);
    prof_end(ctx, CBV_K_BILATERAL);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

int launch_bilateral(cbv_ctx* ctx, const u8* src, u8* dst, Geom g, int batch, const EnhanceRegion* er)
{
    const int R = ctx->btabs_host.radius;
    if (R < 1 || R > 4) return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "bilateral radius %d not supported (d <= 9)", R);
    if (bilateral_nt(ctx, g, batch) == 512) switch (R) {
        case 1: return launch_bilateral_r<1, 512>(ctx, src, dst, g, batch, 2, er);
        case 2: return launch_bilateral_r<2, 512>(ctx, src, dst, g, batch, 2, er);
        case 3: return launch_bilateral_r<3, 512>(ctx, src, dst, g, batch, 2, er);
        default: return launch_bilateral_r<4, 512>(ctx, src, dst, g, batch, 2, er);
        }
    switch (R) {
    case 1: return launch_bilateral_r<1, 768>(ctx, src, dst, g, batch, 1, er);
    case 2: return launch_bilateral_r<2, 768>(ctx, src, dst, g, batch, 1, er);
    case 3: return launch_bilateral_r<3, 768>(ctx, src, dst, g, batch, 1, er);
    default: return launch_bilateral_r<4, 768>(ctx, src, dst, g, batch, 1, er);
    }
}
