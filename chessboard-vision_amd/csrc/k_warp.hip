// cv2.warpPerspective(img, M, (S,S)) INTER_LINEAR / BORDER_CONSTANT(0)
// (board_detection.py:70), optional cv2.rotate(ROTATE_180) (game_session.py:126)
// and — for the fused pipeline — cv2.normalize folded into the gather (a byte
// map commutes with sampling: the four taps are mapped before interpolation).
//
// Coordinates follow WarpPerspectiveInvoker exactly: the destination is cut
// into 64 x 16 blocks (BLOCK_SZ = 32), X0/Y0/W0 are evaluated at the block's
// left edge and advanced by M*x1 inside the block, in double, one rounding per
// operation; coordinates are quantised to 1/32 px and sampled with the 15-bit
// fixed-point bilinear table of remap().
// Also here: the synthetic frame generator used by bench/tests.
#include "cbv_device.h"

struct WarpM {
    double m[9];
};

// FPT frames per thread: the coordinates of a destination pixel depend on the matrix and the pixel, not on the frame, and
// they are most of the kernel's instructions (about 100 of 190 per output pixel, in double precision): a thread works
// them out once and samples FPT consecutive frames of the batch with them (blockIdx.z counts groups of FPT frames).
// CALC: the byte map is worked out here from the frames' [min, max] words (NormSrc::minmax) instead of read from a table.
template <int FPT, bool CALC>
__global__ __launch_bounds__(256) void k_warp(const u8* __restrict__ src, Geom g, WarpM M, int dw, int dh, int bw0,
                                               int bh0, int rot180, u8* __restrict__ dst, int dst_stride,
                                               size_t dst_frame_stride, const u8* __restrict__ norm_lut, const u32* __restrict__ minmax,
                                               size_t mm_stride, u32* __restrict__ zero_word, u32* __restrict__ zero_word2, int batch)
{
    __shared__ u8 lut[FPT][256];
    // the pipeline's HoughCircles worklist counter, filled by the NEXT kernel in the stream (k_squares_pre5_stats):
    // zeroed here instead of by a 4-byte memset, which is one more ~4.5 us launch in a single-frame run
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        if (zero_word) *zero_word = 0u;
        if (zero_word2) *zero_word2 = 0u; // the run's second-pass list, when this launch is the run's only chunk
    }
    const bool use_lut = CALC || norm_lut != nullptr;
    const int f0 = blockIdx.z * FPT;
    const int nf = min(FPT, batch - f0);
    if (CALC) {
#pragma unroll
        for (int k = 0; k < FPT; k++)
            if (k < nf) {
                const u32* mm = minmax + (size_t)(f0 + k) * mm_stride;
                lut[k][threadIdx.x] = d_norm_lut_entry((int)mm[0], (int)mm[1], (int)threadIdx.x);
            }
    } else if (use_lut)
#pragma unroll
        for (int k = 0; k < FPT; k++)
            if (k < nf) lut[k][threadIdx.x] = norm_lut[(size_t)(f0 + k) * 256 + threadIdx.x];
    __syncthreads();
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const int bx = (dx / bw0) * bw0, x1 = dx - bx; // block origin and offset inside it
    (void)bh0;                                     // rows are evaluated independently of the block row
    const double X0 = M.m[0] * bx + M.m[1] * dy + M.m[2];
    const double Y0 = M.m[3] * bx + M.m[4] * dy + M.m[5];
    const double W0 = M.m[6] * bx + M.m[7] * dy + M.m[8];
    double W = W0 + M.m[6] * x1;
    W = W != 0. ? 32. / W : 0.;
    double fX = (X0 + M.m[0] * x1) * W;
    double fY = (Y0 + M.m[3] * x1) * W;
    fX = fmax(-2147483648.0, fmin(2147483647.0, fX));
    fY = fmax(-2147483648.0, fmin(2147483647.0, fY));
    const int X = d_round_d(fX), Y = d_round_d(fY);
    const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool any_in = !(sx >= g.w || sx + 1 < 0 || sy >= g.h || sy + 1 < 0);
    const bool x0in = sx >= 0 && sx < g.w, x1in = sx + 1 >= 0 && sx + 1 < g.w;
    const bool y0in = sy >= 0 && sy < g.h, y1in = sy + 1 >= 0 && sy + 1 < g.h;
    const bool interior = x0in && x1in && y0in && y1in;
    const size_t tap = (size_t)sy * g.stride + (size_t)sx * 3; // (only dereferenced where the taps are inside)
    const int ox = rot180 ? dw - 1 - dx : dx, oy = rot180 ? dh - 1 - dy : dy;
    const size_t out_off = (size_t)oy * dst_stride + (size_t)ox * 3;
    // interior: the two taps of a row are 6 contiguous bytes -> ONE unaligned 8-byte load per row instead of six byte
    // loads (the gather is bound by the number of memory instructions, not by bytes); the two bytes read past the second
    // tap stay inside the buffer (callers keep >= 8 bytes of slack behind the last frame).  All frames' loads are issued
    // before the first is used.
    u64 ta[FPT], tb[FPT];
    if (interior)
#pragma unroll
        for (int k = 0; k < FPT; k++)
            if (k < nf) {
                const u8* p00 = src + (size_t)(f0 + k) * g.frame_stride + tap;
                __builtin_memcpy(&ta[k], p00, 8);
                __builtin_memcpy(&tb[k], p00 + g.stride, 8);
            }
#pragma unroll
    for (int k = 0; k < FPT; k++) {
        if (k >= nf) break;
        int o[3] = {0, 0, 0};
        if (any_in) {
            int v[4][3]; // taps 00, 01, 10, 11
            if (interior) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    v[0][c] = (int)((ta[k] >> (8 * c)) & 255);
                    v[1][c] = (int)((ta[k] >> (8 * (3 + c))) & 255);
                    v[2][c] = (int)((tb[k] >> (8 * c)) & 255);
                    v[3][c] = (int)((tb[k] >> (8 * (3 + c))) & 255);
                }
            } else {
                const u8* p00 = src + (size_t)(f0 + k) * g.frame_stride + tap;
                const u8* p10 = p00 + g.stride;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    v[0][c] = (x0in && y0in) ? p00[c] : -1;
                    v[1][c] = (x1in && y0in) ? p00[3 + c] : -1;
                    v[2][c] = (x0in && y1in) ? p10[c] : -1;
                    v[3][c] = (x1in && y1in) ? p10[3 + c] : -1;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                int t[4];
#pragma unroll
                for (int q = 0; q < 4; q++) t[q] = v[q][c] < 0 ? 0 : (use_lut ? (int)lut[k][v[q][c]] : v[q][c]); // border taps are 0
                o[c] = d_sat8((t[0] * w00 + t[1] * w01 + t[2] * w10 + t[3] * w11 + (1 << 14)) >> 15);
            }
        }
        u8* q = dst + (size_t)(f0 + k) * dst_frame_stride + out_off;
        q[0] = (u8)o[0];
        q[1] = (u8)o[1];
        q[2] = (u8)o[2];
    }
}

int launch_warp(cbv_ctx* ctx, const u8* src, Geom g, const double* Minv9, int dw, int dh, int rot180, u8* dst,
                int dst_stride, size_t dst_frame_stride, NormSrc norm, int batch, u32* zero_word, u32* zero_word2)
{
    WarpM M;
    for (int i = 0; i < 9; i++) M.m[i] = Minv9[i];
    const int BLOCK_SZ = 32;
    int bh0 = BLOCK_SZ / 2 < dh ? BLOCK_SZ / 2 : dh;
    int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < dw ? BLOCK_SZ * BLOCK_SZ / bh0 : dw;
    bh0 = BLOCK_SZ * BLOCK_SZ / bw0 < dh ? BLOCK_SZ * BLOCK_SZ / bw0 : dh;
    prof_begin(ctx, CBV_K_WARP);
    // batched launches: four frames per thread (eight: 13 % fewer instructions again, no change on the path); a launch of
    // a frame or two keeps one thread per pixel and frame
    if (batch >= 8) {
        dim3 grid((dw + 63) / 64, (dh + 3) / 4, (batch + 3) / 4);
        if (norm.minmax)
            hipLaunchKernelGGL((k_warp<4, true>), grid, dim3(256), 0, ctx->stream, src, g, M, dw, dh, bw0, bh0, rot180, dst, dst_stride,
                               dst_frame_stride, nullptr, norm.minmax, norm.mm_stride, zero_word, zero_word2, batch);
        else
            hipLaunchKernelGGL((k_warp<4, false>), grid, dim3(256), 0, ctx->stream, src, g, M, dw, dh, bw0, bh0, rot180, dst, dst_stride,
                               dst_frame_stride, norm.lut, nullptr, 0, zero_word, zero_word2, batch);
    } else {
        dim3 grid((dw + 63) / 64, (dh + 3) / 4, batch);
        if (norm.minmax)
            hipLaunchKernelGGL((k_warp<1, true>), grid, dim3(256), 0, ctx->stream, src, g, M, dw, dh, bw0, bh0, rot180, dst, dst_stride,
                               dst_frame_stride, nullptr, norm.minmax, norm.mm_stride, zero_word, zero_word2, batch);
        else
            hipLaunchKernelGGL((k_warp<1, false>), grid, dim3(256), 0, ctx->stream, src, g, M, dw, dh, bw0, bh0, rot180, dst, dst_stride,
                               dst_frame_stride, norm.lut, nullptr, 0, zero_word, zero_word2, batch);
    }
    prof_end(ctx, CBV_K_WARP);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// synthetic frames; mirrors oracle orc_synth_frame bit for bit
// ---------------------------------------------------------------------------
__device__ __forceinline__ u64 d_mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_synth(u8* __restrict__ dst, Geom g, const u64* __restrict__ seeds,
                                                const double* __restrict__ Hinv, const u8* __restrict__ boards,
                                                const cbv_scene* __restrict__ scp)
{
    __shared__ u8 board[64];
    __shared__ cbv_scene sc;
    if (threadIdx.x < 64) board[threadIdx.x] = boards[(size_t)blockIdx.z * 64 + threadIdx.x];
    if (threadIdx.x == 0) sc = *scp;
    __syncthreads();
    const u64 seed = seeds[blockIdx.z];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.w || y >= g.h) return;
    const u64 idx = (u64)y * (u64)g.w + (u64)x;
    const u64 hn = d_mix64(seed + idx * 0x9E3779B97F4A7C15ULL);
    const u64 hb = d_mix64(0x5851F42D4C957F2DULL + (u64)(y >> 4) * 4096 + (u64)(x >> 4));
    const int bgv = sc.bg_lo + (int)(hb % (u64)(sc.bg_span ? sc.bg_span : 1));
    int c0 = bgv, c1 = bgv, c2 = bgv;
    const double W = Hinv[6] * x + Hinv[7] * y + Hinv[8];
    const double u = (Hinv[0] * x + Hinv[1] * y + Hinv[2]) / W;
    const double v = (Hinv[3] * x + Hinv[4] * y + Hinv[5]) / W;
    if (u >= 0.0 && u < 8.0 && v >= 0.0 && v < 8.0) {
        const int fi = (int)u, ri = (int)v;
        const u8* c = ((fi + ri) & 1) ? sc.dark : sc.light;
        const int piece = board[ri * 8 + fi];
        if (piece) {
            const double du = u - (fi + 0.5), dv = v - (ri + 0.5);
            if (du * du + dv * dv <= sc.radius * sc.radius) c = piece == 1 ? sc.white : sc.black;
        }
        c0 = c[0];
        c1 = c[1];
        c2 = c[2];
    }
    const int span = 2 * sc.noise + 1;
    u8* q = dst + (size_t)blockIdx.z * g.frame_stride + (size_t)y * g.stride + (size_t)x * 3;
    q[0] = d_sat8(c0 + (int)((hn >> 0) & 0xFFFF) % span - sc.noise);
    q[1] = d_sat8(c1 + (int)((hn >> 16) & 0xFFFF) % span - sc.noise);
    q[2] = d_sat8(c2 + (int)((hn >> 32) & 0xFFFF) % span - sc.noise);
}

int launch_synth(cbv_ctx* ctx, u8* dst, Geom g, const u64* seeds_dev, const double* hinv_dev, const u8* boards_dev,
                 const cbv_scene* scene_dev, int batch)
{
    dim3 grid((g.w + 63) / 64, (g.h + 3) / 4, batch);
    prof_begin(ctx, CBV_K_SYNTH);
    hipLaunchKernelGGL(k_synth, grid, dim3(256), 0, ctx->stream, dst, g, seeds_dev, hinv_dev, boards_dev, scene_dev);
    prof_end(ctx, CBV_K_SYNTH);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}
