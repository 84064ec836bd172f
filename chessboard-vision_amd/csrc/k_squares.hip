// Per-square work of ChangeDetector (change_detector.py) and PieceDetector
// (piece_detector.py).  One workgroup per square (<= 128 x 128 px):
//   k_squares_preprocess  BGR2GRAY + GaussianBlur((k,k),0) on the ROI alone
//                         (REFLECT_101 at the square's own edges)
//   k_squares_stats       every sum the host/device decision chains need
//   k_squares_calibrate / _ema / _set_ref   background-model state updates
//   k_scan                detect_all_pieces' temporal logic over a batch of
//                         frames: 64 independent per-square chains
#include "cbv_device.h"

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_squares_preprocess(const u8* __restrict__ src, size_t src_frame_stride,
                                                             const SquareDesc* __restrict__ descs,
                                                             const int* __restrict__ coef, int blur_k,
                                                             u8* __restrict__ gray, size_t gray_frame_stride)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const SquareDesc d = descs[blockIdx.x];
    if (d.cn == 0) return; // square not supplied in this call: its current gray is kept
    const int n = d.w * d.h;
    u8* g = smem;                                   // n bytes
    u16* hb = (u16*)(smem + ((n + 15) & ~15));      // n u16
    __shared__ int cf[32];
    if (threadIdx.x < blur_k && threadIdx.x < 32) cf[threadIdx.x] = coef[threadIdx.x];
    const u8* s = src + (size_t)blockIdx.z * src_frame_stride + d.src_off;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / d.w, x = i - y * d.w;
        const u8* p = s + (size_t)y * d.stride + (size_t)x * d.cn;
        g[i] = d.cn == 3 ? (u8)d_gray(p[0], p[1], p[2]) : p[0];
    }
    __syncthreads();
    u8* out = gray + (size_t)blockIdx.z * gray_frame_stride + d.plane_off;
    if (blur_k <= 1) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = g[i];
        return;
    }
    const int r = blur_k >> 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / d.w, x = i - y * d.w;
        u32 acc = 0;
        for (int j = 0; j < blur_k; j++) acc += (u32)cf[j] * g[y * d.w + d_reflect101(x + j - r, d.w)];
        hb[i] = (u16)min(acc, 65535u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / d.w, x = i - y * d.w;
        u32 acc = 0;
        for (int j = 0; j < blur_k; j++) acc += (u32)cf[j] * hb[d_reflect101(y + j - r, d.h) * d.w + x];
        const u32 v = (acc + (1u << 15)) >> 16;
        out[i] = (u8)min(v, 255u);
    }
}

// The default blur (k = 5: 1-4-6-4-1, 8.8 fixed point 16-64-96-64-16) without per-pixel divisions:
// lanes walk the square as a 16 x 16 grid, reflected neighbour columns/rows are resolved once per
// column/row, gray rows are staged as u8 and the horizontal pass as u16 in LDS.
__global__ __launch_bounds__(256) void k_squares_preprocess5(const u8* __restrict__ src, size_t src_frame_stride,
                                                              const SquareDesc* __restrict__ descs,
                                                              u8* __restrict__ gray, size_t gray_frame_stride)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const SquareDesc d = descs[blockIdx.x];
    if (d.cn == 0) return;
    const int w = d.w, h = d.h, n = w * h;
    u8* g = smem;
    u16* hb = (u16*)(smem + ((n + 15) & ~15));
    const u8* s = src + (size_t)blockIdx.z * src_frame_stride + d.src_off;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (d.cn == 3) {
        // four pixels (12 bytes, any alignment) per lane and load instruction: the ROI read is bound by the number
        // of memory instructions, not by bytes
        const int ngx = (w + 3) >> 2, ntask = ngx * h;
        // four tasks a lane per round, all their loads issued before the first result is stored (a load-compute-
        // store loop pays one memory latency per iteration)
        for (int t0 = threadIdx.x; t0 < ntask; t0 += 4 * 256) {
            u32 v[4][3];
            int yy[4], xx[4];
            bool full[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int t = t0 + q * 256;
                full[q] = false;
                if (t < ntask) {
                    yy[q] = t / ngx;
                    xx[q] = (t - yy[q] * ngx) << 2;
                    full[q] = xx[q] + 3 < w;
                    if (full[q]) __builtin_memcpy(v[q], s + (size_t)yy[q] * d.stride + 3 * xx[q], 12);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int t = t0 + q * 256;
                if (t >= ntask) continue;
                u8* o = g + yy[q] * w + xx[q];
                if (full[q]) {
                    o[0] = (u8)d_gray(v[q][0] & 255, (v[q][0] >> 8) & 255, (v[q][0] >> 16) & 255);
                    o[1] = (u8)d_gray(v[q][0] >> 24, v[q][1] & 255, (v[q][1] >> 8) & 255);
                    o[2] = (u8)d_gray((v[q][1] >> 16) & 255, v[q][1] >> 24, v[q][2] & 255);
                    o[3] = (u8)d_gray((v[q][2] >> 8) & 255, (v[q][2] >> 16) & 255, v[q][2] >> 24);
                } else {
                    const u8* p = s + (size_t)yy[q] * d.stride + 3 * xx[q];
                    for (int k = 0; xx[q] + k < w; k++) o[k] = (u8)d_gray(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
                }
            }
        }
    } else {
        for (int y = ty; y < h; y += 16)
            for (int x = tx; x < w; x += 16) g[y * w + x] = s[(size_t)y * d.stride + x];
    }
    __syncthreads();
    for (int x = tx; x < w; x += 16) {
        const int x0 = d_reflect101(x - 2, w), x1 = d_reflect101(x - 1, w), x3 = d_reflect101(x + 1, w), x4 = d_reflect101(x + 2, w);
        for (int y = ty; y < h; y += 16) {
            const u8* r = g + y * w;
            hb[y * w + x] = (u16)(16 * (r[x0] + r[x4]) + 64 * (r[x1] + r[x3]) + 96 * r[x]);
        }
    }
    __syncthreads();
    u8* out = gray + (size_t)blockIdx.z * gray_frame_stride + d.plane_off;
    for (int y = ty; y < h; y += 16) {
        const int y0 = d_reflect101(y - 2, h) * w, y1 = d_reflect101(y - 1, h) * w, y3 = d_reflect101(y + 1, h) * w, y4 = d_reflect101(y + 2, h) * w;
        for (int x = tx; x < w; x += 16) {
            const u32 acc = 16u * (hb[y0 + x] + hb[y4 + x]) + 64u * (hb[y1 + x] + hb[y3 + x]) + 96u * hb[y * w + x];
            out[y * w + x] = (u8)((acc + (1u << 15)) >> 16);
        }
    }
}

int launch_squares_preprocess(cbv_ctx* ctx, const u8* src, size_t src_frame_stride, const SquareDesc* descs, int n,
                              const int* coef_dev, int blur_k, u8* gray, size_t gray_frame_stride, int batch, int max_px)
{
    // LDS by the largest square of the set (u8 gray + u16 horizontal pass), not by the largest square allowed:
    // 18 KB instead of 48 KB for 77 x 77 squares, i.e. 8 instead of 3 workgroups a CU
    if (max_px <= 0 || max_px > CBV_MAX_SQUARE_DIM * CBV_MAX_SQUARE_DIM) max_px = CBV_MAX_SQUARE_DIM * CBV_MAX_SQUARE_DIM;
    size_t lds = (size_t)((max_px + 15) & ~15) + 2 * (size_t)max_px;
    prof_begin(ctx, CBV_K_SQUARES);
    if (blur_k == 5)
        hipLaunchKernelGGL(k_squares_preprocess5, dim3(n, 1, batch), dim3(256), lds, ctx->stream, src, src_frame_stride, descs,
                           gray, gray_frame_stride);
    else
        hipLaunchKernelGGL(k_squares_preprocess, dim3(n, 1, batch), dim3(256), lds, ctx->stream, src, src_frame_stride,
                           descs, coef_dev, blur_k, gray, gray_frame_stride);
    prof_end(ctx, CBV_K_SQUARES);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// PieceDetector.detect_piece decision chain without HoughCircles
// (piece_detector.py:303-345), evaluated in double like numpy does.
// ---------------------------------------------------------------------------
// detect_piece without its HoughCircles step (piece_detector.py:303-345):
// 0 = uniform square (std < 15: no piece, nothing else is tried), 1 = centre-vs-border or radial symmetry says
// piece, 2 = neither does (only then the outcome of HoughCircles decides `has_piece`).
__device__ int d_detect_piece(const cbv_sq_stats& st)
{
    // np.std(gray) < 15  <=>  n*sumsq - sum^2 < 225 n^2   (exact integer form)
    const long long n = st.n, s = st.sum;
    const long long lhs = n * (long long)st.sumsq - s * s;
    if (lhs < 225ll * n * n) return 0;
    const double cm = (double)st.center_sum / (double)st.center_cnt;
    const double bm = (double)st.border_sum / (double)st.border_cnt;
    const double diff = fabs(cm - bm);
    if (diff > 40.0) return 1;
    double rm[4];
    int nr = 0;
    for (int k = 0; k < 4; k++)
        if (st.ring_cnt[k] > 0) rm[nr++] = (double)st.ring_sum[k] / (double)st.ring_cnt[k];
    if (nr < 2) return 2;
    double sum = 0;
    for (int k = 0; k < nr; k++) sum = sum + rm[k];
    const double mean = sum / nr;
    double sq = 0;
    for (int k = 0; k < nr; k++) {
        const double x = rm[k] - mean;
        sq = sq + x * x;
    }
    const double variance = sq / nr;
    const double score = fmin(1.0, variance / 500.0);
    return score > 0.6 ? 1 : 2;
}

// ---------------------------------------------------------------------------
// per-square statistics, shared by k_squares_stats (planes already in memory) and the fused
// k_squares_pre5_stats (pixels coming straight out of the blur)
// ---------------------------------------------------------------------------
struct SqAccum {
    u32 v[17];
    float zmax;
    int nan_seen;
};

__device__ __forceinline__ void sq_accum_init(SqAccum& A)
{
#pragma unroll
    for (int k = 0; k < 17; k++) A.v[k] = 0;
    A.zmax = 0.f;
    A.nan_seen = 0;
}

// sd = np.sqrt(var) of the pixel, float32 correctly rounded: it does not depend on the frame, so the kernels that write
// the variance plane (calibrate, EMA, a plane set by the host) store it beside the variance and the statistics kernels
// read it instead of taking the square root per pixel and frame
__device__ __forceinline__ void sq_accum_px(SqAccum& A, int gv, u32 mk, bool has_ref, int refv, bool has_model, float mu, float sd,
                                            float z_thresh)
{
    A.v[0] += gv;
    A.v[1] += gv * gv;
    if (has_ref) A.v[2] += (u32)abs(gv - refv);
    // (the regions' pixel counts depend on the square's shape only: SquareDesc::cnt, counted once on the host)
    A.v[3] += (mk & 1) ? gv : 0;
    A.v[5] += (mk & 2) ? gv : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) A.v[7 + k] += (mk & (4u << k)) ? gv : 0;
    if (has_model) {
        // change_detector.py:131-137 in float32: sqrt (stored) and division are IEEE-rounded
        const float df = fabsf((float)gv - mu);
        const float z = __fdiv_rn(df, sd);
        if (z > z_thresh) A.v[15]++;
        if (z != z) A.nan_seen = 1;
        else A.zmax = fmaxf(A.zmax, z);
    }
}

// block reduction + the record and the decision byte (thread 0).  acc[20], zm[waves], nanf_[1] are LDS, zeroed and
// synchronised by the caller before any lane gets here.
__device__ __forceinline__ void sq_accum_finish(SqAccum& A, u32* acc, float* zm, int* nanf_, int n, bool has_model,
                                                cbv_sq_stats* __restrict__ out, int nsq, u8* __restrict__ decisions,
                                                int want_hough, u32* __restrict__ hough_work,
                                                cbv_hough_result* __restrict__ hough_out, const u32* __restrict__ cnt,
                                                const DetectMasks dm = DetectMasks())
{
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k == 4 || k == 6 || (k >= 11 && k <= 14)) continue; // region counts come from the descriptor
        u32 s = wave_sum_u32(A.v[k]);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&acc[k], s);
    }
    const float zmax = wave_max_f32(A.zmax);
    if ((threadIdx.x & 63) == 0) zm[threadIdx.x >> 6] = zmax;
    if (A.nan_seen) nanf_[0] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        cbv_sq_stats st;
        st.n = (u32)n;
        st.sum = acc[0];
        st.sumsq = acc[1];
        st.sad_ref = acc[2];
        st.center_sum = acc[3];
        st.center_cnt = cnt[0];
        st.border_sum = acc[5];
        st.border_cnt = cnt[1];
        for (int k = 0; k < 4; k++) {
            st.ring_sum[k] = acc[7 + k];
            st.ring_cnt[k] = cnt[2 + k];
        }
        st.z_count = acc[15];
        float z = zm[0];
        for (int k = 1; k < (int)(blockDim.x >> 6); k++) z = fmaxf(z, zm[k]);
        st.z_max = nanf_[0] ? __builtin_nanf("") : z;
        out[(size_t)blockIdx.z * nsq + blockIdx.x] = st;
        if (dm.dflags) {
            // detect_all_pieces' per-square gate (piece_detector.py:367-395) for the class API, one frame: which squares
            // changed against their reference, which are processed, and which of those need HoughCircles (the reference
            // runs it on every square it evaluates whose std is >= 15, before the two statistics-based tests)
            const u64 bit = 1ull << blockIdx.x;
            const bool has_ref = (dm.has_ref & bit) != 0, cached = (dm.cached & bit) != 0;
            const bool changed = !has_ref || (double)st.sad_ref / (double)st.n > dm.change_threshold; // np.mean(diff) > thr
            bool should = dm.check_given && (dm.check & bit);
            if (!should && (!dm.check_given || dm.use_delta)) should = !cached || changed;
            const bool evaluated = should || !cached;
            const long long nn = st.n, sm = st.sum;
            const bool std_ok = !(nn * (long long)st.sumsq - sm * sm < 225ll * nn * nn);
            dm.dflags[blockIdx.x] = (u8)((changed ? 1u : 0u) | (should ? 2u : 0u) | (evaluated ? 4u : 0u) | (std_ok ? 8u : 0u));
            cbv_hough_result r;
            memset(&r, 0, sizeof(r));
            r.flags = CBV_HOUGH_SKIPPED;
            if (want_hough && evaluated && std_ok) hough_work[1 + atomicAdd(&hough_work[0], 1u)] = blockIdx.x;
            else if (hough_out) hough_out[blockIdx.x] = r;
        }
        if (decisions) {
            // the frame-parallel part of both detectors' decisions, so the sequential scan only does integer work:
            // bit0 detect_piece(square), bits 1-3 ChangeDetector class (in dict / PARCIAL / TOTAL),
            // bit4 "HoughCircles decides": has_piece is an OR, so k_hough only has to run where the two
            // statistics-based detectors said no on a non-uniform square; it then sets bit0 itself
            const int dp = d_detect_piece(st);
            u32 dc = dp == 1 ? 1u : 0u;
            if (want_hough && (dp == 2 || (want_hough == 2 && dp == 1))) {
                dc |= 16u;
                if (hough_work) hough_work[1 + atomicAdd(&hough_work[0], 1u)] = ((u32)blockIdx.z << 8) | blockIdx.x;
            } else if (want_hough && hough_out) {
                cbv_hough_result r;
                memset(&r, 0, sizeof(r));
                r.flags = CBV_HOUGH_SKIPPED;
                hough_out[(size_t)blockIdx.z * CBV_MAX_SQUARES + blockIdx.x] = r;
            }
            if (has_model) {
                const double pct = ((double)st.z_count / (double)st.n) * 100.0; // change_detector.py:139 as a Python float
                if (!(pct < 5.0)) dc |= 2u | (pct > 75.0 ? 8u : (pct > 15.0 ? 4u : 0u));
            }
            decisions[(size_t)blockIdx.z * CBV_MAX_SQUARES + blockIdx.x] = (u8)dc;
        }
    }
}

__global__ __launch_bounds__(256) void k_squares_stats(const SquareDesc* __restrict__ descs,
                                                        const u8* __restrict__ gray, size_t gray_frame_stride,
                                                        const u8* __restrict__ ref, const float* __restrict__ mean,
                                                        const float* __restrict__ var, const u8* __restrict__ masks,
                                                        float z_thresh, cbv_sq_stats* __restrict__ out, int nsq,
                                                        u8* __restrict__ decisions, int want_hough, u32* __restrict__ hough_work,
                                                        cbv_hough_result* __restrict__ hough_out, const DetectMasks dm)
{
    __shared__ u32 acc[20];
    __shared__ float zm[4];
    __shared__ int nanf_[1];
    const SquareDesc d = descs[blockIdx.x];
    const int n = d.w * d.h;
    if (threadIdx.x < 20) acc[threadIdx.x] = 0;
    if (threadIdx.x == 0) nanf_[0] = 0;
    __syncthreads();
    const u8* g = gray + (size_t)blockIdx.z * gray_frame_stride + d.plane_off;
    const u8* m = masks + d.mask_off;
    SqAccum A;
    sq_accum_init(A);
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        sq_accum_px(A, g[i], m[i], ref != nullptr, ref ? (int)ref[d.plane_off + i] : 0, mean != nullptr,
                    mean ? mean[d.plane_off + i] : 0.f, mean ? var[d.plane_off + i] : 1.f, z_thresh);
    sq_accum_finish(A, acc, zm, nanf_, n, mean != nullptr, out, nsq, decisions, want_hough, hough_work, hough_out, descs[blockIdx.x].cnt, dm);
}

// preprocess (k = 5) and statistics of the pipeline in one pass: the statistics are sums over the plane the blur
// produces, so they are taken as the pixels leave the vertical pass (one launch and one read of the plane less)
// NT lanes per square: 256 in batched launches (the chip is full of squares), 1024 when a launch holds only a frame or
// two (64-128 squares on 256 CUs: then a square's three passes are latency, and four times the lanes cut it).
// GATE: the class-API launch (one frame: |gray - reference| and detect_all_pieces' per-square gate); the batched pipeline
// launches compile without it
template <int NT, bool GATE>
__global__ __launch_bounds__(NT) void k_squares_pre5_stats(const u8* __restrict__ src, size_t src_frame_stride,
                                                             const SquareDesc* __restrict__ descs, u8* __restrict__ gray,
                                                             size_t gray_frame_stride, const float* __restrict__ mean,
                                                             const float* __restrict__ var, const u8* __restrict__ masks,
                                                             float z_thresh, cbv_sq_stats* __restrict__ out, int nsq,
                                                             u8* __restrict__ decisions, int want_hough,
                                                             u32* __restrict__ hough_work, cbv_hough_result* __restrict__ hough_out,
                                                             const u8* __restrict__ ref, const DetectMasks dm)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    __shared__ u32 acc[20];
    __shared__ float zm[NT / 64];
    __shared__ int nanf_[1];
    const SquareDesc d = descs[blockIdx.x];
    const int w = d.w, h = d.h, n = w * h;
    u8* g = smem;
    u16* hb = (u16*)(smem + ((n + 15) & ~15));
    const u8* s = src + (size_t)blockIdx.z * src_frame_stride + d.src_off;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (threadIdx.x < 20) acc[threadIdx.x] = 0;
    if (threadIdx.x == 0) nanf_[0] = 0;
    {
        const int ngx = (w + 3) >> 2, ntask = ngx * h;
        for (int t0 = threadIdx.x; t0 < ntask; t0 += 4 * NT) {
            u32 v[4][3];
            int yy[4], xx[4];
            bool full[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int t = t0 + q * NT;
                full[q] = false;
                if (t < ntask) {
                    yy[q] = t / ngx;
                    xx[q] = (t - yy[q] * ngx) << 2;
                    full[q] = xx[q] + 3 < w;
                    if (full[q]) __builtin_memcpy(v[q], s + (size_t)yy[q] * d.stride + 3 * xx[q], 12);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int t = t0 + q * NT;
                if (t >= ntask) continue;
                u8* o = g + yy[q] * w + xx[q];
                if (full[q]) {
                    o[0] = (u8)d_gray(v[q][0] & 255, (v[q][0] >> 8) & 255, (v[q][0] >> 16) & 255);
                    o[1] = (u8)d_gray(v[q][0] >> 24, v[q][1] & 255, (v[q][1] >> 8) & 255);
                    o[2] = (u8)d_gray((v[q][1] >> 16) & 255, v[q][1] >> 24, v[q][2] & 255);
                    o[3] = (u8)d_gray((v[q][2] >> 8) & 255, (v[q][2] >> 16) & 255, v[q][2] >> 24);
                } else {
                    const u8* p = s + (size_t)yy[q] * d.stride + 3 * xx[q];
                    for (int k = 0; xx[q] + k < w; k++) o[k] = (u8)d_gray(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
                }
            }
        }
    }
    __syncthreads();
    for (int x = tx; x < w; x += 16) {
        const int x0 = d_reflect101(x - 2, w), x1 = d_reflect101(x - 1, w), x3 = d_reflect101(x + 1, w), x4 = d_reflect101(x + 2, w);
        for (int y = ty; y < h; y += NT / 16) {
            const u8* r = g + y * w;
            hb[y * w + x] = (u16)(16 * (r[x0] + r[x4]) + 64 * (r[x1] + r[x3]) + 96 * r[x]);
        }
    }
    __syncthreads();
    u8* outp = gray + (size_t)blockIdx.z * gray_frame_stride + d.plane_off;
    const u8* m = masks + d.mask_off;
    const float* mp = mean ? mean + d.plane_off : nullptr;
    const float* vp = mean ? var + d.plane_off : nullptr;
    // class API: sum |gray - reference| of squares that have one (piece_detector.py:82-93)
    const u8* rp = (GATE && ref && ((dm.has_ref >> blockIdx.x) & 1ull)) ? ref + d.plane_off : nullptr;
    SqAccum A;
    sq_accum_init(A);
    for (int y = ty; y < h; y += NT / 16) {
        const int y0 = d_reflect101(y - 2, h) * w, y1 = d_reflect101(y - 1, h) * w, y3 = d_reflect101(y + 1, h) * w, y4 = d_reflect101(y + 2, h) * w;
        for (int x = tx; x < w; x += 16) {
            const u32 a2 = 16u * (hb[y0 + x] + hb[y4 + x]) + 64u * (hb[y1 + x] + hb[y3 + x]) + 96u * hb[y * w + x];
            const int gv = (int)((a2 + (1u << 15)) >> 16);
            const int i = y * w + x;
            outp[i] = (u8)gv;
            sq_accum_px(A, gv, m[i], GATE && rp != nullptr, (GATE && rp) ? (int)rp[i] : 0, mp != nullptr, mp ? mp[i] : 0.f, mp ? vp[i] : 1.f, z_thresh);
        }
    }
    sq_accum_finish(A, acc, zm, nanf_, n, mp != nullptr, out, nsq, decisions, want_hough, hough_work, hough_out, descs[blockIdx.x].cnt,
                    GATE ? dm : DetectMasks());
}

int launch_squares_pre5_stats(cbv_ctx* ctx, const u8* src, size_t src_frame_stride, const SquareDesc* descs, int n, u8* gray,
                              size_t gray_frame_stride, const float* mean, const float* var, const u8* masks, float z_thresh,
                              cbv_sq_stats* out, int batch, u8* decisions, int want_hough, u32* hough_work,
                              cbv_hough_result* hough_out, int max_px, const u8* ref, const DetectMasks* dmp)
{
    const DetectMasks dm = dmp ? *dmp : DetectMasks();
    if (max_px <= 0 || max_px > CBV_MAX_SQUARE_DIM * CBV_MAX_SQUARE_DIM) max_px = CBV_MAX_SQUARE_DIM * CBV_MAX_SQUARE_DIM;
    const size_t lds = (size_t)((max_px + 15) & ~15) + 2 * (size_t)max_px;
    prof_begin(ctx, CBV_K_SQUARES);
    if (dm.dflags)
        hipLaunchKernelGGL((k_squares_pre5_stats<1024, true>), dim3(n, 1, batch), dim3(1024), lds, ctx->stream, src, src_frame_stride, descs, gray,
                           gray_frame_stride, mean, var, masks, z_thresh, out, n, decisions, want_hough, hough_work, hough_out, ref, dm);
    else if ((long long)n * batch <= 2 * ctx->num_cus)
        hipLaunchKernelGGL((k_squares_pre5_stats<1024, false>), dim3(n, 1, batch), dim3(1024), lds, ctx->stream, src, src_frame_stride, descs, gray,
                           gray_frame_stride, mean, var, masks, z_thresh, out, n, decisions, want_hough, hough_work, hough_out, ref, dm);
    else
        hipLaunchKernelGGL((k_squares_pre5_stats<256, false>), dim3(n, 1, batch), dim3(256), lds, ctx->stream, src, src_frame_stride, descs, gray,
                           gray_frame_stride, mean, var, masks, z_thresh, out, n, decisions, want_hough, hough_work, hough_out, ref, dm);
    prof_end(ctx, CBV_K_SQUARES);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

int launch_squares_stats(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, size_t gray_frame_stride,
                         const u8* ref, const float* mean, const float* var, const u8* masks, float z_thresh,
                         cbv_sq_stats* out, int batch, u8* decisions, int want_hough, u32* hough_work,
                         cbv_hough_result* hough_out, const DetectMasks* dmp)
{
    const DetectMasks dm = dmp ? *dmp : DetectMasks();
    prof_begin(ctx, CBV_K_SQUARES);
    hipLaunchKernelGGL(k_squares_stats, dim3(n, 1, batch), dim3(256), 0, ctx->stream, descs, gray, gray_frame_stride,
                       ref, mean, var, masks, z_thresh, out, n, decisions, want_hough, hough_work, hough_out, dm);
    prof_end(ctx, CBV_K_SQUARES);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// ---------------------------------------------------------------------------
__global__ void k_squares_calibrate(const SquareDesc* __restrict__ descs, const u8* __restrict__ gray,
                                    float* __restrict__ mean, float* __restrict__ var, float* __restrict__ sd, float init_var,
                                    const u8* __restrict__ select)
{
    if (select && !select[blockIdx.x]) return;
    const SquareDesc d = descs[blockIdx.x];
    const float isd = d_sqrt_rn(init_var);
    for (int i = threadIdx.x; i < d.w * d.h; i += blockDim.x) {
        mean[d.plane_off + i] = (float)gray[d.plane_off + i];
        var[d.plane_off + i] = init_var;
        sd[d.plane_off + i] = isd;
    }
}

// sd plane of one square after the host wrote its variance plane (cbv_squares_set)
__global__ void k_squares_refresh_sd(const SquareDesc* __restrict__ descs, const float* __restrict__ var, float* __restrict__ sd, int index)
{
    const SquareDesc d = descs[index];
    for (int i = threadIdx.x; i < d.w * d.h; i += blockDim.x) sd[d.plane_off + i] = d_sqrt_rn(var[d.plane_off + i]);
}

int launch_squares_refresh_sd(cbv_ctx* ctx, const SquareDesc* descs, const float* var, float* sd, int index)
{
    hipLaunchKernelGGL(k_squares_refresh_sd, dim3(1), dim3(256), 0, ctx->stream, descs, var, sd, index);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// update_all_references (change_detector.py:77-92), float32, one rounding per op
__global__ void k_squares_ema(const SquareDesc* __restrict__ descs, const u8* __restrict__ gray,
                              float* __restrict__ mean, float* __restrict__ var, float* __restrict__ sd, float one_minus, float alpha,
                              const u8* __restrict__ select)
{
    if (select && !select[blockIdx.x]) return;
    const SquareDesc d = descs[blockIdx.x];
    for (int i = threadIdx.x; i < d.w * d.h; i += blockDim.x) {
        const float gv = (float)gray[d.plane_off + i];
        const float m1 = one_minus * mean[d.plane_off + i];
        const float m2 = alpha * gv;
        const float nm = m1 + m2;
        const float df = gv - nm;
        const float d2 = df * df;
        const float v1 = one_minus * var[d.plane_off + i];
        const float v2 = alpha * d2;
        float nv = v1 + v2;
        if (!(nv >= 10.0f)) nv = (nv != nv) ? nv : 10.0f; // np.maximum propagates NaN
        mean[d.plane_off + i] = nm;
        var[d.plane_off + i] = nv;
        sd[d.plane_off + i] = d_sqrt_rn(nv);
    }
}

__global__ void k_squares_set_ref(const SquareDesc* __restrict__ descs, const u8* __restrict__ gray,
                                  u8* __restrict__ ref, const u8* __restrict__ select)
{
    if (select && !select[blockIdx.x]) return;
    const SquareDesc d = descs[blockIdx.x];
    for (int i = threadIdx.x; i < d.w * d.h; i += blockDim.x) ref[d.plane_off + i] = gray[d.plane_off + i];
}

// PieceDetector.update_references (piece_detector.py:447-453) on the scan state: reference = the slot's plane,
// cached results cleared, detection history kept
__global__ void k_scan_update_refs(const SquareDesc* __restrict__ descs, const u8* __restrict__ gray, u8* __restrict__ ref,
                                   ScanState* __restrict__ state)
{
    const SquareDesc d = descs[blockIdx.x];
    for (int i = threadIdx.x; i < d.w * d.h; i += blockDim.x) ref[d.plane_off + i] = gray[d.plane_off + i];
    if (threadIdx.x == 0) {
        state[blockIdx.x].has_ref = 1;
        state[blockIdx.x].has_cache = 0;
        state[blockIdx.x].cached_raw = 0;
    }
}

int launch_scan_update_refs(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, u8* ref, ScanState* state)
{
    hipLaunchKernelGGL(k_scan_update_refs, dim3(n), dim3(256), 0, ctx->stream, descs, gray, ref, state);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

int launch_squares_calibrate(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, float* mean, float* var, float* sd,
                             float init_var, const u8* select)
{
    hipLaunchKernelGGL(k_squares_calibrate, dim3(n), dim3(256), 0, ctx->stream, descs, gray, mean, var, sd, init_var, select);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

int launch_squares_ema(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, float* mean, float* var, float* sd,
                       double alpha, const u8* select)
{
    // (1 - self.alpha) and self.alpha are python doubles turned float32 by numpy (weak scalars)
    float one_minus = (float)(1.0 - alpha), a = (float)alpha;
    hipLaunchKernelGGL(k_squares_ema, dim3(n), dim3(256), 0, ctx->stream, descs, gray, mean, var, sd, one_minus, a, select);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// reference_squares[pos] = gray.copy() for the squares of a 64-bit set: the set rides in the kernel arguments, so the
// launch needs no host buffer (and no synchronisation on the host's side)
__global__ void k_squares_set_ref_mask(const SquareDesc* __restrict__ descs, const u8* __restrict__ gray, u8* __restrict__ ref, u64 mask)
{
    if (!((mask >> blockIdx.x) & 1ull)) return;
    const SquareDesc d = descs[blockIdx.x];
    for (int i = threadIdx.x; i < d.w * d.h; i += blockDim.x) ref[d.plane_off + i] = gray[d.plane_off + i];
}

int launch_squares_set_ref_mask(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, u8* ref, u64 mask)
{
    hipLaunchKernelGGL(k_squares_set_ref_mask, dim3(n), dim3(256), 0, ctx->stream, descs, gray, ref, mask);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

int launch_squares_set_ref(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, u8* ref, const u8* select)
{
    hipLaunchKernelGGL(k_squares_set_ref, dim3(n), dim3(256), 0, ctx->stream, descs, gray, ref, select);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

// detect_all_pieces(use_smoothing=True, use_delta=True, squares_to_check=None)
// (piece_detector.py:348-440) over `count` consecutive frames; one workgroup
// per square, state carried in ScanState and `ref`.
// The chain is sequential per square, so the kernel is latency-bound: the
// square's plane moves as 16-byte vectors (planes are 16-byte aligned and
// zero-padded), the reference lives in registers, the next frame's plane is
// prefetched while the current one is reduced, and one barrier per frame
// (double-buffered partial sums) is all the synchronisation there is.
// One WAVE per square: the per-frame reduction is six DPP steps, there is no LDS and no barrier on
// the critical path of the 512-step chain.

template <int VPT, int SCAN_DEPTH>
__device__ __forceinline__ void scan_body(const SquareDesc d, const ScanParams sp, const u8* __restrict__ gray,
                                          size_t gray_frame_stride, const u8* __restrict__ decisions,
                                          u8* __restrict__ ref, ScanState* __restrict__ state,
                                          u8* __restrict__ flags, int count, const u64* __restrict__ check)
{
    const int sq = blockIdx.x;
    const int n = d.w * d.h;
    const int nvec = (n + 15) >> 4;
    ScanState st = state[sq];
    uint4 rv[VPT], ring[SCAN_DEPTH][VPT];
    u32 sring[SCAN_DEPTH];
    const uint4* refv = (const uint4*)(ref + d.plane_off);
    auto fetch = [&](int t, uint4* dst, u32& sdst) {
        const uint4* gp = (const uint4*)(gray + (size_t)t * gray_frame_stride + d.plane_off);
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int vi = threadIdx.x + k * 64;
            dst[k] = vi < nvec ? gp[vi] : make_uint4(0, 0, 0, 0); // lanes past the plane must compare equal
        }
        sdst = decisions[(size_t)t * CBV_MAX_SQUARES + sq];
        if (check) sdst |= (u32)((check[t] >> sq) & 1ull) << 8; // squares_to_check of that frame
    };
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const int vi = threadIdx.x + k * 64;
        rv[k] = (st.has_ref && vi < nvec) ? refv[vi] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < SCAN_DEPTH; j++)
        if (j < count) fetch(j, ring[j], sring[j]);
    for (int t0 = 0; t0 < count; t0 += SCAN_DEPTH) {
#pragma unroll
        for (int j = 0; j < SCAN_DEPTH; j++) {
            const int t = t0 + j;
            if (t >= count) break;
            uint4 cur[VPT];
#pragma unroll
            for (int k = 0; k < VPT; k++) cur[k] = ring[j][k];
            const u32 dc = sring[j];
            if (t + SCAN_DEPTH < count) fetch(t + SCAN_DEPTH, ring[j], sring[j]);
            u32 sad = 0;
            if (st.has_ref) {
#pragma unroll
                for (int k = 0; k < VPT; k++) {
                    sad = __builtin_amdgcn_sad_u8(cur[k].x, rv[k].x, sad);
                    sad = __builtin_amdgcn_sad_u8(cur[k].y, rv[k].y, sad);
                    sad = __builtin_amdgcn_sad_u8(cur[k].z, rv[k].z, sad);
                    sad = __builtin_amdgcn_sad_u8(cur[k].w, rv[k].w, sad);
                }
            }
            // every lane evaluates the (cheap, uniform) decision chain
            const u32 tot = wave_sum_u32(sad);
            bool changed = true;
            if (st.has_ref) {
                // np.mean(diff) > threshold; with an integral threshold t this is exactly sad > t * n
                // (|sad - t n| >= 1 keeps the quotient far from t compared with an ulp)
                if (sp.thr_is_int) changed = (long long)tot > (long long)sp.thr_int * n;
                else changed = (double)tot / (double)n > sp.change_threshold;
            }
            const bool should_process = !st.has_cache || changed || (dc & 256u); // piece_detector.py:381-389
            const bool in_changes = sp.with_model && (dc & 2u);
            const bool fresh = (dc & 1u) != 0; // detect_piece on the current square (evaluated by k_squares_stats)
            bool raw;
            if (should_process) {
                raw = fresh;
                st.cached_raw = raw;
                st.has_cache = 1;
            } else {
                raw = st.cached_raw != 0;
            }
            // _update_history / _get_stable_detection
            st.hist_bits = (st.hist_bits << 1) | (raw ? 1u : 0u);
            if ((int)st.hist_len < sp.history_size) st.hist_len++;
            st.hist_bits &= (1u << st.hist_len) - 1u;
            const bool stable = st.hist_len < 3 ? raw : ((sp.stable_table >> (st.hist_len * 8 + __popc(st.hist_bits))) & 1) != 0;
            if (should_process && (raw == stable)) {
#pragma unroll
                for (int k = 0; k < VPT; k++) rv[k] = cur[k];
                st.has_ref = 1;
            }
            if (threadIdx.x == 0) {
                // one plain byte store per square and frame (64 workgroups hitting the same result words with
                // atomics would queue behind each other AND in front of this wave's prefetches: vmcnt is in order)
                u32 fl = (raw ? 1u : 0u) | (stable ? 2u : 0u) | (changed ? 4u : 0u) | (should_process ? 8u : 0u);
                if (in_changes) fl |= 16u | ((dc & 8u) ? 64u : ((dc & 4u) ? 32u : 0u)) | (fresh ? 128u : 0u);
                flags[(size_t)t * CBV_MAX_SQUARES + sq] = (u8)fl;
            }
        }
    }
    if (st.has_ref) {
        uint4* refw = (uint4*)(ref + d.plane_off);
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int vi = threadIdx.x + k * 64;
            if (vi < nvec) refw[vi] = rv[k];
        }
    }
    if (threadIdx.x == 0) state[sq] = st;
}

__global__ __launch_bounds__(64) void k_scan(const SquareDesc* __restrict__ descs, ScanParams sp,
                                               const u8* __restrict__ gray, size_t gray_frame_stride,
                                               const u8* __restrict__ decisions, u8* __restrict__ ref,
                                               ScanState* __restrict__ state, u8* __restrict__ flags, int count,
                                               const u64* __restrict__ check)
{
    const SquareDesc d = descs[blockIdx.x];
    const int nvec = (d.w * d.h + 15) >> 4;
    // squares up to 90 x 90 px: eight 16-byte vectors per lane, planes fetched 4 frames ahead (the chain is
    // latency-bound); up to 128 x 128: sixteen vectors, 2 frames ahead
    if (nvec <= 512) scan_body<8, 4>(d, sp, gray, gray_frame_stride, decisions, ref, state, flags, count, check);
    else scan_body<16, 2>(d, sp, gray, gray_frame_stride, decisions, ref, state, flags, count, check);
}

// per-square flag bytes of a frame -> the eight 64-bit square sets of cbv_frame_result
// `mirror` (may be null): the same records written to pinned host memory as well, and `over_src` (HoughCircles' overflow
// counter) copied to `over_dst` there, so that reading results back is a wait and a host copy, not two more launches
// (ResultMirror).
__global__ void k_pack_results(const u8* __restrict__ flags, int n, cbv_frame_result* __restrict__ results, int count, ResultMirror mir)
{
    if (mir.over_dst && blockIdx.x == 0 && threadIdx.x == 0) *mir.over_dst = mir.over_src ? *mir.over_src : 0u;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), sq = threadIdx.x & 63;
    if (t >= count) return;
    const u32 fl = sq < n ? flags[(size_t)t * CBV_MAX_SQUARES + sq] : 0u;
    u64* r = (u64*)&results[t];
    u64* hm = mir.records ? (u64*)&mir.records[t] : nullptr;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const u64 m = __ballot((fl >> b) & 1u);
        if (sq == 0) {
            r[b] = m;
            if (hm) hm[b] = m;
        }
    }
}

// ---------------------------------------------------------------------------
// NoiseHandler.process (noise_handler.py:49-213): a 3-state machine over the per-frame set of
// visually changed squares.  Sequential and tiny: one lane walks the frames.
// ---------------------------------------------------------------------------
__device__ void d_noise_run(const u64* __restrict__ changes, size_t stride_words, int count, cbv_noise_state* __restrict__ state,
                            cbv_noise_result* __restrict__ out)
{
    const int NOISE_THRESHOLD = 3, STABILITY_FRAMES = 12, COOLDOWN_FRAMES = 5;
    cbv_noise_state s = *state;
    for (int t = 0; t < count; t++) {
        const u64 ch = changes[(size_t)t * stride_words];
        const int n = __popcll(ch);
        const bool noisy = n > NOISE_THRESHOLD;
        const int single = n == 1 ? (int)__ffsll((long long)ch) - 1 : -1;
        cbv_noise_result r;
        r.state = 0; r.msg = 0; r.stable = 0; r.lifted = -1; r.count = 0; r.blocked = 0; r.squares = 0;
        if (s.state == 0) { // IDLE
            if (n == 0) { r.state = 0; r.msg = 0; }
            else if (noisy) { s.state = 1; s.cooldown_count = 0; r.state = 1; r.msg = 1; r.count = (u16)n; }
            else {
                s.state = 2; s.pending = ch; s.stable_count = 1; s.lifted = single + 1;
                r.state = 2; r.msg = 2; r.squares = ch; r.lifted = (signed char)single; r.count = 1;
            }
        } else if (s.state == 1) { // NOISE_ACTIVE
            if (noisy) { s.cooldown_count = 0; r.state = 1; r.msg = 6; r.count = (u16)n; }
            else {
                s.cooldown_count++;
                const bool done = (int)s.cooldown_count >= COOLDOWN_FRAMES;
                if (n == 0) {
                    if (done) { s.state = 0; s.cooldown_count = 0; r.state = 0; r.msg = 3; }
                    else { r.state = 1; r.msg = 4; r.count = (u16)s.cooldown_count; }
                } else if (done) {
                    s.state = 2; s.pending = ch; s.stable_count = 1;
                    r.state = 2; r.msg = 7; r.squares = ch;
                } else { r.state = 1; r.msg = 5; r.count = (u16)n; }
            }
        } else { // MOVE_PENDING
            if (noisy) {
                s.state = 1; s.pending = 0; s.stable_count = 0; s.cooldown_count = 0;
                r.state = 1; r.msg = 8; r.count = (u16)n;
            } else if (n == 0) {
                s.stable_count++;
                if ((int)s.stable_count >= STABILITY_FRAMES) {
                    r.state = 0; r.msg = 9; r.squares = s.pending; r.stable = 1;
                    s.state = 0; s.pending = 0; s.stable_count = 0; s.cooldown_count = 0; s.lifted = 0;
                } else { r.state = 2; r.msg = 10; r.squares = s.pending; r.count = (u16)s.stable_count; }
            } else if (ch == s.pending) {
                s.stable_count++;
                if ((int)s.stable_count >= STABILITY_FRAMES) { r.state = 2; r.msg = 11; r.squares = s.pending; r.stable = 1; }
                else {
                    r.state = 2; r.msg = 12; r.squares = s.pending; r.count = (u16)s.stable_count;
                    r.lifted = (signed char)(__popcll(s.pending) == 1 ? s.lifted - 1 : -1);
                }
            } else {
                s.pending = ch; s.stable_count = 1; s.lifted = single + 1;
                r.state = 2; r.msg = 13; r.squares = ch; r.lifted = (signed char)single; r.count = 1;
            }
        }
        r.blocked = s.state == 1 ? 1 : 0;
        out[t] = r;
    }
    *state = s;
}

__global__ void k_noise(const u64* __restrict__ changes, size_t stride_words, int count, cbv_noise_state* __restrict__ state,
                        cbv_noise_result* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    d_noise_run(changes, stride_words, count, state, out);
}

// k_pack_results + k_noise of a run of at most four frames (one workgroup packs them all) in ONE launch: on a run of
// one frame every launch in the chain is ~4.5 us of latency
__global__ __launch_bounds__(256) void k_pack_noise(const u8* __restrict__ flags, int n, cbv_frame_result* __restrict__ results, int count,
                                                     cbv_noise_state* __restrict__ state, cbv_noise_result* __restrict__ out, ResultMirror mir)
{
    if (mir.over_dst && threadIdx.x == 0) *mir.over_dst = mir.over_src ? *mir.over_src : 0u;
    const int t = threadIdx.x >> 6, sq = threadIdx.x & 63;
    if (t < count) {
        const u32 fl = sq < n ? flags[(size_t)t * CBV_MAX_SQUARES + sq] : 0u;
        u64* r = (u64*)&results[t];
        u64* hm = mir.records ? (u64*)&mir.records[t] : nullptr;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const u64 m = __ballot((fl >> b) & 1u);
            if (sq == 0) {
                r[b] = m;
                if (hm) hm[b] = m;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) d_noise_run(&results[0].visual_changes, sizeof(cbv_frame_result) / 8, count, state, out);
}

int launch_noise(cbv_ctx* ctx, const u64* changes, size_t stride_words, int count, cbv_noise_state* state, cbv_noise_result* out)
{
    hipLaunchKernelGGL(k_noise, dim3(1), dim3(64), 0, ctx->stream, changes, stride_words, count, state, out);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

int launch_scan(cbv_ctx* ctx, const SquareDesc* descs, ScanParams sp, const u8* gray, size_t gray_frame_stride,
                const u8* decisions, u8* ref, ScanState* state, u8* flags, cbv_frame_result* results, int count,
                const u64* check, cbv_noise_state* noise_state, cbv_noise_result* noise_out, ResultMirror mir)
{
    // noise_state != null: NoiseHandler over the frames' visual_changes sets follows the scan (game_session.py:165)
    prof_begin(ctx, CBV_K_SCAN);
    hipLaunchKernelGGL(k_scan, dim3(sp.n), dim3(64), 0, ctx->stream, descs, sp, gray, gray_frame_stride, decisions, ref,
                       state, flags, count, check);
    if (noise_state && count <= 4)
        hipLaunchKernelGGL(k_pack_noise, dim3(1), dim3(256), 0, ctx->stream, flags, sp.n, results, count, noise_state, noise_out, mir);
    else
        hipLaunchKernelGGL(k_pack_results, dim3((count + 3) / 4), dim3(256), 0, ctx->stream, flags, sp.n, results, count, mir);
    prof_end(ctx, CBV_K_SCAN);
    CBV_HIP(ctx, hipGetLastError());
    if (noise_state && count > 4)
        return launch_noise(ctx, &results->visual_changes, sizeof(cbv_frame_result) / 8, count, noise_state, noise_out);
    return CBV_OK;
}
