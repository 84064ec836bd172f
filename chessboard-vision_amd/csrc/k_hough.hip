// cv2.HoughCircles(gray, HOUGH_GRADIENT, dp=1.2, minDist=min_dim//3, param1, param2, minRadius, maxRadius)
// for every square of a frame batch plus the "nearest circle to the square centre" pick of
// PieceDetector._detect_circle_unified (piece_detector.py:216-270).
//
// One workgroup per (square, frame); the whole transform lives in LDS:
//   P0 blurred gray plane -> LDS                         P1 Sobel 3x3 (replicate) -> L1 magnitude
//   P2 Canny non-maximum suppression, weak list          P3 hysteresis sweeps over the weak list
//   P4 edge list, gradient-line votes (LDS atomics)      P5 accumulator local maxima > param2
//   P6 per centre radius histogram (one wave a centre)   P7 sort, minDist suppression, pick
// The arithmetic follows the published OpenCV 4.x HoughCirclesGradient step for step (same fixed
// point, same float expressions, one rounding per operation); the results do not depend on the
// order in which edges or centres are visited, so the parallel order here is free.
#include "cbv_device.h"

#define HG_MAXC 512 // accumulator maxima / candidate circles kept per square

struct HgCircle {
    float x, y, r;
    int votes;
};

__device__ __forceinline__ bool hg_before(const HgCircle& a, const HgCircle& b)
{
    if (a.votes != b.votes) return a.votes > b.votes;
    if (a.r != b.r) return a.r > b.r;
    if (a.x != b.x) return a.x < b.x;
    return a.y < b.y;
}

__device__ __forceinline__ void hg_sobel(const u8* g, int w, int h, int x, int y, int& dx, int& dy)
{
    const int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
    const int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
    const int a = g[ym * w + xm], b = g[ym * w + x], c = g[ym * w + xp];
    const int d = g[y * w + xm], f = g[y * w + xp];
    const int p = g[yp * w + xm], q = g[yp * w + x], r = g[yp * w + xp];
    dx = (c - a) + 2 * (f - d) + (r - p);
    dy = (p - a) + 2 * (q - b) + (r - c);
}

__global__ __launch_bounds__(256) void k_hough(const SquareDesc* __restrict__ descs, const u8* __restrict__ gray,
                                                size_t gray_frame_stride, HoughCfg cfg,
                                                cbv_hough_result* __restrict__ out, u8* __restrict__ decisions)
{
    extern __shared__ __align__(16) u8 smem[];
    __shared__ int s_cnt[4]; // 0 weak, 1 edges, 2 centres, 3 circles
    __shared__ int s_over;
    const size_t oi = (size_t)blockIdx.z * CBV_MAX_SQUARES + blockIdx.x;
    if (decisions && !(decisions[oi] & 16)) { // workgroup-uniform: the statistics already decided this square
        if (out && threadIdx.x == 0) {
            cbv_hough_result r;
            memset(&r, 0, sizeof(r));
            r.flags = CBV_HOUGH_SKIPPED;
            out[oi] = r;
        }
        return;
    }
    const SquareDesc d = descs[blockIdx.x];
    const int w = d.w, h = d.h, n = w * h, pw = w + 2, ph = h + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // LDS layout, sized on the host for the largest square of the set (hough_layout)
    u8* g = smem;                                 // [n]      P0..P4
    u8* map = smem + cfg.off_map;                 // [pw*ph]  P2..P4
    u16* mag = (u16*)(smem + cfg.off_mag);        // [pw*ph]  P1..P2, then the edge list
    int* acc = (int*)(smem + cfg.off_acc);        // accumulator; the weak list before P4
    u16* centres = (u16*)(smem + cfg.off_centres);
    int* bins = (int*)(smem + cfg.off_bins);      // [4][max_bins]
    u16* order = (u16*)(smem + cfg.off_order);
    HgCircle* circ = (HgCircle*)smem;             // P6..P7, over g + map (both dead by then)
    u16* weak = (u16*)acc;
    u16* edges = mag;

    const float dp = cfg.dp, idp = 1.f / dp;
    const int min_dim = min(w, h);
    const int min_r = (int)((double)min_dim * cfg.min_ratio), max_r0 = (int)((double)min_dim * cfg.max_ratio);
    const int min_radius = max(min_r, 0);
    const int max_radius = max_r0 <= 0 ? max(w, h) : (max_r0 <= min_radius ? min_radius + 2 : max_r0);
    const int low = max(1, cfg.canny_thr / 2), high = cfg.canny_thr;
    const int arows = (int)ceilf(h * idp), acols = (int)ceilf(w * idp), astep = acols + 2;
    const int acells = (arows + 2) * astep;
    const u32 inv_w = (u32)((0x100000000ull + (u32)w - 1) / (u32)w);
    const u32 inv_ac = (u32)((0x100000000ull + (u32)acols - 1) / (u32)acols);

#ifdef HG_TIMING
    long long tk[10];
    int tki = 0;
#define HG_TICK() do { __syncthreads(); tk[tki++] = __builtin_readcyclecounter(); } while (0)
#else
#define HG_TICK() do { } while (0)
#endif
    HG_TICK();
    const u8* src = gray + (size_t)blockIdx.z * gray_frame_stride + d.plane_off;
    for (int i = tid; i < n; i += 256) g[i] = src[i];
    for (int i = tid; i < pw * ph; i += 256) {
        mag[i] = 0;
        map[i] = 1;
    }
    if (tid < 4) s_cnt[tid] = 0;
    if (tid == 0) s_over = 0;
    __syncthreads();
    HG_TICK();
    // P1
    for (int i = tid; i < n; i += 256) {
        const int y = __umulhi((u32)i, inv_w), x = i - y * w;
        int dx, dy;
        hg_sobel(g, w, h, x, y, dx, dy);
        mag[(y + 1) * pw + x + 1] = (u16)(abs(dx) + abs(dy));
    }
    __syncthreads();
    HG_TICK();
    // P2
    const int TG22 = 13573; // (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5)
    for (int i = tid; i < n; i += 256) {
        const int y = __umulhi((u32)i, inv_w), x = i - y * w;
        const int idx = (y + 1) * pw + x + 1;
        const int m = mag[idx];
        if (m <= low) continue;
        int xs, ys;
        hg_sobel(g, w, h, x, y, xs, ys);
        const int ax = abs(xs), ay = abs(ys) << 15;
        const int tg22x = ax * TG22;
        bool keep;
        if (ay < tg22x) keep = m > mag[idx - 1] && m >= mag[idx + 1];
        else {
            const int tg67x = tg22x + (ax << 16);
            if (ay > tg67x) keep = m > mag[idx - pw] && m >= mag[idx + pw];
            else {
                const int s = (xs ^ ys) < 0 ? -1 : 1;
                keep = m > mag[idx - pw - s] && m > mag[idx + pw + s];
            }
        }
        if (!keep) continue;
        if (m > high) map[idx] = 2;
        else {
            map[idx] = 0;
            weak[atomicAdd(&s_cnt[0], 1)] = (u16)idx;
        }
    }
    __syncthreads();
    HG_TICK();
    // P3: grow strong edges through 8-connected weak candidates until nothing changes
    const int nweak = s_cnt[0];
    for (;;) {
        int changed = 0;
        for (int k = tid; k < nweak; k += 256) {
            const int idx = weak[k];
            if (map[idx] != 0) continue;
            const bool hit = map[idx - pw - 1] == 2 || map[idx - pw] == 2 || map[idx - pw + 1] == 2 || map[idx - 1] == 2 ||
                             map[idx + 1] == 2 || map[idx + pw - 1] == 2 || map[idx + pw] == 2 || map[idx + pw + 1] == 2;
            if (hit) {
                map[idx] = 2;
                changed = 1;
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    HG_TICK();
    // P4: edge list (over the dead magnitude plane), zero the accumulator (over the dead weak list)
    for (int i = tid; i < acells; i += 256) acc[i] = 0;
    for (int i = tid; i < n; i += 256) {
        const int y = __umulhi((u32)i, inv_w), x = i - y * w;
        if (map[(y + 1) * pw + x + 1] == 2) edges[atomicAdd(&s_cnt[1], 1)] = (u16)(x | (y << 8));
    }
    __syncthreads();
    const int nedges = s_cnt[1];
    for (int e = tid; e < nedges; e += 256) {
        const int x = edges[e] & 255, y = edges[e] >> 8;
        int ix, iy;
        hg_sobel(g, w, h, x, y, ix, iy);
        const float vx = (float)ix, vy = (float)iy;
        const float mg = __fsqrt_rn(vx * vx + vy * vy);
        int sx = d_round_f((vx * idp) * 1024.f / mg);
        int sy = d_round_f((vy * idp) * 1024.f / mg);
        const int x0 = d_round_f((x * idp) * 1024.f), y0 = d_round_f((y * idp) * 1024.f);
        for (int k1 = 0; k1 < 2; k1++) {
            int x1 = x0 + min_radius * sx, y1 = y0 + min_radius * sy;
            for (int r = min_radius; r <= max_radius; x1 += sx, y1 += sy, r++) {
                const int x2 = x1 >> 10, y2 = y1 >> 10;
                if ((unsigned)x2 >= (unsigned)acols || (unsigned)y2 >= (unsigned)arows) break;
                atomicAdd(&acc[y2 * astep + x2], 1);
            }
            sx = -sx;
            sy = -sy;
        }
    }
    __syncthreads();
    HG_TICK();
    // P5
    for (int i = tid; i < arows * acols; i += 256) {
        const int yy = __umulhi((u32)i, inv_ac), xx = i - yy * acols;
        const int base = (yy + 1) * astep + xx + 1;
        const int a = acc[base];
        if (a > cfg.acc_thr && a > acc[base - 1] && a >= acc[base + 1] && a > acc[base - astep] && a >= acc[base + astep]) {
            const int k = atomicAdd(&s_cnt[2], 1);
            if (k < HG_MAXC) centres[k] = (u16)base;
            else s_over = 1;
        }
    }
    __syncthreads();
    const int ncent = min(s_cnt[2], HG_MAXC);
    HG_TICK();
    // P6: radius histogram of every centre; wave `wave` takes centre c0 + wave
    const int nbins = d_round_f((max_radius - min_radius) / dp * 10);
    const float minR2 = (float)min_radius * min_radius, maxR2 = (float)max_radius * max_radius;
    int* mybins = bins + wave * cfg.max_bins;
    (void)map;
    for (int c0 = 0; c0 < ncent; c0 += 4) {
        const int c = c0 + wave;
        const bool valid = c < ncent;
        for (int b = lane; b < nbins; b += 64) mybins[b] = 0;
        __syncthreads();
        float ccx = 0, ccy = 0;
        if (valid) {
            const int ofs = centres[c];
            const int cy = ofs / astep, cx = ofs - cy * astep;
            ccx = (cx + 0.5f) * dp;
            ccy = (cy + 0.5f) * dp;
            for (int j = lane; j < nedges; j += 64) {
                const float ex = ccx - (float)(edges[j] & 255), ey = ccy - (float)(edges[j] >> 8);
                const float r2 = ex * ex + ey * ey;
                if (minR2 <= r2 && r2 <= maxR2) {
                    const int bin = max(0, min(nbins - 1, d_round_f((__fsqrt_rn(r2) - min_radius) / dp * 10)));
                    atomicAdd(&mybins[bin], 1);
                }
            }
        }
        __syncthreads();
        if (valid && lane == 0) {
            int max_count = 0;
            float r_best = 0;
            for (int j = nbins - 1; j > 0; j--) {
                if (mybins[j]) {
                    const int upbin = j;
                    int cur = 0;
                    for (; j > upbin - 10 && j >= 0; j--) cur += mybins[j];
                    const float r_cur = (upbin + j) / 2.f / 10 * dp + min_radius;
                    if ((cur * r_best >= max_count * r_cur) || (r_best < 1.1920929e-07f && cur >= max_count)) {
                        r_best = r_cur;
                        max_count = cur;
                    }
                }
            }
            if (max_count > cfg.acc_thr) {
                const int k = atomicAdd(&s_cnt[3], 1);
                // candidates live over g/map, which are dead now; every wave is past P4
                circ[k].x = ccx;
                circ[k].y = ccy;
                circ[k].r = r_best;
                circ[k].votes = max_count;
            }
        }
        __syncthreads();
    }
    HG_TICK();
    // P7: rank sort (total order), then minDist suppression and the pick in one thread
    const int ncirc = s_cnt[3];
    for (int i = tid; i < ncirc; i += 256) {
        const HgCircle ci = circ[i];
        int rank = 0;
        for (int j = 0; j < ncirc; j++) rank += (j != i && hg_before(circ[j], ci)) ? 1 : 0;
        order[rank] = (u16)i;
    }
    __syncthreads();
    if (tid == 0) {
        float md = (float)(min_dim / 3);
        if (md < dp) md = dp;
        const float md2 = md * md;
        int kept = 0;
        for (int i = 0; i < ncirc; i++) {
            const HgCircle ci = circ[order[i]];
            bool close = false;
            for (int j = 0; j < kept && !close; j++) {
                const HgCircle cj = circ[order[j]];
                const float ex = cj.x - ci.x, ey = cj.y - ci.y;
                close = ex * ex + ey * ey < md2;
            }
            if (!close) order[kept++] = order[i];
        }
        // nearest circle to (w//2, h//2) inside 0.3 * min_dim (float32, as numpy evaluates it)
        const float max_off = (float)((double)min_dim * 0.3);
        float best = __builtin_inff();
        int pick = -1;
        for (int i = 0; i < kept; i++) {
            const HgCircle ci = circ[order[i]];
            const float ex = ci.x - (float)(w / 2), ey = ci.y - (float)(h / 2);
            const float dist = __fsqrt_rn(ex * ex + ey * ey);
            if (dist < max_off && dist < best) {
                best = dist;
                pick = i;
            }
        }
        u8 found = 0, kind = 0;
        HgCircle pc = {0.f, 0.f, 0.f, 0};
        if (pick >= 0) {
            pc = circ[order[pick]];
            found = 1;
            kind = ((double)(int)pc.r < (double)min_dim * 0.20) ? 2 : 1;
        }
        if (decisions && found) decisions[oi] = decisions[oi] | 1;
        if (out) {
            cbv_hough_result r;
            r.found = found;
            r.kind = kind;
            r.n_circles = (uint16_t)kept;
            r.cx = pc.x;
            r.cy = pc.y;
            r.r = pc.r;
            r.votes = pc.votes;
            r.n_edges = (uint32_t)nedges;
            r.n_centres = (uint16_t)ncent;
            r.flags = (uint16_t)(s_over ? CBV_HOUGH_OVERFLOW : 0);
            for (int i = 0; i < CBV_HOUGH_KEEP; i++) {
                const bool ok = i < kept;
                const HgCircle ci = ok ? circ[order[i]] : HgCircle{0.f, 0.f, 0.f, 0};
                r.circles[i][0] = ci.x;
                r.circles[i][1] = ci.y;
                r.circles[i][2] = ci.r;
                r.circles[i][3] = (float)ci.votes;
            }
#ifdef HG_TIMING
            for (int i = 0; i + 1 < tki && i < 8; i++) r.circles[2 + i / 4][i % 4] = (float)(tk[i + 1] - tk[i]);
            r.circles[4][0] = (float)(__builtin_readcyclecounter() - tk[tki - 1]);
            r.circles[4][1] = (float)nweak;
            r.circles[4][2] = (float)ncirc;
#endif
            out[oi] = r;
        }
    }
}

// LDS layout for squares up to maxw x maxh
static size_t hough_layout(HoughCfg& cfg)
{
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t maxn = (size_t)cfg.maxw * cfg.maxh, maxp = (size_t)(cfg.maxw + 2) * (cfg.maxh + 2);
    const float idp = 1.f / cfg.dp;
    const int arows = (int)ceilf(cfg.maxh * idp), acols = (int)ceilf(cfg.maxw * idp);
    const size_t acells = (size_t)(arows + 2) * (acols + 2);
    const int md = cfg.maxw > cfg.maxh ? cfg.maxw : cfg.maxh;
    cfg.max_bins = (int)((md + 2) / cfg.dp * 10) + 16;
    size_t off = 0;
    cfg.off_map = (int)up16(maxn);
    off = cfg.off_map + up16(maxp);
    if (off < HG_MAXC * sizeof(HgCircle)) off = HG_MAXC * sizeof(HgCircle); // candidates overlay g + map
    cfg.off_mag = (int)off;
    off += up16(maxp * 2);
    cfg.off_acc = (int)off;
    off += up16(acells * 4 > maxp * 2 ? acells * 4 : maxp * 2); // the weak list (u16 a pixel) shares it
    cfg.off_centres = (int)off;
    off += HG_MAXC * 2;
    cfg.off_bins = (int)off;
    off += (size_t)4 * cfg.max_bins * 4;
    cfg.off_order = (int)off;
    off += HG_MAXC * 2;
    return off;
}

int launch_hough(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, size_t gray_frame_stride, HoughCfg cfg,
                 cbv_hough_result* out, u8* decisions, int batch)
{
    if (cfg.maxw < 2 || cfg.maxh < 2 || cfg.maxw > 250 || cfg.maxh > 250)
        return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "HoughCircles stage: squares must be 2..250 px (got %dx%d)", cfg.maxw, cfg.maxh);
    const size_t lds = hough_layout(cfg);
    if (lds > 150 * 1024)
        return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "HoughCircles stage: %dx%d squares do not fit the LDS layout", cfg.maxw, cfg.maxh);
    static size_t attr_set = 0;
    if (lds > 64 * 1024 && lds > attr_set) {
        CBV_HIP(ctx, hipFuncSetAttribute((const void*)k_hough, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = lds;
    }
    prof_begin(ctx, CBV_K_HOUGH);
    hipLaunchKernelGGL(k_hough, dim3(n, 1, batch), dim3(256), lds, ctx->stream, descs, gray, gray_frame_stride, cfg, out, decisions);
    prof_end(ctx, CBV_K_HOUGH);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}
