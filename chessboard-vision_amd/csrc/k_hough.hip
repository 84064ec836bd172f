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

#define HG_MAXC 512 // accumulator maxima / candidate circles the FIRST pass keeps per square; a square with more
                    // (white noise, never a board square) is redone by the second pass, sized for the worst case
#define HG_NT 512   // lanes per workgroup: the phases are chains of LDS round trips; 8 waves hide them as well as 16 did (1.19 -> 0.95 us/frame alone)
#define HG_NW (HG_NT / 64)

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

// wave-aggregated append of up to four items per lane: ONE LDS atomic per wave (all 64 lanes must call it).
// slot[k] = position of item k in the list, or -1.
__device__ __forceinline__ void hg_append4(int* counter, const bool pred[4], int slot[4])
{
    u64 m[4];
    int total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        m[k] = __ballot(pred[k]);
        total += __popcll(m[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) slot[k] = -1;
    if (total == 0) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(counter, total);
    base = __builtin_amdgcn_readfirstlane(base);
    const u64 below = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (pred[k]) slot[k] = base + __popcll(m[k] & below);
        base += __popcll(m[k]);
    }
}

// Sobel at image pixel (x, y) from the padded gray plane (replicated borders are stored)
__device__ __forceinline__ void hg_sobel(const u8* g, int gs, int x, int y, int& dx, int& dy)
{
    const u8* r0 = g + y * gs + 3 + x; // row y-1, column x-1
    const u8* r1 = r0 + gs;
    const u8* r2 = r1 + gs;
    const int a = r0[0], b = r0[1], c = r0[2], d = r1[0], f = r1[2], p = r2[0], q = r2[1], r = r2[2];
    dx = (c - a) + 2 * (f - d) + (r - p);
    dy = (p - a) + 2 * (q - b) + (r - c);
}

__device__ __forceinline__ int hg_sel4(int i, int a, int b, int c, int d) { return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d)); }

__global__ __launch_bounds__(HG_NT) void k_hough(const SquareDesc* __restrict__ descs, const u8* __restrict__ gray,
                                                size_t gray_frame_stride, HoughCfg cfg,
                                                cbv_hough_result* __restrict__ out, u8* __restrict__ decisions,
                                                const u32* __restrict__ work, int nsq, int total_items)
{
    extern __shared__ __align__(16) u8 smem[];
    __shared__ int s_cnt[4]; // 0 weak, 1 edges, 2 centres, 3 circles
    __shared__ int s_over;
    // Work items: with a worklist (pipeline), work[0] = count and work[1 + i] = frame << 8 | square, filled by
    // k_squares_stats for the squares whose has_piece the statistics left open; otherwise every (square, frame).
    // A fixed grid of workgroups strides over the items, so idle workgroups never hold an LDS slot.
    const int n_items = work ? (int)work[0] : total_items;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int sqi = work ? (int)(work[1 + item] & 255u) : item % nsq;
    const int fri = work ? (int)(work[1 + item] >> 8) : item / nsq;
    const size_t oi = (size_t)fri * CBV_MAX_SQUARES + sqi;
    const SquareDesc d = descs[sqi];
    const int w = d.w, h = d.h, n = w * h;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // LDS planes, sized on the host for the largest square of the set (hough_layout).  Rows are padded so that
    // image column 4k starts a dword: gray/map column x is byte 4 + x of a row of gs bytes (rows -1 .. h stored,
    // gray with replicated borders), magnitude column x is element 2 + x of a row of mw u16 (zero borders).
    const int gs = cfg.gs, mw = cfg.mw;
    u8* g = smem;                                 // P0..P4
    u8* map = smem + cfg.off_map;                 // P1 direction class, P2.. 0 weak / 1 none / 2 edge
    u16* mag = (u16*)(smem + cfg.off_mag);        // P1..P2, then the edge list
    int* acc = (int*)(smem + cfg.off_acc);        // accumulator; the weak list before P4
    u16* centres = (u16*)(smem + cfg.off_centres);
    int* bins = (int*)(smem + cfg.off_bins);      // [HG_NW][max_bins]
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
    const int ngx = (w + 3) >> 2, ngroups = ngx * h; // 4-pixel groups of a row / of the square
    const u32 inv_ngx = (u32)((0x100000000ull + (u32)ngx - 1) / (u32)ngx);

#ifdef HG_TIMING
    long long tk[10];
    int tki = 0;
#define HG_TICK() do { __syncthreads(); tk[tki++] = __builtin_readcyclecounter(); } while (0)
#else
#define HG_TICK() do { } while (0)
#endif
    HG_TICK();
    // P0: plane (tight, 16-byte aligned and zero padded to 16) -> padded rows; zero the magnitude plane
    const u32* src = (const u32*)(gray + (size_t)fri * gray_frame_stride + d.plane_off);
    for (int i = tid; i < (n + 3) >> 2; i += HG_NT) {
        const u32 v = src[i];
        int y = __umulhi((u32)(4 * i), inv_w), x = 4 * i - y * w;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            if (4 * i + b < n) g[(y + 1) * gs + 4 + x] = (u8)(v >> (8 * b));
            if (++x == w) {
                x = 0;
                y++;
            }
        }
    }
    {
        uint4* mz = (uint4*)mag;
        const int nq = (cfg.mag_bytes + 15) >> 4;
        for (int i = tid; i < nq; i += HG_NT) mz[i] = make_uint4(0, 0, 0, 0);
        u32* mp = (u32*)map;
        const int nm = ((h + 2) * gs) >> 2;
        for (int i = tid; i < nm; i += HG_NT) mp[i] = 0x01010101u;
    }
    if (tid < 4) s_cnt[tid] = 0;
    if (tid == 0) s_over = 0;
    __syncthreads();
    for (int y = tid; y < h; y += HG_NT) {
        u8* row = g + (y + 1) * gs;
        row[3] = row[4];
        row[4 + w] = row[3 + w];
    }
    __syncthreads();
    for (int x = tid; x < w + 2; x += HG_NT) {
        g[3 + x] = g[gs + 3 + x];
        g[(h + 1) * gs + 3 + x] = g[h * gs + 3 + x];
    }
    __syncthreads();
    HG_TICK();
    // P1: Sobel, L1 magnitude and the non-maximum-suppression direction class, four pixels per lane
    for (int t = tid; t < ngroups; t += HG_NT) {
        const int y = __umulhi((u32)t, inv_ngx), x0 = (t - y * ngx) << 2;
        int S[6], D[6];
        {
            int T[6], M[6], B[6];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const u32* row = (const u32*)(g + (y + r) * gs + x0);
                const u32 d0 = row[0], d1 = row[1], d2 = row[2];
                int* V = r == 0 ? T : (r == 1 ? M : B);
                V[0] = d0 >> 24;
                V[1] = d1 & 255;
                V[2] = (d1 >> 8) & 255;
                V[3] = (d1 >> 16) & 255;
                V[4] = d1 >> 24;
                V[5] = d2 & 255;
            }
#pragma unroll
            for (int i = 0; i < 6; i++) {
                S[i] = T[i] + 2 * M[i] + B[i];
                D[i] = B[i] - T[i];
            }
        }
        u32 mg[4], dirs = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int dx = S[k + 2] - S[k], dy = D[k] + 2 * D[k + 1] + D[k + 2];
            const int ax = abs(dx), ay = abs(dy) << 15;
            const int tg22x = ax * 13573; // (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5)
            const int tg67x = tg22x + (ax << 16);
            const u32 dir = ay < tg22x ? 0u : (ay > tg67x ? 1u : (((dx ^ dy) < 0) ? 3u : 2u));
            const bool in = x0 + k < w;
            mg[k] = in ? (u32)(ax + abs(dy)) : 0u;
            dirs |= dir << (8 * k);
        }
        u32* mrow = (u32*)(mag + (y + 1) * mw + 2 + x0);
        mrow[0] = mg[0] | (mg[1] << 16);
        mrow[1] = mg[2] | (mg[3] << 16);
        *(u32*)(map + (y + 1) * gs + 4 + x0) = dirs;
    }
    __syncthreads();
    HG_TICK();
    // P2: non-maximum suppression in registers
    for (int t0 = 0; t0 < ngroups; t0 += HG_NT) { // uniform trip count: hg_append uses wave ballots
        const int t = t0 + tid;
        const bool act = t < ngroups;
        const int y = act ? __umulhi((u32)t, inv_ngx) : 0, x0 = act ? (t - y * ngx) << 2 : 0;
        int E[3][6];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const u32* row = (const u32*)(mag + (y + r) * mw + x0);
            const u32 a = row[0], b = row[1], c = row[2], e = row[3];
            E[r][0] = a >> 16;
            E[r][1] = b & 0xFFFF;
            E[r][2] = b >> 16;
            E[r][3] = c & 0xFFFF;
            E[r][4] = c >> 16;
            E[r][5] = e & 0xFFFF;
        }
        u32* mp = (u32*)(map + (y + 1) * gs + 4 + x0);
        const u32 dirs = *mp;
        u32 codes = 0;
        bool isweak[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int dir = (dirs >> (8 * k)) & 3, m = E[1][k + 1];
            const int na = hg_sel4(dir, E[1][k], E[0][k + 1], E[0][k], E[0][k + 2]);
            const int nb = hg_sel4(dir, E[1][k + 2], E[2][k + 1], E[2][k + 2], E[2][k]);
            const bool keep = act && x0 + k < w && m > low && m > na && (dir < 2 ? m >= nb : m > nb);
            const u32 code = !keep ? 1u : (m > high ? 2u : 0u);
            codes |= code << (8 * k);
            isweak[k] = code == 0u;
        }
        int slot[4];
        hg_append4(&s_cnt[0], isweak, slot);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (slot[k] >= 0) weak[slot[k]] = (u16)((y + 1) * gs + 4 + x0 + k);
        if (act) *mp = codes;
    }
    __syncthreads();
    HG_TICK();
    // P3: grow strong edges through 8-connected weak candidates until nothing changes.  A chain of weak pixels
    // advances one pixel a sweep, so sweeps are many and short: with few candidates one wave floods alone (LDS
    // operations of a wave are ordered, no barrier a sweep), the others wait at the barrier below.
    const int nweak = s_cnt[0];
    if (nweak <= 256) {
        if (wave == 0) {
            int idx[4];
#pragma unroll
            for (int q = 0; q < 4; q++) idx[q] = lane + 64 * q < nweak ? (int)weak[lane + 64 * q] : -1;
            for (;;) {
                bool ch = false;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (idx[q] < 0) continue;
                    const u8* c = map + idx[q];
                    const int any2 = (c[-gs - 1] | c[-gs] | c[-gs + 1] | c[-1] | c[1] | c[gs - 1] | c[gs] | c[gs + 1]) & 2;
                    if (any2) {
                        map[idx[q]] = 2;
                        idx[q] = -1;
                        ch = true;
                    }
                }
                if (!__ballot(ch)) break;
            }
        }
        __syncthreads();
    } else {
        for (;;) {
            int changed = 0;
            for (int k = tid; k < nweak; k += HG_NT) {
                const int idx = weak[k];
                if (map[idx] != 0) continue;
                const u8* c = map + idx;
                const int any2 = (c[-gs - 1] | c[-gs] | c[-gs + 1] | c[-1] | c[1] | c[gs - 1] | c[gs] | c[gs + 1]) & 2;
                if (any2) {
                    map[idx] = 2;
                    changed = 1;
                }
            }
            if (!__syncthreads_or(changed)) break;
        }
    }
    HG_TICK();
    // P4: zero the accumulator (over the dead weak list), list the edges (over the dead magnitude plane) ...
    {
        uint4* az = (uint4*)acc;
        for (int i = tid; i < (acells + 3) >> 2; i += HG_NT) az[i] = make_uint4(0, 0, 0, 0);
    }
    for (int t0 = 0; t0 < ngroups; t0 += HG_NT) {
        const int t = t0 + tid;
        const bool act = t < ngroups;
        const int y = act ? __umulhi((u32)t, inv_ngx) : 0, x0 = act ? (t - y * ngx) << 2 : 0;
        const u32 codes = act ? *(const u32*)(map + (y + 1) * gs + 4 + x0) : 0x01010101u;
        bool isedge[4];
        int slot[4];
#pragma unroll
        for (int k = 0; k < 4; k++) isedge[k] = ((codes >> (8 * k)) & 255u) == 2u && x0 + k < w;
        hg_append4(&s_cnt[1], isedge, slot);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (slot[k] >= 0) edges[slot[k]] = (u16)((x0 + k) | (y << 8));
    }
    __syncthreads();
    // ... and vote: one lane per (edge, direction) walks r = min_radius..max_radius along the gradient line
    const int nedges = s_cnt[1];
    for (int t = tid; t < 2 * nedges; t += HG_NT) {
        const int e = t >> 1;
        const int x = edges[e] & 255, y = edges[e] >> 8;
        int ix, iy;
        hg_sobel(g, gs, x, y, ix, iy);
        const float vx = (float)ix, vy = (float)iy;
        const float mg = d_sqrt_rn(vx * vx + vy * vy);
        int sx = d_round_f((vx * idp) * 1024.f / mg);
        int sy = d_round_f((vy * idp) * 1024.f / mg);
        if (t & 1) {
            sx = -sx;
            sy = -sy;
        }
        int x1 = d_round_f((x * idp) * 1024.f) + min_radius * sx, y1 = d_round_f((y * idp) * 1024.f) + min_radius * sy;
        for (int r = min_radius; r <= max_radius; x1 += sx, y1 += sy, r++) {
            const int x2 = x1 >> 10, y2 = y1 >> 10;
            if ((unsigned)x2 >= (unsigned)acols || (unsigned)y2 >= (unsigned)arows) break;
            atomicAdd(&acc[y2 * astep + x2], 1);
        }
    }
    __syncthreads();
    HG_TICK();
    // P5
    for (int i = tid; i < arows * acols; i += HG_NT) {
        const int yy = __umulhi((u32)i, inv_ac), xx = i - yy * acols;
        const int base = (yy + 1) * astep + xx + 1;
        const int a = acc[base];
        if (a > cfg.acc_thr && a > acc[base - 1] && a >= acc[base + 1] && a > acc[base - astep] && a >= acc[base + astep]) {
            const int k = atomicAdd(&s_cnt[2], 1);
            if (k < cfg.maxc) centres[k] = (u16)base;
            else s_over = 1;
        }
    }
    __syncthreads();
    HG_TICK();
    if (s_over && cfg.retry) { // workgroup-uniform: hand the square to the second pass, decide nothing here
        if (tid == 0) {
            const u32 k = atomicAdd(&cfg.retry[0], 1u);
            cfg.retry[1 + k] = ((u32)(fri + cfg.retry_frame_base) << 8) | (u32)sqi;
        }
        __syncthreads(); // s_over / s_cnt are reset at the top of the next item
        continue;
    }
    const int ncent = min(s_cnt[2], cfg.maxc);
    // P6: radius of every centre.  Wave `wave` histograms centre c0 + wave into its own bins, turns them into
    // inclusive prefix sums plus "highest non-empty bin <= i"; then 16 lanes of wave 0 walk one centre each the way
    // the reference does: the highest non-empty bin opens a window of 10 bins, the walk resumes two bins below it.
    const int nbins = d_round_f((max_radius - min_radius) / dp * 10);
    const float minR2 = (float)min_radius * min_radius, maxR2 = (float)max_radius * max_radius;
    int* mybins = bins + wave * cfg.max_bins; // counts, then (inclusive prefix sum << 16) | highest non-empty bin <= i
    const int per_lane = (nbins + 63) >> 6;
    for (int c0 = 0; c0 < ncent; c0 += HG_NW) {
        const int c = c0 + wave;
        if (c < ncent) { // wave-uniform; the wave's bins are private, LDS operations of a wave are ordered
            for (int b = lane; b < nbins; b += 64) mybins[b] = 0;
            const int ofs = centres[c];
            const int cy = ofs / astep, cx = ofs - cy * astep;
            const float ccx = (cx + 0.5f) * dp, ccy = (cy + 0.5f) * dp;
            for (int j = lane; j < nedges; j += 64) {
                const float ex = ccx - (float)(edges[j] & 255), ey = ccy - (float)(edges[j] >> 8);
                const float r2 = ex * ex + ey * ey;
                if (minR2 <= r2 && r2 <= maxR2) {
                    const int bin = max(0, min(nbins - 1, d_round_f((d_sqrt_rn(r2) - min_radius) / dp * 10)));
                    atomicAdd(&mybins[bin], 1);
                }
            }
            // lane l owns bins [l * per_lane, (l + 1) * per_lane)
            const int b0 = lane * per_lane, b1 = min(b0 + per_lane, nbins);
            int tot = 0, last = 0;
            for (int b = b0; b < b1; b++) {
                const int v = mybins[b];
                tot += v;
                if (v) last = b; // bin 0 never opens a window: "none" and "bin 0" may share the value 0
            }
            int run = tot, pv = last;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(run, o, WAVE), q = __shfl_up(pv, o, WAVE);
                if (lane >= o) {
                    run += t;
                    pv = max(pv, q);
                }
            }
            int sum = run - tot;                      // exclusive prefix of the lane's chunk
            int prev = __shfl_up(pv, 1, WAVE);        // highest non-empty bin below the chunk
            if (lane == 0) prev = 0;
            for (int b = b0; b < b1; b++) {
                const int v = mybins[b];
                sum += v;
                if (v) prev = b;
                mybins[b] = (sum << 16) | prev; // both < 65536: at most 128 x 128 edges, bins < 64 K
            }
        }
        __syncthreads();
        if (wave == 0 && lane < HG_NW && c0 + lane < ncent) {
            const u32* W = (const u32*)(bins + lane * cfg.max_bins);
            int max_count = 0;
            float r_best = 0;
            int j = nbins - 1;
            u32 wj = W[j];
            while (j > 0) {
                const int up = (int)(wj & 0xFFFFu); // bins (up, j] are empty: prefix(up) == prefix(j)
                if (up < 1) break;
                const int lo = max(up - 10, -1);
                const u32 wlo = lo >= 0 ? W[lo] : 0u, wnext = lo >= 1 ? W[lo - 1] : 0u; // one round trip a window
                const int cur = (int)(wj >> 16) - (int)(wlo >> 16);
                const float r_cur = (up + lo) / 2.f / 10 * dp + min_radius;
                if ((cur * r_best >= max_count * r_cur) || (r_best < 1.1920929e-07f && cur >= max_count)) {
                    r_best = r_cur;
                    max_count = cur;
                }
                j = lo - 1;
                wj = wnext;
            }
            if (max_count > cfg.acc_thr) {
                const int ofs = centres[c0 + lane];
                const int cy = ofs / astep, cx = ofs - cy * astep;
                const int k = atomicAdd(&s_cnt[3], 1);
                // candidates live over g/map, which are dead now; every wave is past P4
                circ[k].x = (cx + 0.5f) * dp;
                circ[k].y = (cy + 0.5f) * dp;
                circ[k].r = r_best;
                circ[k].votes = max_count;
            }
        }
        __syncthreads();
    }
    HG_TICK();
    // P7: rank sort (total order) into `sorted`, then minDist suppression and the pick.  Up to 64 candidates one
    // wave does it in registers (lane i = i-th circle); more fall back to one thread.
    const int ncirc = s_cnt[3];
    HgCircle* sorted = (HgCircle*)(smem + cfg.off_order);
    for (int i = tid; i < ncirc; i += HG_NT) {
        const HgCircle ci = circ[i];
        int rank = 0;
        for (int j = 0; j < ncirc; j++) rank += (j != i && hg_before(circ[j], ci)) ? 1 : 0;
        sorted[rank] = ci;
    }
    __syncthreads();
    if (wave == 0) {
    float md = (float)(min_dim / 3);
    if (md < dp) md = dp;
    const float md2 = md * md;
    const float max_off = (float)((double)min_dim * 0.3); // float32, as numpy evaluates the comparison
    int kept = 0, pick = -1;
    HgCircle pc = {0.f, 0.f, 0.f, 0};
    if (ncirc <= 64) {
        const HgCircle me = lane < ncirc ? sorted[lane] : HgCircle{0.f, 0.f, 0.f, 0};
        bool alive = lane < ncirc;
        for (int i = 0; i < ncirc; i++) {
            // circle i survives iff no earlier survivor is closer than minDist; it then suppresses later ones
            const u64 am = __ballot(alive);
            if (!((am >> i) & 1)) continue;
            const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.x), i));
            const float yi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.y), i));
            const float ex = xi - me.x, ey = yi - me.y;
            if (lane > i && ex * ex + ey * ey < md2) alive = false;
        }
        const u64 am = __ballot(alive);
        kept = __popcll(am);
        const int pos = __popcll(am & ((1ull << lane) - 1ull)); // index among the survivors
        const float ex = me.x - (float)(w / 2), ey = me.y - (float)(h / 2);
        const float dist = d_sqrt_rn(ex * ex + ey * ey);
        const bool cand = alive && dist < max_off;
        float best = cand ? dist : __builtin_inff();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = fminf(best, __shfl_xor(best, o, WAVE));
        const u64 bm = __ballot(cand && dist == best); // first survivor with the smallest distance
        if (bm) {
            const int pl = __builtin_ctzll(bm);
            pick = __builtin_amdgcn_readlane(pos, pl);
            pc.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.x), pl));
            pc.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.y), pl));
            pc.r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.r), pl));
            pc.votes = __builtin_amdgcn_readlane(me.votes, pl);
        }
        if (alive && pos < CBV_HOUGH_KEEP) circ[pos] = me; // survivors in order, for the result record
    } else if (lane == 0) {
        for (int i = 0; i < ncirc; i++) {
            const HgCircle ci = sorted[i];
            bool close = false;
            for (int j = 0; j < kept && !close; j++) {
                const float ex = sorted[j].x - ci.x, ey = sorted[j].y - ci.y;
                close = ex * ex + ey * ey < md2;
            }
            if (!close) sorted[kept++] = ci;
        }
        float best = __builtin_inff();
        for (int i = 0; i < kept; i++) {
            const HgCircle ci = sorted[i];
            const float ex = ci.x - (float)(w / 2), ey = ci.y - (float)(h / 2);
            const float dist = d_sqrt_rn(ex * ex + ey * ey);
            if (dist < max_off && dist < best) {
                best = dist;
                pick = i;
            }
        }
        if (pick >= 0) pc = sorted[pick];
        for (int i = 0; i < CBV_HOUGH_KEEP && i < kept; i++) circ[i] = sorted[i];
    }
    if (ncirc > 64) { // the serial branch ran on lane 0 only
        kept = __builtin_amdgcn_readfirstlane(kept);
        pick = __builtin_amdgcn_readfirstlane(pick);
    }
    if (lane == 0) {
        u8 found = 0, kind = 0;
        if (pick >= 0) {
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
            if (s_over && cfg.overflow_count) atomicAdd(cfg.overflow_count, 1u);
            for (int i = 0; i < CBV_HOUGH_KEEP; i++) {
                const bool ok = i < kept;
                const HgCircle ci = ok ? circ[i] : HgCircle{0.f, 0.f, 0.f, 0};
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
    } // wave 0
    __syncthreads(); // LDS is reused by the next item
    } // items
}

// LDS layout for squares up to maxw x maxh
static size_t hough_layout(HoughCfg& cfg)
{
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    cfg.gs = (cfg.maxw + 11) & ~3;
    cfg.mw = ((cfg.maxw + 3) & ~3) + 4;
    const size_t maxn = (size_t)cfg.maxw * cfg.maxh;
    const size_t gbytes = (size_t)(cfg.maxh + 2) * cfg.gs;
    cfg.mag_bytes = (int)up16((size_t)(cfg.maxh + 2) * cfg.mw * 2);
    const float idp = 1.f / cfg.dp;
    const int arows = (int)ceilf(cfg.maxh * idp), acols = (int)ceilf(cfg.maxw * idp);
    const size_t acells = (size_t)(arows + 2) * (acols + 2);
    // bins of the radius histogram: round((max_radius - min_radius) / dp * 10) for the largest square
    const int md = cfg.maxw > cfg.maxh ? cfg.maxw : cfg.maxh, mind = cfg.maxw < cfg.maxh ? cfg.maxw : cfg.maxh;
    int span = md + 2;
    if (cfg.max_ratio > 0 && cfg.max_ratio <= 4) {
        span = (int)(mind * cfg.max_ratio) - (int)(mind * cfg.min_ratio) + 2;
        if (span < 4) span = 4;
        if (span > md + 2) span = md + 2;
    }
    cfg.max_bins = (int)(span / cfg.dp * 10) + 16;
    size_t off = 0;
    cfg.off_map = (int)up16(gbytes);
    off = cfg.off_map + up16(gbytes);
    if (off < cfg.maxc * sizeof(HgCircle)) off = cfg.maxc * sizeof(HgCircle); // candidates overlay g + map
    cfg.off_mag = (int)off;
    off += (size_t)cfg.mag_bytes; // >= 2 bytes a pixel: the edge list reuses it
    cfg.off_acc = (int)off;
    off += up16(acells * 4 > maxn * 2 ? acells * 4 : maxn * 2); // the weak list (u16 a pixel) shares it
    cfg.off_centres = (int)off;
    off += up16((size_t)cfg.maxc * 2);
    cfg.off_bins = (int)off;
    off += (size_t)HG_NW * cfg.max_bins * 4;
    cfg.off_order = (int)off; // candidates in HoughCircles' order
    off += cfg.maxc * sizeof(HgCircle);
    return off;
}

static int launch_hough_pass(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, size_t gray_frame_stride, HoughCfg cfg,
                             cbv_hough_result* out, u8* decisions, const u32* work, int total, int max_grid)
{
    const size_t lds = hough_layout(cfg);
    if (lds > 150 * 1024)
        return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "HoughCircles stage: %dx%d squares do not fit the LDS layout", cfg.maxw, cfg.maxh);
    if (lds > 64 * 1024 && !ctx->hough_lds_raised) { // once per context: the attribute is an upper bound
        CBV_HIP(ctx, hipFuncSetAttribute((const void*)k_hough, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        ctx->hough_lds_raised = true;
    }
    // as many workgroups as the chip holds at once (LDS-limited), striding over the work items
    const int per_cu = (int)(160 * 1024 / (lds + 1024)) < 2 ? ((int)(160 * 1024 / (lds + 1024)) < 1 ? 1 : (int)(160 * 1024 / (lds + 1024))) : 2;
    int grid = ctx->num_cus * per_cu;
    if (grid > max_grid) grid = max_grid;
    if (grid > total) grid = total;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_hough, dim3(grid), dim3(HG_NT), lds, ctx->stream, descs, gray, gray_frame_stride, cfg, out, decisions, work, n, total);
    CBV_HIP(ctx, hipGetLastError());
    return CBV_OK;
}

int launch_hough(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, size_t gray_frame_stride, HoughCfg cfg,
                 cbv_hough_result* out, u8* decisions, const u32* work, int batch, u32* retry, int retry_frame_base)
{
    if (cfg.maxw < 2 || cfg.maxh < 2 || cfg.maxw > 250 || cfg.maxh > 250)
        return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "HoughCircles stage: squares must be 2..250 px (got %dx%d)", cfg.maxw, cfg.maxh);
    // first pass: small candidate lists (two workgroups per CU); squares that overflow them go to `retry`
    cfg.maxc = HG_MAXC;
    cfg.retry = retry;
    cfg.retry_frame_base = retry_frame_base;
    prof_begin(ctx, CBV_K_HOUGH);
    const int rc = launch_hough_pass(ctx, descs, n, gray, gray_frame_stride, cfg, out, decisions, work, n * batch, 1 << 30);
    prof_end(ctx, CBV_K_HOUGH);
    return rc;
}

int launch_hough_second(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, size_t gray_frame_stride, HoughCfg cfg,
                        cbv_hough_result* out, u8* decisions, const u32* retry, int max_items)
{
    // Second pass over the listed squares only (normally none: the workgroups read a zero count and leave).
    // No two 4-neighbours can both be maxima (a > left and a >= right exclude each other), so half the cells
    // + 1 is room for every possible maximum; larger squares are capped by LDS and can still flag an overflow.
    const float idp = 1.f / (cfg.dp < 1.f ? 1.f : cfg.dp);
    const int cells = (int)ceilf(cfg.maxh * idp) * (int)ceilf(cfg.maxw * idp);
    cfg.maxc = (cells + 1) / 2 + 1;
    cfg.retry = nullptr;
    cfg.retry_frame_base = 0;
    for (;;) {
        HoughCfg probe = cfg;
        if (hough_layout(probe) <= 150 * 1024 || cfg.maxc <= HG_MAXC) break;
        cfg.maxc = cfg.maxc * 3 / 4;
    }
    if (cfg.maxc < HG_MAXC) cfg.maxc = HG_MAXC;
    return launch_hough_pass(ctx, descs, n, gray, gray_frame_stride, cfg, out, decisions, retry, max_items, 32);
}
