// Host-side table builders and the small host-only pieces of the path
// (getPerspectiveTransform's 8x8 LU solve, 3x3 inverse, PieceDetector masks).
// Compiled with -ffp-contract=off: every float/double operation below rounds
// once, in the order written; the only fused operations are explicit fmaf()
// calls where OpenCV's FMA3-dispatched code uses v_fma.
#include <float.h>
#include <limits.h>
#include <math.h>
#include <string.h>

#include "cbv_internal.h"

static inline int round_d(double v) { return (int)lrint(v); }
static inline int round_f(float v) { return (int)lrintf(v); }
static inline u8 sat8(int v) { return (u8)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// Tables of cv2.cvtColor BGR2HSV / BGR2LAB / LAB2BGR for 8-bit images
// (frame_enhancer.py:74,108,120).
void build_static_tabs(StaticTabs* t)
{
    static const double D65[3] = {0.950456, 1.0, 1.088754};
    static const double RGB2XYZ[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160,
                                      0.072169, 0.019334, 0.119193, 0.950227};
    static const double XYZ2RGB[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991,
                                      0.041556, 0.055648, -0.204043, 1.057311};
    t->sdiv[0] = t->hdiv[0] = 0;
    for (int i = 1; i < 256; i++) {
        t->sdiv[i] = round_d((255 << 12) / (1. * i));
        t->hdiv[i] = round_d((180 << 12) / (6. * i));
    }
    // cv::initLabTabs evaluates the Lab tables in softfloat (IEEE binary32, one rounding per operation; the gamma
    // curves in softdouble, narrowed to float before scaling; mulAdd is a fused multiply-add): same steps in float.
    const float k255 = 255.0f, gamma_scale = (float)(255 * (1 << GAMMA_SHIFT));
    for (int i = 0; i < 256; i++) {
        const double xd = (double)((float)i / k255);
        const double lin = xd <= 0.04045 ? xd / 12.92 : pow((xd + 0.055) / (1.0 + 0.055), 2.4);
        t->gamma[i] = (u16)round_f(gamma_scale * (float)lin);
    }
    const float cbrt_step = 1.0f / (k255 * (float)(1 << GAMMA_SHIFT)), cbrt_gain = (float)(1 << LAB_SHIFT2);
    const float lin_limit = 216.0f / 24389.0f, lin_slope = 841.0f / 108.0f, lin_offset = 16.0f / 116.0f;
    for (int i = 0; i < LAB_CBRT_TAB_SIZE_B; i++) {
        const float x = cbrt_step * (float)i;
        const float fx = x < lin_limit ? fmaf(x, lin_slope, lin_offset) : (float)cbrt((double)x);
        t->cbrt[i] = (u16)round_f(cbrt_gain * fx);
    }
    const float inv_step = 1.0f / (float)(INV_GAMMA_TAB_SIZE - 1);
    for (int i = 0; i < INV_GAMMA_TAB_SIZE; i++) {
        const double xd = (double)(inv_step * (float)i);
        const double enc = xd <= 0.0031308 ? xd * 12.92 : pow(xd, 1.0 / 2.4) * (1.0 + 0.055) - 0.055;
        t->inv_gamma[i] = (u16)round_f(k255 * (float)enc);
    }
    for (int i = 0; i < 256; i++) {
        int y, ify;
        if (i <= 20) { // L <= 8: the linear piece of L -> Y
            y = round_f((float)(i * LAB_BASE * 20 * 9) / (float)(17 * 29 * 29 * 29));
            ify = round_f((float)LAB_BASE * (16.0f / 116.0f + (float)(i * 5) / (float)(3 * 17 * 29)));
        } else {
            const float fy = (float)(i * 100 * LAB_BASE) / (float)(255 * 116) + (float)(16 * LAB_BASE) / 116.0f;
            ify = round_f(fy);
            y = round_f(fy * fy * fy / (float)(LAB_BASE * LAB_BASE));
        }
        t->lab_yf[i * 2] = (u16)y;
        t->lab_yf[i * 2 + 1] = (u16)ify;
    }
    for (int i = 0; i < 3; i++) {
        t->fwd[i * 3 + 2] = round_d((1 << LAB_SHIFT) * RGB2XYZ[i * 3 + 0] / D65[i]);
        t->fwd[i * 3 + 1] = round_d((1 << LAB_SHIFT) * RGB2XYZ[i * 3 + 1] / D65[i]);
        t->fwd[i * 3 + 0] = round_d((1 << LAB_SHIFT) * RGB2XYZ[i * 3 + 2] / D65[i]);
        t->inv[i + 0] = round_d((1 << LAB_SHIFT) * XYZ2RGB[i + 0] * D65[i]);
        t->inv[i + 3] = round_d((1 << LAB_SHIFT) * XYZ2RGB[i + 3] * D65[i]);
        t->inv[i + 6] = round_d((1 << LAB_SHIFT) * XYZ2RGB[i + 6] * D65[i]);
    }
}

static inline float np_mod_f32(float a, float b)
{
    float mod = fmodf(a, b);
    if (b == 0.0f) return mod;
    if (mod != 0.0f) {
        if ((b < 0) != (mod < 0)) mod = mod + b;
    } else {
        mod = copysignf(0.0f, b);
    }
    return mod;
}
static inline float clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// The whole numpy section of apply_color_profile (frame_enhancer.py:71-97)
// acts on each uint8 channel value independently (S additionally on the
// radical-mode window of H), so it is tabulated here in float32 exactly as
// numpy evaluates it (python scalars are weak: they become float32).
void build_profile_tabs(const cbv_color_profile* p, ProfileTabs* t)
{
    memset(t, 0, sizeof(*t));
    t->enabled = p->enabled;
    t->radical = p->radical_mode ? 1 : 0;
    float a = (float)p->contrast, b = (float)p->brightness;
    for (int i = 0; i < 256; i++) {
        float x = fmaf((float)i, a, b); // cvtabs_32f: v_fma(src, alpha, beta)
        t->csa[i] = sat8(round_f(fabsf(x)));
        float h = (float)i;
        if (p->radical_mode) {
            float hd = fabsf(h - (float)p->target_hue);
            float alt = 180.0f - hd;
            hd = hd < alt ? hd : alt;
            t->hmask[i] = hd < (float)p->hue_window ? 1 : 0;
        }
        float hh = np_mod_f32(h + (float)p->hue_shift, 180.0f);
        t->hmap[i] = (u8)(int)clipf(hh, 0.f, 179.f);
        float v = (float)i * (float)p->val_scale;
        t->vmap[i] = (u8)(int)clipf(v, 0.f, 255.f);
        for (int m = 0; m < 2; m++) {
            float s = (float)i;
            if (p->radical_mode) s = m ? s * 2.0f : s * 0.5f;
            s = s * (float)p->sat_scale;
            t->smap[m][i] = (u8)(int)clipf(s, 0.f, 255.f);
        }
    }
    // HSV2RGB_b (hrange 180) front end per byte: h -> sector, fraction; s, v -> [0,1] floats
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    for (int i = 0; i < 256; i++) {
        float h = (float)t->hmap[i] * (6.0f / 180.0f);
        h = fmodf(h, 6.f);
        int sector = (int)floorf(h);
        h = h - (float)sector;
        if ((unsigned)sector >= 6u) {
            sector = 0;
            h = 0.f;
        }
        t->hfr[i] = h;
        t->hsel[i] = (u32)sector_data[sector][0] | ((u32)sector_data[sector][1] << 8) | ((u32)sector_data[sector][2] << 16) | (0x0cu << 24);
        t->v_f[i] = (float)t->vmap[i] * (1.0f / 255.0f);
        for (int m = 0; m < 2; m++) t->s_f[m][i] = (float)t->smap[m][i] * (1.0f / 255.0f);
    }
}

// cv2.bilateralFilter weights (frame_enhancer.py:131)
int build_bilateral_tabs(int d, double sigma_color, double sigma_space, BilateralTabs* t)
{
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    double gcc = -0.5 / (sigma_color * sigma_color);
    double gsc = -0.5 / (sigma_space * sigma_space);
    int radius = d <= 0 ? round_d(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    if (radius > 5) return -1; // (2r+1)^2 taps must fit the 128-entry tables
    for (int i = 0; i < 768; i++) t->color_w[i] = (float)exp(i * i * gcc);
    int maxk = 0;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            t->space_w[maxk] = (float)exp(r * r * gsc);
            t->dy[maxk] = (signed char)i;
            t->dx[maxk] = (signed char)j;
            maxk++;
        }
    t->maxk = maxk;
    t->radius = radius;
    // folded tables (k_bilateral.hip): one per distinct squared distance
    t->ncls = 0;
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 16; j++) t->tap_off[i][j] = -1;
    if (radius <= 4) {
        int r2_of_cls[CBV_BL_MAXCLS];
        for (int k = 0; k < maxk; k++) {
            const int r2 = t->dy[k] * t->dy[k] + t->dx[k] * t->dx[k];
            int c = 0;
            while (c < t->ncls && r2_of_cls[c] != r2) c++;
            if (c == t->ncls) {
                if (t->ncls == CBV_BL_MAXCLS) return -1;
                r2_of_cls[t->ncls++] = r2;
                for (int i = 0; i < 768; i++) t->folded[c][i] = t->space_w[k] * t->color_w[i];  // float * float, rounded once
            }
            t->tap_off[t->dy[k] + radius][t->dx[k] + radius] = c * 768;
        }
    }
    return 0;
}

// cv2.GaussianBlur((k,k), sigma) 8-bit kernel in 8.8 fixed point for an explicit sigma > 0: exp kernel, normalised,
// rounded with error diffusion so that the taps sum to 256 (OpenCV's fixed-point Gaussian kernel)
void build_gaussian_q8_sigma(int k, double sigma, int* coef)
{
    if (!(sigma > 0)) {
        build_gaussian_q8(k, coef);
        return;
    }
    const double scale2 = -0.5 / (sigma * sigma);
    std::vector<double> cf(k);
    double sum = 0;
    for (int i = 0; i < k; i++) {
        const double x = i - (k - 1) * 0.5;
        cf[i] = exp(scale2 * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    const int n2 = k / 2;
    double err = 0;
    long long s = 0;
    for (int i = 0; i < n2; i++) {
        const double adj = cf[i] * sum * 256.0 + err;
        const long long v0 = (long long)floor(adj + 0.5);
        err = adj - (double)v0;
        coef[i] = coef[k - 1 - i] = (int)v0;
        s += v0;
    }
    coef[n2] = (int)(256 - 2 * s);
}

// cv2.GaussianBlur((k,k),0) 8-bit kernel in 8.8 fixed point
void build_gaussian_q8(int k, int* coef)
{
    static const double small_tab[4][7] = {{1.},
                                           {0.25, 0.5, 0.25},
                                           {0.0625, 0.25, 0.375, 0.25, 0.0625},
                                           {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125}};
    if (k <= 7) {
        for (int i = 0; i < k; i++) coef[i] = (int)(small_tab[k >> 1][i] * 256);
        return;
    }
    double sigma = ((k - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2 = -0.5 / (sigma * sigma);
    std::vector<double> cf(k);
    double sum = 0;
    for (int i = 0; i < k; i++) {
        double x = i - (k - 1) * 0.5;
        cf[i] = exp(scale2 * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < k; i++) cf[i] *= sum;
    int n2 = k / 2;
    double err = 0;
    long long s = 0;
    for (int i = 0; i < n2; i++) {
        double adj = cf[i] * 256.0 + err;
        long long v0 = (long long)floor(adj + 0.5);
        err = adj - (double)v0;
        coef[i] = (int)v0;
        coef[k - 1 - i] = (int)v0;
        s += v0;
    }
    coef[n2] = (int)(256 - 2 * s);
}

// Region membership of PieceDetector._detect_center_vs_border and
// _analyze_radial_symmetry (piece_detector.py:141-207):
// bit0 centre disc, bit1 corners, bits 2..5 the four rings.
void build_piece_mask(int w, int h, u8* mask)
{
    int cy = h / 2, cx = w / 2;
    int mn = h < w ? h : w;
    int radius = mn / 4, corner = mn / 4;
    static const double ratios[4] = {0.15, 0.25, 0.35, 0.45};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            u8 m = 0;
            long long d2 = (long long)(x - cx) * (x - cx) + (long long)(y - cy) * (y - cy);
            if (d2 <= (long long)radius * radius) m |= 1;
            bool in_r = (y < corner) || (y >= h - corner);
            bool in_c = (x < corner) || (x >= w - corner);
            if (corner > 0 && in_r && in_c) m |= 2;
            double dist = sqrt((double)d2);
            for (int k = 0; k < 4; k++) {
                double r = mn * ratios[k];
                if (dist >= r - 5 && dist <= r + 5) m |= (u8)(4 << k);
            }
            mask[(size_t)y * w + x] = m;
        }
}

void square_region_counts(const u8* mask, int n, u32 cnt[6])
{
    for (int k = 0; k < 6; k++) cnt[k] = 0;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 6; k++) cnt[k] += (mask[i] >> k) & 1u;
}

// cv::invert for a 3x3 double matrix (closed form), as warpPerspective uses
// it when WARP_INVERSE_MAP is not set (board_detection.py:70).
int host_invert3x3(const double* S, double* D)
{
#define Sd(r, c) S[(r) * 3 + (c)]
    double d = Sd(0, 0) * (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) -
               Sd(0, 1) * (Sd(1, 0) * Sd(2, 2) - Sd(1, 2) * Sd(2, 0)) +
               Sd(0, 2) * (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0));
    if (d == 0.) return 0;
    d = 1. / d;
    double t[9];
    t[0] = (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) * d;
    t[1] = (Sd(0, 2) * Sd(2, 1) - Sd(0, 1) * Sd(2, 2)) * d;
    t[2] = (Sd(0, 1) * Sd(1, 2) - Sd(0, 2) * Sd(1, 1)) * d;
    t[3] = (Sd(1, 2) * Sd(2, 0) - Sd(1, 0) * Sd(2, 2)) * d;
    t[4] = (Sd(0, 0) * Sd(2, 2) - Sd(0, 2) * Sd(2, 0)) * d;
    t[5] = (Sd(0, 2) * Sd(1, 0) - Sd(0, 0) * Sd(1, 2)) * d;
    t[6] = (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0)) * d;
    t[7] = (Sd(0, 1) * Sd(2, 0) - Sd(0, 0) * Sd(2, 1)) * d;
    t[8] = (Sd(0, 0) * Sd(1, 1) - Sd(0, 1) * Sd(1, 0)) * d;
#undef Sd
    memcpy(D, t, sizeof(t));
    return 1;
}

// cv2.getPerspectiveTransform (board_detection.py:69): 8x8 system solved by
// Gaussian elimination with partial pivoting (DECOMP_LU).
extern "C" int cbv_get_perspective_transform(const float* src, const float* dst, double* M)
{
    if (!src || !dst || !M) {
        g_cbv_err = "cbv_get_perspective_transform: null argument";
        return CBV_ERR_ARG;
    }
    const int m = 8;
    double A[64], b[8];
    memset(A, 0, sizeof(A));
    for (int i = 0; i < 4; i++) {
        double sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        A[i * 8 + 0] = A[(i + 4) * 8 + 3] = sx;
        A[i * 8 + 1] = A[(i + 4) * 8 + 4] = sy;
        A[i * 8 + 2] = A[(i + 4) * 8 + 5] = 1;
        A[i * 8 + 6] = -sx * dx;
        A[i * 8 + 7] = -sy * dx;
        A[(i + 4) * 8 + 6] = -sx * dy;
        A[(i + 4) * 8 + 7] = -sy * dy;
        b[i] = dx;
        b[i + 4] = dy;
    }
    bool ok = true;
    for (int i = 0; i < m && ok; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
        if (fabs(A[k * m + i]) < DBL_EPSILON * 100) {
            ok = false;
            break;
        }
        if (k != i) {
            for (int j = i; j < m; j++) {
                double t = A[i * m + j];
                A[i * m + j] = A[k * m + j];
                A[k * m + j] = t;
            }
            double t = b[i];
            b[i] = b[k];
            b[k] = t;
        }
        double d = -1 / A[i * m + i];
        for (int j = i + 1; j < m; j++) {
            double alpha = A[j * m + i] * d;
            for (int c = i + 1; c < m; c++) {
                double t = alpha * A[i * m + c];
                A[j * m + c] = A[j * m + c] + t;
            }
            double t = alpha * b[i];
            b[j] = b[j] + t;
        }
    }
    if (ok) {
        for (int i = m - 1; i >= 0; i--) {
            double s = b[i];
            for (int k = i + 1; k < m; k++) {
                double t = A[i * m + k] * b[k];
                s = s - t;
            }
            b[i] = s / A[i * m + i];
        }
    }
    for (int i = 0; i < 8; i++) M[i] = ok ? b[i] : 0.;
    M[8] = 1.;
    return CBV_OK;
}

ClaheGeom clahe_geom(int w, int h, double clip_limit, int tiles_x, int tiles_y)
{
    ClaheGeom g;
    g.tiles_x = tiles_x;
    g.tiles_y = tiles_y;
    int ext_w = w, ext_h = h;
    g.divisible = (w % tiles_x == 0 && h % tiles_y == 0);
    if (!g.divisible) {
        // CLAHE_Impl::apply pads both dimensions whenever either is not divisible
        ext_h = h + (tiles_y - (h % tiles_y));
        ext_w = w + (tiles_x - (w % tiles_x));
    }
    g.tw = ext_w / tiles_x;
    g.th = ext_h / tiles_y;
    int area = g.tw * g.th;
    g.lut_scale = (float)255 / area;
    g.clip = 0;
    if (clip_limit > 0.0) {
        g.clip = (int)(clip_limit * area / 256);
        if (g.clip < 1) g.clip = 1;
    }
    return g;
}
