/*
 * cbv_oracle.c — CPU restatement of the chessboard-vision digitisation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (chessboard-vision_amd/)
 * may import, link or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS.  The reference (hericmr/chessboard-vision) does all pixel
 * arithmetic in `opencv-python` (requirements.txt:1, version unpinned), which
 * is absent from /root/reference and cannot be installed here.  This file
 * restates the published OpenCV 4.x algorithms for exactly the calls the
 * reference makes, with the reference's parameters.  It is pinned only by
 *   - the reference's own test_change_detector_regression.py:31-54 case,
 *   - goldens captured from the reference's pure-numpy functions
 *     (tests/golden/, made by tests/golden/make_goldens.py),
 *   - closed-form known-answer tests.
 * For every other stage (colour conversions, CLAHE, bilateral, normalise,
 * warp, Otsu) the status is "PARITY UNPINNED": no OpenCV output exists in
 * this environment to compare with.  That includes HoughCircles
 * (piece_detector.py:232-241), restated at the end of this file.
 *
 * Floating point: compiled with -ffp-contract=off, so nothing is fused unless
 * written as fmaf().  Float ops follow OpenCV 4.x's code paths one rounding
 * per operation; where the AVX2/FMA3-dispatched SIMD body of OpenCV (the one
 * an opencv-python wheel runs on any x86-64 CPU since Haswell) uses a fused
 * multiply-add (v_fma / v_muladd: convertScaleAbs, convertTo inside
 * normalize, the bilateral accumulation), fmaf() is used here too.  The SSE
 * baseline path of the same functions differs by at most 1 LSB on rare pixels.
 *
 * Each function cites the reference file:line whose cv2/numpy call it
 * restates.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <limits.h>

typedef uint8_t u8;
typedef uint16_t u16;

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* helpers                                                             */
/* ------------------------------------------------------------------ */

static inline int cv_round_f(float v) { return (int)lrintf(v); }   /* round-half-even */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline u8 sat_u8_i(int v) { return (u8)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline u8 sat_u8_f(float v) { return sat_u8_i(cv_round_f(v)); }
#define CV_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* ------------------------------------------------------------------ */
/* A1  cv2.convertScaleAbs  (frame_enhancer.py:71)                     */
/* dst = saturate_u8(round(|fma(float(src), alpha, beta)|)), alpha/beta */
/* as float32 (cvtabs_32f: v_fma in the SIMD body).                    */
/* ------------------------------------------------------------------ */
ORC_API void orc_convert_scale_abs(const u8* src, int w, int h, int sstride, int cn,
                                   double alpha, double beta, u8* dst, int dstride)
{
    float a = (float)alpha, b = (float)beta;
    for (int y = 0; y < h; y++) {
        const u8* s = src + (size_t)y * sstride;
        u8* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w * cn; x++) {
            float t = fmaf((float)s[x], a, b);
            d[x] = sat_u8_f(fabsf(t));
        }
    }
}

/* ------------------------------------------------------------------ */
/* A2  cv2.cvtColor(BGR2HSV) 8-bit, H in [0,180)  (frame_enhancer.py:74) */
/* RGB2HSV_b: integer tables sdiv/hdiv with hsv_shift = 12.            */
/* ------------------------------------------------------------------ */
static int g_sdiv[256], g_hdiv180[256];
static int g_hsv_init = 0;
static void hsv_init(void)
{
    if (g_hsv_init) return;
    g_sdiv[0] = g_hdiv180[0] = 0;
    for (int i = 1; i < 256; i++) {
        g_sdiv[i] = cv_round_d((255 << 12) / (1. * i));
        g_hdiv180[i] = cv_round_d((180 << 12) / (6. * i));
    }
    g_hsv_init = 1;
}

static inline void bgr2hsv_px(int b, int g, int r, u8* out)
{
    const int hsv_shift = 12;
    int h, s, v = b, vmin = b, vr, vg;
    if (g > v) v = g;
    if (r > v) v = r;
    if (g < vmin) vmin = g;
    if (r < vmin) vmin = r;
    int diff = v - vmin;
    vr = v == r ? -1 : 0;
    vg = v == g ? -1 : 0;
    s = (diff * g_sdiv[v] + (1 << (hsv_shift - 1))) >> hsv_shift;
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * g_hdiv180[diff] + (1 << (hsv_shift - 1))) >> hsv_shift;
    h += h < 0 ? 180 : 0;
    out[0] = sat_u8_i(h);
    out[1] = (u8)s;
    out[2] = (u8)v;
}

ORC_API void orc_bgr2hsv(const u8* src, int w, int h, int sstride, u8* dst, int dstride)
{
    hsv_init();
    for (int y = 0; y < h; y++) {
        const u8* s = src + (size_t)y * sstride;
        u8* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) bgr2hsv_px(s[3 * x], s[3 * x + 1], s[3 * x + 2], d + 3 * x);
    }
}

/* ------------------------------------------------------------------ */
/* A4  cv2.cvtColor(HSV2BGR) 8-bit (frame_enhancer.py:99)              */
/* HSV2RGB_b scalar path: float HSV2RGB_native with hscale = 6/180.    */
/* ------------------------------------------------------------------ */
static inline void hsv2bgr_px(int hh, int ss, int vv, u8* out)
{
    static const int sector_data[][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    const float hscale = 6.0f / 180.0f;
    float h = (float)hh;
    float s = (float)ss * (1.0f / 255.0f);
    float v = (float)vv * (1.0f / 255.0f);
    float b, g, r;
    if (s == 0) {
        b = g = r = v;
    } else {
        float tab[4];
        int sector;
        h = h * hscale;
        h = fmodf(h, 6.f);
        sector = cv_floor_f(h);
        h = h - (float)sector;
        if ((unsigned)sector >= 6u) {
            sector = 0;
            h = 0.f;
        }
        tab[0] = v;
        {
            float t1 = 1.f - s;
            tab[1] = v * t1;
            float sh = s * h;
            float t2 = 1.f - sh;
            tab[2] = v * t2;
            float omh = 1.f - h;
            float s3 = s * omh;
            float t3 = 1.f - s3;
            tab[3] = v * t3;
        }
        b = tab[sector_data[sector][0]];
        g = tab[sector_data[sector][1]];
        r = tab[sector_data[sector][2]];
    }
    out[0] = sat_u8_f(b * 255.0f);
    out[1] = sat_u8_f(g * 255.0f);
    out[2] = sat_u8_f(r * 255.0f);
}

ORC_API void orc_hsv2bgr(const u8* src, int w, int h, int sstride, u8* dst, int dstride)
{
    for (int y = 0; y < h; y++) {
        const u8* s = src + (size_t)y * sstride;
        u8* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) hsv2bgr_px(s[3 * x], s[3 * x + 1], s[3 * x + 2], d + 3 * x);
    }
}

/* ------------------------------------------------------------------ */
/* a2  ImageEnhancer.apply_color_profile (frame_enhancer.py:56-99)     */
/* convertScaleAbs -> BGR2HSV -> float32 numpy section -> HSV2BGR.     */
/* numpy section: float32 arrays with weak python scalars; `%` is      */
/* floor-mod; astype(uint8) truncates toward zero.                     */
/* ------------------------------------------------------------------ */
typedef struct {
    double hue_shift, sat_scale, val_scale, contrast, brightness;
    int radical_mode;
    double target_hue, hue_window;
    int enabled; /* 0 = profile {} -> no-op (frame_enhancer.py:57-58) */
} orc_profile;

static inline float np_mod_f32(float a, float b)
{
    /* numpy npy_divmodf remainder part */
    float mod = fmodf(a, b);
    if (b == 0.0f) return mod;
    if (mod != 0.0f) {
        if ((b < 0) != (mod < 0)) mod = mod + b;
    } else {
        mod = copysignf(0.0f, b);
    }
    return mod;
}

static inline void profile_hsv_adjust(const orc_profile* p, u8* hsv)
{
    float h = (float)hsv[0], s = (float)hsv[1], v = (float)hsv[2];
    if (p->radical_mode) {
        float hd = fabsf(h - (float)p->target_hue);
        float alt = 180.0f - hd;
        hd = hd < alt ? hd : alt;
        if (hd < (float)p->hue_window) s = s * 2.0f;
        else s = s * 0.5f;
    }
    h = np_mod_f32(h + (float)p->hue_shift, 180.0f);
    s = s * (float)p->sat_scale;
    v = v * (float)p->val_scale;
    h = h < 0.f ? 0.f : (h > 179.f ? 179.f : h);
    s = s < 0.f ? 0.f : (s > 255.f ? 255.f : s);
    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    hsv[0] = (u8)(int)h;
    hsv[1] = (u8)(int)s;
    hsv[2] = (u8)(int)v;
}

ORC_API void orc_apply_color_profile(const u8* src, int w, int h, int sstride, const orc_profile* p,
                                     u8* dst, int dstride)
{
    hsv_init();
    float a = (float)p->contrast, bta = (float)p->brightness;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        const u8* s = src + (size_t)y * sstride;
        u8* d = dst + (size_t)y * dstride;
        if (!p->enabled) {
            memcpy(d, s, (size_t)w * 3);
            continue;
        }
        for (int x = 0; x < w; x++) {
            u8 c[3], hsv[3];
            for (int k = 0; k < 3; k++) {
                float t = fmaf((float)s[3 * x + k], a, bta);
                c[k] = sat_u8_f(fabsf(t));
            }
            bgr2hsv_px(c[0], c[1], c[2], hsv);
            profile_hsv_adjust(p, hsv);
            hsv2bgr_px(hsv[0], hsv[1], hsv[2], d + 3 * x);
        }
    }
}

/* ------------------------------------------------------------------ */
/* A5  cv2.cvtColor(BGR2LAB / LAB2BGR) 8-bit (frame_enhancer.py:108,120) */
/* RGB2Lab_b integer path and Lab2RGBinteger (bit-exact mode).         */
/* Tables are built in double here (OpenCV builds them in softfloat);  */
/* an entry may differ by 1 unit of its fixed-point scale.              */
/* ------------------------------------------------------------------ */
enum { LAB_SHIFT = 12, GAMMA_SHIFT = 3, LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT,
       LAB_CBRT_TAB_SIZE_B = 256 * 3 / 2 * (1 << GAMMA_SHIFT),
       INV_GAMMA_SHIFT = 12, INV_GAMMA_TAB_SIZE = 1 << INV_GAMMA_SHIFT,
       LAB_BASE_SHIFT = 14, LAB_BASE = 1 << LAB_BASE_SHIFT, MIN_AB = -8145 };

static u16 g_srgb_gamma[256];
static u16 g_lab_cbrt[LAB_CBRT_TAB_SIZE_B];
static u16 g_inv_gamma[INV_GAMMA_TAB_SIZE];
static int g_lab_to_yf[512];
static int g_fwd_coeffs[9], g_inv_coeffs[9];
static int g_lab_init = 0;

static const double kD65[3] = {0.950456, 1.0, 1.088754};
static const double kRGB2XYZ[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160,
                                   0.072169, 0.019334, 0.119193, 0.950227};
static const double kXYZ2RGB[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991,
                                   0.041556, 0.055648, -0.204043, 1.057311};

static void lab_init(void)
{
    if (g_lab_init) return;
    /* OpenCV's initLabTabs computes these tables in `softfloat` = IEEE binary32, one rounding per operation
     * (softdouble inside applyGamma / applyInvGamma), mulAdd = fmaf.  Same operations, same order, in float here.
     * Against an all-double evaluation this changes 4 of the 3072 cube-root entries by one unit (the float
     * rounding of x = scale * i) and nothing else.  cv::cbrt(softfloat) itself is OpenCV's own polynomial
     * algorithm; the correctly rounded cbrt stands in for it (PARITY UNPINNED like every OpenCV-side number). */
    const float f255 = 255.0f;
    const float int_scale = (float)(255 * (1 << GAMMA_SHIFT));
    for (int i = 0; i < 256; i++) {
        const float x = (float)i / f255;
        const double xd = (double)x;
        const double g = xd <= 0.04045 ? xd / 12.92 : pow((xd + 0.055) / (1.0 + 0.055), 2.4);
        g_srgb_gamma[i] = (u16)cv_round_f(int_scale * (float)g);
    }
    {
        const float lthresh = 216.0f / 24389.0f, lscale = 841.0f / 108.0f, lbias = 16.0f / 116.0f;
        const float cb_scale = 1.0f / (f255 * (float)(1 << GAMMA_SHIFT));
        const float lshift2 = (float)(1 << LAB_SHIFT2);
        for (int i = 0; i < LAB_CBRT_TAB_SIZE_B; i++) {
            const float x = cb_scale * (float)i;
            const float f = x < lthresh ? fmaf(x, lscale, lbias) : (float)cbrt((double)x);
            g_lab_cbrt[i] = (u16)cv_round_f(lshift2 * f);
        }
    }
    {
        const float inv_scale = 1.0f / (float)(INV_GAMMA_TAB_SIZE - 1);
        for (int i = 0; i < INV_GAMMA_TAB_SIZE; i++) {
            const float x = inv_scale * (float)i;
            const double xd = (double)x;
            const double g = xd <= 0.0031308 ? xd * 12.92 : pow(xd, 1.0 / 2.4) * (1.0 + 0.055) - 0.055;
            g_inv_gamma[i] = (u16)cv_round_f(f255 * (float)g);
        }
    }
    for (int i = 0; i < 256; i++) {
        int y, ify;
        if (i <= 20) {
            y = cv_round_f((float)(i * LAB_BASE * 20 * 9) / (float)(17 * 29 * 29 * 29));
            ify = cv_round_f((float)LAB_BASE * (16.0f / 116.0f + (float)(i * 5) / (float)(3 * 17 * 29)));
        } else {
            const float fy = (float)(i * 100 * LAB_BASE) / (float)(255 * 116) + (float)(16 * LAB_BASE) / 116.0f;
            ify = cv_round_f(fy);
            y = cv_round_f(fy * fy * fy / (float)(LAB_BASE * LAB_BASE));
        }
        g_lab_to_yf[i * 2] = y;
        g_lab_to_yf[i * 2 + 1] = ify;
    }
    /* forward, BGR source (blueIdx = 0): coeffs[i*3+0] weights src[0] (blue) */
    for (int i = 0; i < 3; i++) {
        g_fwd_coeffs[i * 3 + 2] = cv_round_d((1 << LAB_SHIFT) * kRGB2XYZ[i * 3 + 0] / kD65[i]);
        g_fwd_coeffs[i * 3 + 1] = cv_round_d((1 << LAB_SHIFT) * kRGB2XYZ[i * 3 + 1] / kD65[i]);
        g_fwd_coeffs[i * 3 + 0] = cv_round_d((1 << LAB_SHIFT) * kRGB2XYZ[i * 3 + 2] / kD65[i]);
    }
    /* inverse (blueIdx = 0): coeffs[i + 0] = R row, [i+3] = G row, [i+6] = B row */
    for (int i = 0; i < 3; i++) {
        g_inv_coeffs[i + 0] = cv_round_d((1 << LAB_SHIFT) * kXYZ2RGB[i + 0] * kD65[i]);
        g_inv_coeffs[i + 3] = cv_round_d((1 << LAB_SHIFT) * kXYZ2RGB[i + 3] * kD65[i]);
        g_inv_coeffs[i + 6] = cv_round_d((1 << LAB_SHIFT) * kXYZ2RGB[i + 6] * kD65[i]);
    }
    g_lab_init = 1;
}

static inline void bgr2lab_px(const u8* s, u8* d)
{
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);
    const int* C = g_fwd_coeffs;
    int R = g_srgb_gamma[s[0]], G = g_srgb_gamma[s[1]], B = g_srgb_gamma[s[2]];
    int fX = g_lab_cbrt[CV_DESCALE(R * C[0] + G * C[1] + B * C[2], LAB_SHIFT)];
    int fY = g_lab_cbrt[CV_DESCALE(R * C[3] + G * C[4] + B * C[5], LAB_SHIFT)];
    int fZ = g_lab_cbrt[CV_DESCALE(R * C[6] + G * C[7] + B * C[8], LAB_SHIFT)];
    int L = CV_DESCALE(Lscale * fY + Lshift, LAB_SHIFT2);
    int a = CV_DESCALE(500 * (fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2);
    int b = CV_DESCALE(200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2);
    d[0] = sat_u8_i(L);
    d[1] = sat_u8_i(a);
    d[2] = sat_u8_i(b);
}

static inline int ab_to_xz(int i)
{
    if (i <= 3390) return i * 108 / 841 - LAB_BASE * 16 / 116 * 108 / 841;
    return i * i / LAB_BASE * i / LAB_BASE;
}

static inline void lab2bgr_px(const u8* s, u8* d)
{
    const int shift = LAB_SHIFT + (LAB_BASE_SHIFT - INV_GAMMA_SHIFT);
    int LL = s[0], aa = s[1], bb = s[2];
    int y = g_lab_to_yf[LL * 2], ify = g_lab_to_yf[LL * 2 + 1];
    int adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * LAB_BASE / 500;
    int bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * LAB_BASE / 200 + 1;
    int x = ab_to_xz(ify + adiv);
    int z = ab_to_xz(ify - bdiv);
    const int* C = g_inv_coeffs;
    int ro = CV_DESCALE(C[0] * x + C[1] * y + C[2] * z, shift);
    int go = CV_DESCALE(C[3] * x + C[4] * y + C[5] * z, shift);
    int bo = CV_DESCALE(C[6] * x + C[7] * y + C[8] * z, shift);
    ro = ro < 0 ? 0 : (ro > INV_GAMMA_TAB_SIZE - 1 ? INV_GAMMA_TAB_SIZE - 1 : ro);
    go = go < 0 ? 0 : (go > INV_GAMMA_TAB_SIZE - 1 ? INV_GAMMA_TAB_SIZE - 1 : go);
    bo = bo < 0 ? 0 : (bo > INV_GAMMA_TAB_SIZE - 1 ? INV_GAMMA_TAB_SIZE - 1 : bo);
    d[0] = sat_u8_i(g_inv_gamma[bo]);
    d[1] = sat_u8_i(g_inv_gamma[go]);
    d[2] = sat_u8_i(g_inv_gamma[ro]);
}

ORC_API void orc_bgr2lab(const u8* src, int w, int h, int sstride, u8* dst, int dstride)
{
    lab_init();
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            bgr2lab_px(src + (size_t)y * sstride + 3 * x, dst + (size_t)y * dstride + 3 * x);
}

ORC_API void orc_lab2bgr(const u8* src, int w, int h, int sstride, u8* dst, int dstride)
{
    lab_init();
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            lab2bgr_px(src + (size_t)y * sstride + 3 * x, dst + (size_t)y * dstride + 3 * x);
}

/* ------------------------------------------------------------------ */
/* A6  cv2.createCLAHE(clipLimit, tileGridSize).apply (frame_enhancer.py:36,114) */
/* src/dst: single channel u8.  lut_out (optional): tiles*256 bytes.   */
/* ------------------------------------------------------------------ */
ORC_API void orc_clahe(const u8* src, int w, int h, int sstride, double clip_limit, int tiles_x,
                       int tiles_y, u8* dst, int dstride, u8* lut_out)
{
    int ext_w = w, ext_h = h;
    if (!(w % tiles_x == 0 && h % tiles_y == 0)) {
        ext_h = h + (tiles_y - (h % tiles_y));
        ext_w = w + (tiles_x - (w % tiles_x));
    }
    int tw = ext_w / tiles_x, th = ext_h / tiles_y;
    int area = tw * th;
    float lut_scale = (float)(255) / area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * area / 256);
        if (clip < 1) clip = 1;
    }
    u8* lut = (u8*)malloc((size_t)tiles_x * tiles_y * 256);
#pragma omp parallel for schedule(static)
    for (int ty = 0; ty < tiles_y; ty++) {
        for (int tx = 0; tx < tiles_x; tx++) {
            int hist[256];
            memset(hist, 0, sizeof(hist));
            for (int yy = 0; yy < th; yy++) {
                int sy = reflect101(ty * th + yy, h);
                for (int xx = 0; xx < tw; xx++) {
                    int sx = reflect101(tx * tw + xx, w);
                    hist[src[(size_t)sy * sstride + sx]]++;
                }
            }
            if (clip > 0) {
                int clipped = 0;
                for (int i = 0; i < 256; i++) {
                    if (hist[i] > clip) {
                        clipped += hist[i] - clip;
                        hist[i] = clip;
                    }
                }
                int batch = clipped / 256;
                int residual = clipped - batch * 256;
                for (int i = 0; i < 256; i++) hist[i] += batch;
                if (residual != 0) {
                    int step = 256 / residual;
                    if (step < 1) step = 1;
                    for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
                }
            }
            u8* tl = lut + (size_t)(ty * tiles_x + tx) * 256;
            int sum = 0;
            for (int i = 0; i < 256; i++) {
                sum += hist[i];
                tl[i] = sat_u8_f((float)sum * lut_scale);
            }
        }
    }
    float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        float tyf = (float)y * inv_th - 0.5f;
        int ty1 = cv_floor_f(tyf), ty2 = ty1 + 1;
        float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        const u8* p1 = lut + (size_t)ty1 * tiles_x * 256;
        const u8* p2 = lut + (size_t)ty2 * tiles_x * 256;
        for (int x = 0; x < w; x++) {
            float txf = (float)x * inv_tw - 0.5f;
            int tx1 = cv_floor_f(txf), tx2 = tx1 + 1;
            float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
            int v = src[(size_t)y * sstride + x];
            float a0 = (float)p1[tx1 * 256 + v] * xa1;
            float a1 = (float)p1[tx2 * 256 + v] * xa;
            float b0 = (float)p2[tx1 * 256 + v] * xa1;
            float b1 = (float)p2[tx2 * 256 + v] * xa;
            float ra = a0 + a1;
            float rb = b0 + b1;
            ra = ra * ya1;
            rb = rb * ya;
            float res = ra + rb;
            dst[(size_t)y * dstride + x] = sat_u8_f(res);
        }
    }
    if (lut_out) memcpy(lut_out, lut, (size_t)tiles_x * tiles_y * 256);
    free(lut);
}

/* a3  ImageEnhancer.correct_lighting (frame_enhancer.py:101-120) */
ORC_API void orc_correct_lighting(const u8* src, int w, int h, int sstride, double clip_limit,
                                  int tiles_x, int tiles_y, u8* dst, int dstride)
{
    lab_init();
    u8* lab = (u8*)malloc((size_t)w * h * 3);
    u8* L = (u8*)malloc((size_t)w * h);
    u8* L2 = (u8*)malloc((size_t)w * h);
    orc_bgr2lab(src, w, h, sstride, lab, w * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) L[i] = lab[3 * i];
    orc_clahe(L, w, h, w, clip_limit, tiles_x, tiles_y, L2, w, NULL);
    for (size_t i = 0; i < (size_t)w * h; i++) lab[3 * i] = L2[i];
    orc_lab2bgr(lab, w, h, w * 3, dst, dstride);
    free(lab);
    free(L);
    free(L2);
}

/* ------------------------------------------------------------------ */
/* A7  cv2.bilateralFilter(d, sigmaColor, sigmaSpace) 8UC3            */
/* (frame_enhancer.py:131).  Scalar body of bilateralFilter_8u.        */
/* ------------------------------------------------------------------ */
ORC_API int orc_bilateral_tables(int d, double sigma_color, double sigma_space, float* color_w /*768*/,
                                 float* space_w /*d*d*/, int* ofs_dy, int* ofs_dx)
{
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    double gcc = -0.5 / (sigma_color * sigma_color);
    double gsc = -0.5 / (sigma_space * sigma_space);
    int radius = d <= 0 ? cv_round_d(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    for (int i = 0; i < 256 * 3; i++) color_w[i] = (float)exp(i * i * gcc);
    int maxk = 0;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            space_w[maxk] = (float)exp(r * r * gsc);
            ofs_dy[maxk] = i;
            ofs_dx[maxk] = j;
            maxk++;
        }
    return maxk;
}

ORC_API void orc_bilateral(const u8* src, int w, int h, int sstride, int d, double sigma_color,
                           double sigma_space, u8* dst, int dstride)
{
    int radius = d <= 0 ? cv_round_d((sigma_space <= 0 ? 1 : sigma_space) * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    int dd = 2 * radius + 1;
    float color_w[768];
    float* space_w = (float*)malloc(sizeof(float) * dd * dd);
    int* ody = (int*)malloc(sizeof(int) * dd * dd);
    int* odx = (int*)malloc(sizeof(int) * dd * dd);
    int maxk = orc_bilateral_tables(d, sigma_color, sigma_space, color_w, space_w, ody, odx);
    /* copyMakeBorder(REFLECT_101) */
    int pw = w + 2 * radius, ph = h + 2 * radius;
    u8* tmp = (u8*)malloc((size_t)pw * ph * 3);
    for (int y = 0; y < ph; y++) {
        int sy = reflect101(y - radius, h);
        for (int x = 0; x < pw; x++) {
            int sx = reflect101(x - radius, w);
            memcpy(tmp + ((size_t)y * pw + x) * 3, src + (size_t)sy * sstride + 3 * sx, 3);
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        const u8* sp = tmp + ((size_t)(y + radius) * pw + radius) * 3;
        u8* dp = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            float sb = 0, sg = 0, sr = 0, ws = 0;
            int b0 = sp[3 * x], g0 = sp[3 * x + 1], r0 = sp[3 * x + 2];
            for (int k = 0; k < maxk; k++) {
                const u8* q = sp + ((ptrdiff_t)ody[k] * pw + x + odx[k]) * 3;
                int b = q[0], g = q[1], r = q[2];
                float wgt = space_w[k] * color_w[abs(b - b0) + abs(g - g0) + abs(r - r0)];
                /* v_muladd(v_cvt_f32(b), w, sum_b) in the dispatched SIMD body */
                sb = fmaf((float)b, wgt, sb);
                sg = fmaf((float)g, wgt, sg);
                sr = fmaf((float)r, wgt, sr);
                ws = ws + wgt;
            }
            ws = 1.f / ws;
            dp[3 * x] = (u8)cv_round_f(sb * ws);
            dp[3 * x + 1] = (u8)cv_round_f(sg * ws);
            dp[3 * x + 2] = (u8)cv_round_f(sr * ws);
        }
    }
    free(tmp);
    free(space_w);
    free(ody);
    free(odx);
}

/* ------------------------------------------------------------------ */
/* A8  cv2.filter2D(frame, -1, 3x3 kernel) (frame_enhancer.py:138)     */
/* correlation, anchor centre, REFLECT_101, float accumulate in        */
/* row-major kernel order over non-zero taps, round, saturate.         */
/* ------------------------------------------------------------------ */
ORC_API void orc_filter3x3(const u8* src, int w, int h, int sstride, int cn, const float* k9, u8* dst,
                           int dstride)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            for (int c = 0; c < cn; c++) {
                float s = 0.f;
                for (int i = 0; i < 3; i++) {
                    int sy = reflect101(y + i - 1, h);
                    for (int j = 0; j < 3; j++) {
                        float kv = k9[i * 3 + j];
                        if (kv == 0.f) continue;
                        int sx = reflect101(x + j - 1, w);
                        float t = kv * (float)src[(size_t)sy * sstride + sx * cn + c];
                        s = s + t;
                    }
                }
                dst[(size_t)y * dstride + x * cn + c] = sat_u8_f(s);
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* A9  cv2.normalize(NORM_MINMAX, 0, 255) (frame_enhancer.py:146)      */
/* ------------------------------------------------------------------ */
ORC_API void orc_minmax(const u8* src, int wbytes, int h, int sstride, int* mn, int* mx)
{
    int lo = 255, hi = 0;
    for (int y = 0; y < h; y++) {
        const u8* s = src + (size_t)y * sstride;
        for (int x = 0; x < wbytes; x++) {
            if (s[x] < lo) lo = s[x];
            if (s[x] > hi) hi = s[x];
        }
    }
    *mn = lo;
    *mx = hi;
}

ORC_API void orc_normalize_lut(int smin_i, int smax_i, u8* lut256)
{
    double smin = smin_i, smax = smax_i, dmin = 0, dmax = 255;
    double scale = (dmax - dmin) * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
    double shift = dmin - smin * scale;
    float a = (float)scale, b = (float)shift;
    for (int i = 0; i < 256; i++) {
        float t = fmaf((float)i, a, b); /* cvt_32f: v_fma(src, scale, shift) */
        lut256[i] = sat_u8_f(t);
    }
}

ORC_API void orc_normalize_minmax(const u8* src, int w, int h, int sstride, int cn, u8* dst, int dstride)
{
    int mn, mx;
    u8 lut[256];
    orc_minmax(src, w * cn, h, sstride, &mn, &mx);
    orc_normalize_lut(mn, mx, lut);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w * cn; x++) dst[(size_t)y * dstride + x] = lut[src[(size_t)y * sstride + x]];
}

/* ------------------------------------------------------------------ */
/* A10 cv2.cvtColor(BGR2GRAY) 8-bit: 15-bit coefficients (OpenCV 4.x)  */
/* (frame_enhancer.py:154; change_detector.py:51; piece_detector.py:128) */
/* ------------------------------------------------------------------ */
static inline u8 gray_px(const u8* p) { return (u8)((p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15); }

ORC_API void orc_bgr2gray(const u8* src, int w, int h, int sstride, u8* dst, int dstride)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[(size_t)y * dstride + x] = gray_px(src + (size_t)y * sstride + 3 * x);
}

/* ------------------------------------------------------------------ */
/* A11 cv2.GaussianBlur(gray, (k,k), 0) 8-bit fixed-point (8.8) path   */
/* (frame_enhancer.py:156; change_detector.py:56; piece_detector.py:133) */
/* Returns the 8.8 kernel in coef[k]; k odd.                           */
/* ------------------------------------------------------------------ */
ORC_API void orc_gaussian_kernel_q8(int k, int* coef)
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
    double* cf = (double*)malloc(sizeof(double) * k);
    double sum = 0;
    for (int i = 0; i < k; i++) {
        double x = i - (k - 1) * 0.5;
        cf[i] = exp(scale2 * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < k; i++) cf[i] *= sum;
    /* getGaussianKernelFixedPoint_ED: error diffusion, symmetric, sum == 256 */
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
    free(cf);
}

ORC_API void orc_gaussian_blur(const u8* src, int w, int h, int sstride, int k, u8* dst, int dstride)
{
    if (k <= 1) {
        for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * dstride, src + (size_t)y * sstride, w);
        return;
    }
    int coef[64];
    orc_gaussian_kernel_q8(k, coef);
    int r = k / 2;
    u16* hbuf = (u16*)malloc(sizeof(u16) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            unsigned acc = 0;
            for (int j = 0; j < k; j++) acc += (unsigned)coef[j] * src[(size_t)y * sstride + reflect101(x + j - r, w)];
            hbuf[(size_t)y * w + x] = (u16)(acc > 65535 ? 65535 : acc);
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            unsigned acc = 0;
            for (int j = 0; j < k; j++) acc += (unsigned)coef[j] * hbuf[(size_t)reflect101(y + j - r, h) * w + x];
            unsigned v = (acc + (1u << 15)) >> 16;
            dst[(size_t)y * dstride + x] = (u8)(v > 255 ? 255 : v);
        }
    free(hbuf);
}

/* ------------------------------------------------------------------ */
/* A12 cv2.threshold(THRESH_BINARY + THRESH_OTSU) (frame_enhancer.py:158) */
/* ------------------------------------------------------------------ */
ORC_API int orc_otsu_from_hist(const int* hist, int total)
{
    double mu = 0, scale = 1. / total;
    for (int i = 0; i < 256; i++) mu += i * (double)hist[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; i++) {
        double p_i, q2, mu2, sigma;
        p_i = hist[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        q2 = 1. - q1;
        if (fmin(q1, q2) < FLT_EPSILON || fmax(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        mu2 = (mu - q1 * mu1) / q2;
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    return (int)max_val;
}

/* a7  ImageEnhancer.prepare_analysis (frame_enhancer.py:148-159) */
ORC_API int orc_prepare_analysis(const u8* src, int w, int h, int sstride, u8* gray, u8* binary)
{
    u8* blur = (u8*)malloc((size_t)w * h);
    orc_bgr2gray(src, w, h, sstride, gray, w);
    orc_gaussian_blur(gray, w, h, w, 5, blur, w);
    int hist[256];
    memset(hist, 0, sizeof(hist));
    for (size_t i = 0; i < (size_t)w * h; i++) hist[blur[i]]++;
    int t = orc_otsu_from_hist(hist, w * h);
    for (size_t i = 0; i < (size_t)w * h; i++) binary[i] = blur[i] > t ? 255 : 0;
    free(blur);
    return t;
}

/* a8  ImageEnhancer.process_pipeline (frame_enhancer.py:161-181) */
ORC_API void orc_process_pipeline(const u8* src, int w, int h, int sstride, const orc_profile* prof,
                                  double clip_limit, int tiles_x, int tiles_y, const float* k9, u8* dst,
                                  int dstride)
{
    size_t n = (size_t)w * h * 3;
    u8* a = (u8*)malloc(n);
    u8* b = (u8*)malloc(n);
    orc_apply_color_profile(src, w, h, sstride, prof, a, w * 3);
    orc_correct_lighting(a, w, h, w * 3, clip_limit, tiles_x, tiles_y, b, w * 3);
    orc_bilateral(b, w, h, w * 3, 9, 75, 75, a, w * 3);
    orc_filter3x3(a, w, h, w * 3, 3, k9, b, w * 3);
    orc_normalize_minmax(b, w, h, w * 3, 3, dst, dstride);
    free(a);
    free(b);
}

/* ------------------------------------------------------------------ */
/* A13 cv2.getPerspectiveTransform (DECOMP_LU) + cv2.warpPerspective   */
/* INTER_LINEAR, BORDER_CONSTANT(0)  (board_detection.py:67-70)        */
/* ------------------------------------------------------------------ */
static int lu_solve(double* A, int m, double* b)
{
    int p = 1;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
        if (fabs(A[k * m + i]) < DBL_EPSILON * 100) return 0;
        if (k != i) {
            for (int j = i; j < m; j++) {
                double t = A[i * m + j];
                A[i * m + j] = A[k * m + j];
                A[k * m + j] = t;
            }
            double t = b[i];
            b[i] = b[k];
            b[k] = t;
            p = -p;
        }
        double d = -1 / A[i * m + i];
        for (int j = i + 1; j < m; j++) {
            double alpha = A[j * m + i] * d;
            for (k = i + 1; k < m; k++) {
                double t = alpha * A[i * m + k];
                A[j * m + k] = A[j * m + k] + t;
            }
            double t = alpha * b[i];
            b[j] = b[j] + t;
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < m; k++) {
            double t = A[i * m + k] * b[k];
            s = s - t;
        }
        b[i] = s / A[i * m + i];
    }
    return p;
}

/* src, dst: 4 points (x,y) float32 each. M: 3x3 row-major double. */
ORC_API int orc_get_perspective_transform(const float* src, const float* dst, double* M)
{
    double a[8][8], b[8];
    for (int i = 0; i < 4; i++) {
        double sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        a[i][0] = a[i + 4][3] = sx;
        a[i][1] = a[i + 4][4] = sy;
        a[i][2] = a[i + 4][5] = 1;
        a[i][3] = a[i][4] = a[i][5] = a[i + 4][0] = a[i + 4][1] = a[i + 4][2] = 0;
        a[i][6] = -sx * dx;
        a[i][7] = -sy * dx;
        a[i + 4][6] = -sx * dy;
        a[i + 4][7] = -sy * dy;
        b[i] = dx;
        b[i + 4] = dy;
    }
    int ok = lu_solve(&a[0][0], 8, b);
    for (int i = 0; i < 8; i++) M[i] = ok ? b[i] : 0;
    M[8] = 1.;
    return ok != 0;
}

/* cv::invert 3x3 CV_64F (closed form) */
ORC_API int orc_invert3x3(const double* S, double* D)
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

static inline int sat_int_d(double v)
{
    if (v <= (double)INT_MIN) return INT_MIN;
    if (v >= (double)INT_MAX) return INT_MAX;
    return (int)lrint(v);
}
static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

/* M is the forward (src->dst) matrix as returned by getPerspectiveTransform;
 * it is inverted here like warpPerspective does without WARP_INVERSE_MAP.
 * Coordinates are produced per 64x16 block exactly like WarpPerspectiveInvoker
 * (BLOCK_SZ = 32), then sampled like remapBilinear with INTER_BITS = 5. */
ORC_API void orc_warp_perspective(const u8* src, int sw, int sh, int sstride, const double* Mfwd,
                                  int dw, int dh, u8* dst, int dstride)
{
    double M[9];
    if (!orc_invert3x3(Mfwd, M)) memset(M, 0, sizeof(M));
    const int BLOCK_SZ = 32;
    int bh0 = BLOCK_SZ / 2 < dh ? BLOCK_SZ / 2 : dh;
    int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < dw ? BLOCK_SZ * BLOCK_SZ / bh0 : dw;
    bh0 = BLOCK_SZ * BLOCK_SZ / bw0 < dh ? BLOCK_SZ * BLOCK_SZ / bw0 : dh;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y += bh0) {
        for (int x = 0; x < dw; x += bw0) {
            int bw = bw0 < dw - x ? bw0 : dw - x;
            int bh = bh0 < dh - y ? bh0 : dh - y;
            for (int y1 = 0; y1 < bh; y1++) {
                double X0 = M[0] * x + M[1] * (y + y1) + M[2];
                double Y0 = M[3] * x + M[4] * (y + y1) + M[5];
                double W0 = M[6] * x + M[7] * (y + y1) + M[8];
                for (int x1 = 0; x1 < bw; x1++) {
                    double W = W0 + M[6] * x1;
                    W = W ? 32. / W : 0;
                    double fX = (X0 + M[0] * x1) * W;
                    double fY = (Y0 + M[3] * x1) * W;
                    fX = fmax((double)INT_MIN, fmin((double)INT_MAX, fX));
                    fY = fmax((double)INT_MIN, fmin((double)INT_MAX, fY));
                    int X = sat_int_d(fX), Y = sat_int_d(fY);
                    int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
                    int fx = X & 31, fy = Y & 31;
                    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
                    int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
                    u8* D = dst + (size_t)(y + y1) * dstride + 3 * (x + x1);
                    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
                        D[0] = D[1] = D[2] = 0;
                        continue;
                    }
                    for (int k = 0; k < 3; k++) {
                        int v00 = (sx >= 0 && sy >= 0 && sx < sw && sy < sh) ? src[(size_t)sy * sstride + 3 * sx + k] : 0;
                        int v01 = (sx + 1 >= 0 && sy >= 0 && sx + 1 < sw && sy < sh) ? src[(size_t)sy * sstride + 3 * (sx + 1) + k] : 0;
                        int v10 = (sx >= 0 && sy + 1 >= 0 && sx < sw && sy + 1 < sh) ? src[(size_t)(sy + 1) * sstride + 3 * sx + k] : 0;
                        int v11 = (sx + 1 >= 0 && sy + 1 >= 0 && sx + 1 < sw && sy + 1 < sh) ? src[(size_t)(sy + 1) * sstride + 3 * (sx + 1) + k] : 0;
                        D[k] = sat_u8_i((v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15);
                    }
                }
            }
        }
    }
}

/* bench.py's cpu_baseline leg times the oracle at 1 thread and at all cores: OpenMP team size of the calling thread */
#ifdef _OPENMP
#include <omp.h>
ORC_API int orc_set_threads(int n)
{
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
}
#else
ORC_API int orc_set_threads(int n) { (void)n; return 1; }
#endif

/* A14 cv2.rotate(ROTATE_180) (game_session.py:126) */
ORC_API void orc_rotate180(const u8* src, int w, int h, int sstride, int cn, u8* dst, int dstride)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            memcpy(dst + (size_t)(h - 1 - y) * dstride + (size_t)(w - 1 - x) * cn, src + (size_t)y * sstride + (size_t)x * cn, cn);
}

/* ------------------------------------------------------------------ */
/* a12/a18  ChangeDetector._preprocess (change_detector.py:49-56) and  */
/* PieceDetector._preprocess_square (piece_detector.py:124-135):        */
/* optional BGR2GRAY then GaussianBlur((k,k),0) on the ROI alone.       */
/* out: tight w*h.                                                      */
/* ------------------------------------------------------------------ */
ORC_API void orc_square_preprocess(const u8* roi, int w, int h, int stride, int cn, int blur_k, u8* out)
{
    u8* g = (u8*)malloc((size_t)w * h);
    if (cn == 3) orc_bgr2gray(roi, w, h, stride, g, w);
    else
        for (int y = 0; y < h; y++) memcpy(g + (size_t)y * w, roi + (size_t)y * stride, w);
    orc_gaussian_blur(g, w, h, w, blur_k, out, w);
    free(g);
}

/* ------------------------------------------------------------------ */
/* a19 region masks of PieceDetector (piece_detector.py:141-207):      */
/* bit0 centre disc, bit1 corners, bits 2..5 rings r = min*{.15,.25,.35,.45} */
/* ------------------------------------------------------------------ */
ORC_API void orc_piece_masks(int w, int h, u8* mask)
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
            int in_r = (y < corner) || (y >= h - corner);
            int in_c = (x < corner) || (x >= w - corner);
            if (corner > 0 && in_r && in_c) m |= 2;
            double dist = sqrt((double)d2);
            for (int k = 0; k < 4; k++) {
                double r = mn * ratios[k];
                if (dist >= r - 5 && dist <= r + 5) m |= (u8)(4 << k);
            }
            mask[(size_t)y * w + x] = m;
        }
}

/* Per-square integer statistics consumed by the detectors' host logic. */
typedef struct {
    uint32_t n, sum, sumsq, sad_ref;
    uint32_t center_sum, center_cnt, border_sum, border_cnt;
    uint32_t ring_sum[4], ring_cnt[4];
    uint32_t z_count;
    float z_max;
} orc_sq_stats;

/* gray: preprocessed square (tight w*h).  ref/mean/var may be NULL. */
ORC_API void orc_square_stats(const u8* gray, int w, int h, const u8* ref, const float* mean,
                              const float* var, double z_thresh, orc_sq_stats* st)
{
    memset(st, 0, sizeof(*st));
    u8* mask = (u8*)malloc((size_t)w * h);
    orc_piece_masks(w, h, mask);
    st->n = (uint32_t)(w * h);
    float zmax = 0.f;
    int nan_seen = 0;
    float zt = (float)z_thresh;
    for (int i = 0; i < w * h; i++) {
        int g = gray[i];
        st->sum += g;
        st->sumsq += g * g;
        if (ref) st->sad_ref += (uint32_t)abs(g - (int)ref[i]);
        u8 m = mask[i];
        if (m & 1) { st->center_sum += g; st->center_cnt++; }
        if (m & 2) { st->border_sum += g; st->border_cnt++; }
        for (int k = 0; k < 4; k++)
            if (m & (4 << k)) { st->ring_sum[k] += g; st->ring_cnt[k]++; }
        if (mean && var) {
            /* change_detector.py:131-137,160 in float32 */
            float sd = sqrtf(var[i]);
            float df = fabsf((float)g - mean[i]);
            float z = df / sd;
            if (z > zt) st->z_count++;
            if (z != z) nan_seen = 1; /* np.max propagates NaN */
            else if (z > zmax) zmax = z;
        }
    }
    if (nan_seen) zmax = NAN;
    st->z_max = zmax;
    free(mask);
}

/* a14 ChangeDetector.update_all_references EMA (change_detector.py:77-92), float32 */
ORC_API void orc_ema_update(const u8* gray, int n, double alpha, float* mean, float* var)
{
    /* (1 - self.alpha) is a python double; multiplying a float32 array by it
       converts it to float32 first (weak scalar). */
    float one_minus = (float)(1.0 - alpha);
    float a = (float)alpha;
    for (int i = 0; i < n; i++) {
        float g = (float)gray[i];
        float m1 = one_minus * mean[i];
        float m2 = a * g;
        float nm = m1 + m2;
        float d = g - nm;
        float d2 = d * d;
        float v1 = one_minus * var[i];
        float v2 = a * d2;
        float nv = v1 + v2;
        if (!(nv >= 10.0f)) nv = (nv != nv) ? nv : 10.0f; /* np.maximum propagates NaN */
        mean[i] = nm;
        var[i] = nv;
    }
}

/* ------------------------------------------------------------------ */
/* Synthetic frames (ours).  Deterministic counter hash. */
/* ------------------------------------------------------------------ */
typedef struct {
    u8 bg_lo, bg_span;       /* background value = bg_lo + hash % bg_span, 16x16 cells */
    u8 light[3], dark[3];    /* BGR of light / dark squares */
    u8 white[3], black[3];   /* BGR of white / black pieces */
    u8 noise;                /* uniform noise amplitude: value in [-noise, +noise] */
    u8 pad[3];
    double radius;           /* disc radius in square units */
} orc_scene;

static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* Hinv: frame px -> board units [0,8)x[0,8); board[64]: row-major from rank 8,
 * 0 empty, 1 white piece, 2 black piece. */
ORC_API void orc_synth_frame(uint64_t seed, int w, int h, const double* Hinv, const u8* board,
                             const orc_scene* sc, u8* dst, int dstride)
{
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            uint64_t idx = (uint64_t)y * (uint64_t)w + (uint64_t)x;
            uint64_t hn = mix64(seed + idx * 0x9E3779B97F4A7C15ULL);
            int col[3];
            uint64_t hb = mix64(0x5851F42D4C957F2DULL + (uint64_t)(y >> 4) * 4096 + (uint64_t)(x >> 4));
            int bgv = sc->bg_lo + (int)(hb % (uint64_t)(sc->bg_span ? sc->bg_span : 1));
            col[0] = col[1] = col[2] = bgv;
            double W = Hinv[6] * x + Hinv[7] * y + Hinv[8];
            double u = (Hinv[0] * x + Hinv[1] * y + Hinv[2]) / W;
            double v = (Hinv[3] * x + Hinv[4] * y + Hinv[5]) / W;
            if (u >= 0.0 && u < 8.0 && v >= 0.0 && v < 8.0) {
                int fi = (int)u, ri = (int)v;
                const u8* c = ((fi + ri) & 1) ? sc->dark : sc->light;
                int piece = board[ri * 8 + fi];
                if (piece) {
                    double du = u - (fi + 0.5), dv = v - (ri + 0.5);
                    if (du * du + dv * dv <= sc->radius * sc->radius) c = piece == 1 ? sc->white : sc->black;
                }
                col[0] = c[0];
                col[1] = c[1];
                col[2] = c[2];
            }
            int span = 2 * sc->noise + 1;
            for (int k = 0; k < 3; k++) {
                int n = (int)((hn >> (16 * k)) & 0xFFFF) % span - sc->noise;
                dst[(size_t)y * dstride + 3 * x + k] = sat_u8_i(col[k] + n);
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* A15 cv2.HoughCircles(gray, HOUGH_GRADIENT, dp, minDist, param1,     */
/* param2, minRadius, maxRadius) (piece_detector.py:232-241).          */
/* Restated from the published OpenCV 4.x algorithm (hough.cpp:        */
/* HoughCirclesGradient): Sobel 3x3 (BORDER_REPLICATE) -> Canny on     */
/* (dx, dy) with L1 magnitude, thresholds max(1, param1/2) and param1  */
/* -> gradient-line voting into a 1/dp accumulator (SHIFT = 10 fixed   */
/* point) -> local maxima above param2, sorted by votes -> per centre  */
/* radius histogram (10 bins per dp) -> support > param2 -> sort by    */
/* support, suppress centres closer than minDist.                      */
/* PARITY UNPINNED: no OpenCV output is available to compare with.     */
/* ------------------------------------------------------------------ */
typedef struct { float x, y, r; int votes; } orc_circle;

static void orc_sobel3(const u8* g, int w, int h, short* dx, short* dy)
{
    for (int y = 0; y < h; y++) {
        int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
        for (int x = 0; x < w; x++) {
            int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
            int a = g[ym * w + xm], b = g[ym * w + x], c = g[ym * w + xp];
            int d = g[y * w + xm], f = g[y * w + xp];
            int p = g[yp * w + xm], q = g[yp * w + x], r = g[yp * w + xp];
            dx[y * w + x] = (short)((c - a) + 2 * (f - d) + (r - p));
            dy[y * w + x] = (short)((p - a) + 2 * (q - b) + (r - c));
        }
    }
}

/* Canny(dx, dy, low, high, L2gradient = false): edges[] = 255 / 0 */
static void orc_canny(const short* dx, const short* dy, int w, int h, int low, int high, u8* edges)
{
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
    int pw = w + 2;
    int* mag = (int*)calloc((size_t)pw * (h + 2), sizeof(int));
    u8* map = (u8*)malloc((size_t)pw * (h + 2)); /* 0 weak candidate, 1 not an edge, 2 edge */
    memset(map, 1, (size_t)pw * (h + 2));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) mag[(y + 1) * pw + x + 1] = abs((int)dx[y * w + x]) + abs((int)dy[y * w + x]);
    int* stack = (int*)malloc(sizeof(int) * (size_t)w * h);
    int sp = 0;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int idx = (y + 1) * pw + x + 1;
            int m = mag[idx];
            if (m <= low) continue;
            int xs = dx[y * w + x], ys = dy[y * w + x];
            int ax = abs(xs), ay = abs(ys) << 15;
            int tg22x = ax * TG22;
            int keep = 0;
            if (ay < tg22x) {
                keep = m > mag[idx - 1] && m >= mag[idx + 1];
            } else {
                int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) keep = m > mag[idx - pw] && m >= mag[idx + pw];
                else {
                    int s = (xs ^ ys) < 0 ? -1 : 1;
                    keep = m > mag[idx - pw - s] && m > mag[idx + pw + s];
                }
            }
            if (!keep) continue;
            if (m > high) {
                map[idx] = 2;
                stack[sp++] = idx;
            } else map[idx] = 0;
        }
    }
    while (sp > 0) {
        int idx = stack[--sp];
        static const int dxs[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dys[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        for (int k = 0; k < 8; k++) {
            int n = idx + dys[k] * pw + dxs[k];
            if (map[n] == 0) {
                map[n] = 2;
                stack[sp++] = n;
            }
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) edges[y * w + x] = map[(y + 1) * pw + x + 1] == 2 ? 255 : 0;
    free(mag);
    free(map);
    free(stack);
}

static int cmp_center(const void* a, const void* b, void* acc)
{
    const int* A = (const int*)acc;
    int l1 = *(const int*)a, l2 = *(const int*)b;
    if (A[l1] != A[l2]) return A[l1] > A[l2] ? -1 : 1;
    return l1 < l2 ? -1 : (l1 > l2 ? 1 : 0);
}
static int cmp_circle(const void* a, const void* b)
{
    const orc_circle *L = (const orc_circle*)a, *R = (const orc_circle*)b;
    if (L->votes != R->votes) return L->votes > R->votes ? -1 : 1;
    if (L->r != R->r) return L->r > R->r ? -1 : 1;
    if (L->x != R->x) return L->x < R->x ? -1 : 1;
    if (L->y != R->y) return L->y < R->y ? -1 : 1;
    return 0;
}
static inline int cv_ceil_f(float v) { int i = (int)v; return i + (i < v); }

ORC_API int orc_hough_circles(const u8* gray, int w, int h, double dp_d, double min_dist_d, double param1,
                              double param2, int min_radius, int max_radius, orc_circle* out, int max_out,
                              u8* edges_out)
{
    float dp = (float)dp_d;
    if (dp < 1.f) dp = 1.f;
    const float idp = 1.f / dp;
    int canny_thr = cv_round_d(param1), acc_thr = cv_round_d(param2);
    if (min_radius < 0) min_radius = 0;
    if (max_radius <= 0) max_radius = w > h ? w : h;
    else if (max_radius <= min_radius) max_radius = min_radius + 2;
    short* dx = (short*)malloc(sizeof(short) * (size_t)w * h);
    short* dy = (short*)malloc(sizeof(short) * (size_t)w * h);
    u8* edges = (u8*)malloc((size_t)w * h);
    orc_sobel3(gray, w, h, dx, dy);
    int low = canny_thr / 2;
    if (low < 1) low = 1;
    orc_canny(dx, dy, w, h, low, canny_thr, edges);
    if (edges_out) memcpy(edges_out, edges, (size_t)w * h);
    const int SHIFT = 10, ONE = 1 << SHIFT;
    int arows = cv_ceil_f(h * idp), acols = cv_ceil_f(w * idp), astep = acols + 2;
    int* acc = (int*)calloc((size_t)(arows + 2) * astep, sizeof(int));
    int* nzx = (int*)malloc(sizeof(int) * (size_t)w * h);
    int* nzy = (int*)malloc(sizeof(int) * (size_t)w * h);
    int nz = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (!edges[y * w + x]) continue;
            float vx = dx[y * w + x], vy = dy[y * w + x];
            if (vx == 0 && vy == 0) continue;
            float mag = sqrtf(vx * vx + vy * vy);
            if (mag < 1.0f) continue;
            nzx[nz] = x;
            nzy[nz] = y;
            nz++;
            int sx = cv_round_f((vx * idp) * ONE / mag);
            int sy = cv_round_f((vy * idp) * ONE / mag);
            int x0 = cv_round_f((x * idp) * ONE);
            int y0 = cv_round_f((y * idp) * ONE);
            for (int k1 = 0; k1 < 2; k1++) {
                int x1 = x0 + min_radius * sx, y1 = y0 + min_radius * sy;
                for (int r = min_radius; r <= max_radius; x1 += sx, y1 += sy, r++) {
                    int x2 = x1 >> SHIFT, y2 = y1 >> SHIFT;
                    if ((unsigned)x2 >= (unsigned)acols || (unsigned)y2 >= (unsigned)arows) break;
                    acc[y2 * astep + x2]++;
                }
                sx = -sx;
                sy = -sy;
            }
        }
    int ncirc = 0;
    orc_circle* circ = NULL;
    if (nz > 0) {
        int* centers = (int*)malloc(sizeof(int) * (size_t)(arows + 2) * astep);
        int nc = 0;
        for (int y = 1; y < arows + 1; y++)
            for (int x = 1; x <= acols; x++) {
                int base = y * astep + x;
                if (acc[base] > acc_thr && acc[base] > acc[base - 1] && acc[base] >= acc[base + 1] &&
                    acc[base] > acc[base - astep] && acc[base] >= acc[base + astep])
                    centers[nc++] = base;
            }
        qsort_r(centers, nc, sizeof(int), cmp_center, acc);
        const int nBinsPerDr = 10;
        const int nBins = cv_round_f((max_radius - min_radius) / dp * nBinsPerDr);
        int* bins = (int*)malloc(sizeof(int) * (nBins > 0 ? nBins : 1));
        circ = (orc_circle*)malloc(sizeof(orc_circle) * (nc > 0 ? nc : 1));
        const float minR2 = (float)min_radius * min_radius, maxR2 = (float)max_radius * max_radius;
        for (int i = 0; i < nc; i++) {
            int ofs = centers[i];
            int cy = ofs / astep, cx = ofs - cy * astep;
            float ccx = (cx + 0.5f) * dp, ccy = (cy + 0.5f) * dp;
            int max_count = 0;
            float r_best = 0;
            int cnt = 0;
            memset(bins, 0, sizeof(int) * (nBins > 0 ? nBins : 1));
            for (int j = 0; j < nz; j++) {
                float ddx = ccx - nzx[j], ddy = ccy - nzy[j];
                float r2 = ddx * ddx + ddy * ddy;
                if (minR2 <= r2 && r2 <= maxR2) {
                    int bin = cv_round_f((sqrtf(r2) - min_radius) / dp * nBinsPerDr);
                    if (bin > nBins - 1) bin = nBins - 1;
                    if (bin < 0) bin = 0;
                    bins[bin]++;
                    cnt++;
                }
            }
            if (cnt) {
                for (int j = nBins - 1; j > 0; j--) {
                    if (bins[j]) {
                        int upbin = j, cur = 0;
                        for (; j > upbin - nBinsPerDr && j >= 0; j--) cur += bins[j];
                        float r_cur = (upbin + j) / 2.f / nBinsPerDr * dp + min_radius;
                        if ((cur * r_best >= max_count * r_cur) || (r_best < FLT_EPSILON && cur >= max_count)) {
                            r_best = r_cur;
                            max_count = cur;
                        }
                    }
                }
            }
            if (max_count > acc_thr) {
                circ[ncirc].x = ccx;
                circ[ncirc].y = ccy;
                circ[ncirc].r = r_best;
                circ[ncirc].votes = max_count;
                ncirc++;
            }
        }
        qsort(circ, ncirc, sizeof(orc_circle), cmp_circle);
        /* RemoveOverlaps */
        float md = (float)min_dist_d;
        if (md < dp) md = dp;
        float md2 = md * md;
        if (ncirc > 1) {
            int end = 1;
            for (int i = 1; i < ncirc; i++) {
                int close = 0;
                for (int j = 0; j < end; j++) {
                    float ddx = circ[j].x - circ[i].x, ddy = circ[j].y - circ[i].y;
                    if (ddx * ddx + ddy * ddy < md2) { close = 1; break; }
                }
                if (!close) circ[end++] = circ[i];
            }
            ncirc = end;
        }
        free(centers);
        free(bins);
    }
    int n_out = ncirc < max_out ? ncirc : max_out;
    for (int i = 0; i < n_out; i++) out[i] = circ[i];
    free(circ);
    free(acc);
    free(nzx);
    free(nzy);
    free(dx);
    free(dy);
    free(edges);
    return ncirc;
}

/* cv2.Canny(gray, t1, t2) (grid_extractor.py:75): aperture 3, L1 gradient; thresholds ordered and floored. */
ORC_API void orc_canny_u8(const u8* gray, int w, int h, double t1, double t2, u8* edges)
{
    short* dx = (short*)malloc(sizeof(short) * (size_t)w * h);
    short* dy = (short*)malloc(sizeof(short) * (size_t)w * h);
    double lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
    orc_sobel3(gray, w, h, dx, dy);
    orc_canny(dx, dy, w, h, (int)floor(lo), (int)floor(hi), edges);
    free(dx);
    free(dy);
}
