// C-ABI of libcbv_hip.so (include/cbv.h): context, host-buffer stage entry
// points, per-square detector state and the device-resident batched pipeline.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <stdlib.h>

#include "cbv_internal.h"

thread_local std::string g_cbv_err;

int cbv_fail(cbv_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_cbv_err = buf;
    if (ctx) ctx->err = buf;
    return code;
}

int dev_ensure(cbv_ctx* ctx, DevBuf* b, size_t bytes)
{
    if (b->cap >= bytes && b->p) return CBV_OK;
    if (b->p) {
        CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
    }
    b->tag = 0;
    // whole 2 MiB units for anything large: the driver can then map the buffer with large GPU pages
    size_t cap = bytes >= (256u << 10) ? (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1) : (bytes + 4095) & ~(size_t)4095;
    CBV_HIP(ctx, hipMalloc(&b->p, cap));
    b->cap = cap;
    return CBV_OK;
}

int ctx_hstage(cbv_ctx* ctx, size_t bytes, u8** p)
{
    if (ctx->h_stage_cap < bytes) {
        if (ctx->h_stage) {
            CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipHostFree(ctx->h_stage);
        }
        ctx->h_stage = nullptr;
        ctx->h_stage_cap = 0;
        const size_t cap = (bytes + 65535) & ~(size_t)65535;
        CBV_HIP(ctx, hipHostMalloc((void**)&ctx->h_stage, cap, hipHostMallocDefault));
        ctx->h_stage_cap = cap;
    }
    *p = ctx->h_stage;
    return CBV_OK;
}

int ctx_worker_stream(cbv_ctx* ctx, hipStream_t* slot, hipStream_t* out)
{
    if (!*slot) CBV_HIP(ctx, hipStreamCreateWithFlags(slot, hipStreamNonBlocking));
    *out = *slot;
    return CBV_OK;
}

void dev_free(DevBuf* b)
{
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr;
    b->cap = 0;
    b->tag = 0;
}

// ---------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------
static const char* kKernelNames[CBV_K_COUNT] = {
    "k_color_lab_hist", "k_clahe_lut", "k_clahe_apply", "k_bilateral", "k_sharpen", "k_norm_lut", "k_normalize",
    "k_warp", "k_squares", "k_gray_blur_hist", "k_otsu", "k_threshold", "k_scan", "k_synth", "k_reset_aux", "k_hough"};

static hipEvent_t prof_get_event(cbv_ctx* ctx)
{
    if (!ctx->prof_pool.empty()) {
        hipEvent_t e = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void prof_begin(cbv_ctx* ctx, int kid)
{
    if (ctx->prof_kid == -2 || (ctx->prof_kid != -1 && ctx->prof_kid != kid)) return;
    ProfSlot s;
    s.kid = kid;
    s.a = prof_get_event(ctx);
    s.b = prof_get_event(ctx);
    (void)hipEventRecord(s.a, ctx->stream);
    ctx->prof_pending.push_back(s);
}

void prof_end(cbv_ctx* ctx, int kid)
{
    if (ctx->prof_kid == -2 || (ctx->prof_kid != -1 && ctx->prof_kid != kid)) return;
    if (ctx->prof_pending.empty()) return;
    (void)hipEventRecord(ctx->prof_pending.back().b, ctx->stream);
}

static void prof_drain(cbv_ctx* ctx)
{
    for (auto& s : ctx->prof_pending) {
        float ms = 0;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            ctx->prof_ms[s.kid] += ms;
            ctx->prof_n[s.kid] += 1;
        }
        ctx->prof_pool.push_back(s.a);
        ctx->prof_pool.push_back(s.b);
    }
    ctx->prof_pending.clear();
}

extern "C" int cbv_profile_enable(cbv_ctx* ctx, int kid)
{
    if (!ctx) return CBV_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    prof_drain(ctx);
    ctx->prof_kid = kid;
    return CBV_OK;
}

extern "C" int cbv_profile_read(cbv_ctx* ctx, int kid, double* total_ms, long long* launches)
{
    if (!ctx || kid < 0 || kid >= CBV_K_COUNT) return CBV_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    prof_drain(ctx);
    if (total_ms) *total_ms = ctx->prof_ms[kid];
    if (launches) *launches = ctx->prof_n[kid];
    return CBV_OK;
}

extern "C" int cbv_profile_reset(cbv_ctx* ctx)
{
    if (!ctx) return CBV_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    prof_drain(ctx);
    for (int i = 0; i < CBV_K_COUNT; i++) {
        ctx->prof_ms[i] = 0;
        ctx->prof_n[i] = 0;
    }
    return CBV_OK;
}

extern "C" const char* cbv_kernel_name(int kid) { return (kid >= 0 && kid < CBV_K_COUNT) ? kKernelNames[kid] : ""; }

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
extern "C" int cbv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* cbv_last_error(const cbv_ctx* ctx) { return ctx ? ctx->err.c_str() : g_cbv_err.c_str(); }
extern "C" const char* cbv_device_name(const cbv_ctx* ctx) { return ctx ? ctx->devname : ""; }

extern "C" int cbv_ctx_create(int device_id, cbv_ctx** out)
{
    if (!out) return cbv_fail(nullptr, CBV_ERR_ARG, "cbv_ctx_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return cbv_fail(nullptr, CBV_ERR_NODEV, "no HIP device available (%s); this library has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_id < 0 || device_id >= n) return cbv_fail(nullptr, CBV_ERR_ARG, "device %d out of range (%d devices)", device_id, n);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return cbv_fail(nullptr, CBV_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return cbv_fail(nullptr, CBV_ERR_NODEV, "device %d is %s; libcbv_hip.so carries gfx950 (MI355X) code only", device_id,
                        prop.gcnArchName);
    cbv_ctx* ctx = new cbv_ctx();
    ctx->device = device_id;
    ctx->num_cus = prop.multiProcessorCount;
    snprintf(ctx->devname, sizeof(ctx->devname), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
#define CK(call)                                                                                       \
    do {                                                                                               \
        hipError_t e2 = (call);                                                                        \
        if (e2 != hipSuccess) {                                                                        \
            int rc = cbv_fail(nullptr, CBV_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e2));    \
            delete ctx;                                                                                \
            return rc;                                                                                 \
        }                                                                                              \
    } while (0)
    CK(hipSetDevice(device_id));
    CK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    CK(hipMalloc((void**)&ctx->tabs, sizeof(StaticTabs)));
    CK(hipMalloc((void**)&ctx->ptabs, sizeof(ProfileTabs)));
    CK(hipMalloc((void**)&ctx->btabs, sizeof(BilateralTabs)));
    {
        StaticTabs* st = new StaticTabs();
        build_static_tabs(st);
        hipError_t up = hipMemcpy(ctx->tabs, st, sizeof(StaticTabs), hipMemcpyHostToDevice);
        delete st;
        CK(up);
        cbv_color_profile none;
        memset(&none, 0, sizeof(none));
        ProfileTabs pt;
        build_profile_tabs(&none, &pt);
        CK(hipMemcpy(ctx->ptabs, &pt, sizeof(pt), hipMemcpyHostToDevice));
        ctx->ptabs_key = none;
        ctx->ptabs_valid = true;
    }
#undef CK
    *out = ctx;
    return CBV_OK;
}

extern "C" void cbv_ctx_destroy(cbv_ctx* ctx)
{
    if (!ctx) return;
    {
        std::lock_guard<std::recursive_mutex> lock(ctx->mu); // a call still running on another thread finishes first
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        prof_drain(ctx);
    }
    for (auto e : ctx->prof_pool) (void)hipEventDestroy(e);
    dev_free(&ctx->in);
    dev_free(&ctx->a);
    dev_free(&ctx->b);
    dev_free(&ctx->c);
    dev_free(&ctx->small);
    if (ctx->tabs) (void)hipFree(ctx->tabs);
    if (ctx->ptabs) (void)hipFree(ctx->ptabs);
    if (ctx->btabs) (void)hipFree(ctx->btabs);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    for (auto& st : ctx->lane_streams)
        if (st) (void)hipStreamDestroy(st);
    if (ctx->scan_stream) (void)hipStreamDestroy(ctx->scan_stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int cbv_ctx_set_stream(cbv_ctx* ctx, void* hip_stream)
{
    if (!ctx) return CBV_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu); // cbv_pipeline_run repoints ctx->stream while it enqueues
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return CBV_OK;
}

extern "C" int cbv_ctx_synchronize(cbv_ctx* ctx)
{
    if (!ctx) return CBV_ERR_ARG;
    hipStream_t st;
    {
        CBV_ENTER(ctx);
        st = ctx->stream; // the caller's stream, never a lane a concurrent cbv_pipeline_run points at for the moment
    }
    CBV_HIP(ctx, hipStreamSynchronize(st));
    return CBV_OK;
}

int ctx_set_profile(cbv_ctx* ctx, const cbv_color_profile* p)
{
    cbv_color_profile key;
    memset(&key, 0, sizeof(key));
    if (p) {
        key.hue_shift = p->hue_shift;
        key.sat_scale = p->sat_scale;
        key.val_scale = p->val_scale;
        key.contrast = p->contrast;
        key.brightness = p->brightness;
        key.radical_mode = p->radical_mode ? 1 : 0;
        key.target_hue = p->target_hue;
        key.hue_window = p->hue_window;
        key.enabled = p->enabled ? 1 : 0;
    }
    if (ctx->ptabs_valid && memcmp(&key, &ctx->ptabs_key, sizeof(key)) == 0) return CBV_OK;
    ProfileTabs pt;
    build_profile_tabs(&key, &pt);
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CBV_HIP(ctx, hipMemcpy(ctx->ptabs, &pt, sizeof(pt), hipMemcpyHostToDevice));
    ctx->ptabs_key = key;
    ctx->ptabs_valid = true;
    return CBV_OK;
}

int ctx_set_bilateral(cbv_ctx* ctx, int d, double sc, double ss)
{
    if (ctx->b_d == d && ctx->b_sc == sc && ctx->b_ss == ss) return CBV_OK;
    if (build_bilateral_tabs(d, sc, ss, &ctx->btabs_host) != 0)
        return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "bilateral d=%d not supported (radius <= 4)", d);
    if (ctx->btabs_host.radius > 4) return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "bilateral d=%d not supported (radius <= 4)", d);
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CBV_HIP(ctx, hipMemcpy(ctx->btabs, &ctx->btabs_host, sizeof(BilateralTabs), hipMemcpyHostToDevice));
    ctx->b_d = d;
    ctx->b_sc = sc;
    ctx->b_ss = ss;
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// host-buffer stage entry points
// ---------------------------------------------------------------------------
static int check_img(cbv_ctx* ctx, const void* p, int w, int h, int stride, int cn, const char* what)
{
    if (!ctx) return cbv_fail(nullptr, CBV_ERR_ARG, "%s: ctx is null", what);
    if (!p || w <= 0 || h <= 0 || stride < w * cn) return cbv_fail(ctx, CBV_ERR_ARG, "%s: bad image arguments (w=%d h=%d stride=%d)", what, w, h, stride);
    if ((size_t)w * h > (size_t)1 << 28) return cbv_fail(ctx, CBV_ERR_ARG, "%s: image too large", what);
    return CBV_OK;
}

static Geom tight_geom(int w, int h)
{
    Geom g;
    g.w = w;
    g.h = h;
    g.stride = w * 3;
    g.frame_stride = ((size_t)w * 3 * h + 255) & ~(size_t)255;
    return g;
}

#define RC(x)             \
    do {                  \
        int rc__ = (x);   \
        if (rc__) return rc__; \
    } while (0)

// rows between a host image (any row stride) and a tight device image, asynchronous.  Tight host rows travel as ONE
// linear copy: the 2-D entry point takes a slower path even when width == pitch (a 1080p frame to pageable memory:
// 365 us against 119 us, tools/ubench_copy.hip).
static int rows_h2d(cbv_ctx* ctx, void* dst, const u8* src, int stride, int wbytes, int h)
{
    if (stride == wbytes) CBV_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)wbytes * h, hipMemcpyHostToDevice, ctx->stream));
    else CBV_HIP(ctx, hipMemcpy2DAsync(dst, wbytes, src, stride, wbytes, h, hipMemcpyHostToDevice, ctx->stream));
    return CBV_OK;
}

static int rows_d2h(cbv_ctx* ctx, u8* dst, int stride, const void* src, int wbytes, int h)
{
    if (stride == wbytes) CBV_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)wbytes * h, hipMemcpyDeviceToHost, ctx->stream));
    else CBV_HIP(ctx, hipMemcpy2DAsync(dst, stride, src, wbytes, wbytes, h, hipMemcpyDeviceToHost, ctx->stream));
    return CBV_OK;
}

static int upload(cbv_ctx* ctx, DevBuf* dst, const u8* src, int wbytes, int h, int stride)
{
    int rc = dev_ensure(ctx, dst, (size_t)wbytes * h + 256);
    if (rc) return rc;
    return rows_h2d(ctx, dst->p, src, stride, wbytes, h);
}

static int download(cbv_ctx* ctx, const void* src, u8* dst, int wbytes, int h, int stride)
{
    int rc = rows_d2h(ctx, dst, stride, src, wbytes, h);
    if (rc) return rc;
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

struct SmallLayout {
    u32* aux;
    u8* luts;
    u32* packed; // CLAHE corner words (k_clahe_lut -> k_clahe_apply), null when not reserved
    u8* norm_lut;
    // The buffer's tag says which (tiles, batch) layout of `aux` is known to be as k_reset_aux leaves it: enhance_dev's own
    // kernels restore that state as they go (launch_clahe_lut, self_clean), so a pass over the same layout needs no reset
    // launch.  Everything else that writes `aux` leaves the tag 0 (aux_dirty).
    DevBuf* owner;
};
static unsigned long long aux_clean_tag(int tiles, int batch) { return (1ull << 63) | ((unsigned long long)tiles << 32) | (unsigned)batch; }
static int aux_reset(cbv_ctx* ctx, const SmallLayout& S, int tiles, int batch) // reset for a stand-alone stage
{
    S.owner->tag = 0;
    return launch_reset_aux(ctx, S.aux, tiles, batch);
}

// tiles_x / tiles_y > 0 also reserve the packed CLAHE corner words ([batch][tiles_y + 1][tiles_x + 1][256] u32)
static int small_layout(cbv_ctx* ctx, DevBuf* buf, int tiles, int batch, SmallLayout* L, int tiles_x = 0, int tiles_y = 0)
{
    size_t aux_b = aux_words(tiles) * 4 * batch;
    aux_b = (aux_b + 255) & ~(size_t)255;
    size_t lut_b = ((size_t)tiles * 256 * batch + 255) & ~(size_t)255;
    size_t nl_b = ((size_t)256 * batch + 255) & ~(size_t)255;
    size_t pk_b = (size_t)(tiles_x + 1) * (tiles_y + 1) * 1024 * batch;
    if (tiles_x <= 0 || tiles_y <= 0) pk_b = 0;
    int rc = dev_ensure(ctx, buf, aux_b + lut_b + nl_b + pk_b);
    if (rc) return rc;
    L->owner = buf;
    L->aux = (u32*)buf->p;
    L->luts = (u8*)buf->p + aux_b;
    L->norm_lut = (u8*)buf->p + aux_b + lut_b;
    L->packed = pk_b ? (u32*)((u8*)buf->p + aux_b + lut_b + nl_b) : nullptr;
    return CBV_OK;
}


extern "C" int cbv_apply_color_profile(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride,
                                       const cbv_color_profile* profile, uint8_t* out, int out_stride)
{
    RC(check_img(ctx, bgr, w, h, stride, 3, "cbv_apply_color_profile"));
    if (!out || !profile) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_apply_color_profile: null argument");
    CBV_ENTER(ctx);
    RC(upload(ctx, &ctx->in, bgr, w * 3, h, stride));
    if (!profile->enabled) return download(ctx, ctx->in.p, out, w * 3, h, out_stride); // `{}` profile: identity
    RC(ctx_set_profile(ctx, profile));
    Geom g = tight_geom(w, h);
    RC(dev_ensure(ctx, &ctx->a, g.frame_stride));
    SmallLayout S;
    RC(small_layout(ctx, &ctx->small, 1, 1, &S));
    S.owner->tag = 0;
    ClaheGeom cg = clahe_geom(w, h, 0.0, 1, 1); // one tile = whole image; only used for the traversal
    RC(launch_color_lab_hist(ctx, (const u8*)ctx->in.p, (u8*)ctx->a.p, S.aux, g, cg, 1, 1, 0));
    return download(ctx, ctx->a.p, out, w * 3, h, out_stride);
}

extern "C" int cbv_correct_lighting(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, double clip_limit,
                                    int tiles_x, int tiles_y, uint8_t* out, int out_stride)
{
    RC(check_img(ctx, bgr, w, h, stride, 3, "cbv_correct_lighting"));
    if (!out || tiles_x <= 0 || tiles_y <= 0 || tiles_x > 64 || tiles_y > 64) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_correct_lighting: bad arguments");
    CBV_ENTER(ctx);
    RC(upload(ctx, &ctx->in, bgr, w * 3, h, stride));
    Geom g = tight_geom(w, h);
    RC(dev_ensure(ctx, &ctx->a, g.frame_stride));
    RC(dev_ensure(ctx, &ctx->b, g.frame_stride));
    ClaheGeom cg = clahe_geom(w, h, clip_limit, tiles_x, tiles_y);
    int tiles = tiles_x * tiles_y;
    SmallLayout S;
    RC(small_layout(ctx, &ctx->small, tiles, 1, &S, tiles_x, tiles_y));
    RC(aux_reset(ctx, S, tiles, 1));
    RC(launch_color_lab_hist(ctx, (const u8*)ctx->in.p, (u8*)ctx->a.p, S.aux, g, cg, 1, 0, 1));
    RC(launch_clahe_lut(ctx, S.aux, S.luts, cg, 1, S.packed));
    RC(launch_clahe_apply(ctx, (const u8*)ctx->a.p, S.packed, (u8*)ctx->b.p, g, cg, 1));
    return download(ctx, ctx->b.p, out, w * 3, h, out_stride);
}

extern "C" int cbv_clahe_apply(cbv_ctx* ctx, const uint8_t* gray, int w, int h, int stride, double clip_limit, int tiles_x,
                               int tiles_y, uint8_t* out, int out_stride)
{
    RC(check_img(ctx, gray, w, h, stride, 1, "cbv_clahe_apply"));
    if (!out || tiles_x <= 0 || tiles_y <= 0 || tiles_x > 64 || tiles_y > 64) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_clahe_apply: bad arguments");
    CBV_ENTER(ctx);
    RC(upload(ctx, &ctx->in, gray, w, h, stride));
    RC(dev_ensure(ctx, &ctx->a, ((size_t)w * h + 255) & ~(size_t)255));
    ClaheGeom cg = clahe_geom(w, h, clip_limit, tiles_x, tiles_y);
    SmallLayout S;
    RC(small_layout(ctx, &ctx->small, tiles_x * tiles_y, 1, &S));
    S.owner->tag = 0;
    RC(launch_clahe_gray(ctx, (const u8*)ctx->in.p, w, h, w, cg, S.aux, S.luts, (u8*)ctx->a.p));
    return download(ctx, ctx->a.p, out, w, h, out_stride);
}

extern "C" int cbv_reduce_noise(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, int d, double sigma_color,
                                double sigma_space, uint8_t* out, int out_stride)
{
    RC(check_img(ctx, bgr, w, h, stride, 3, "cbv_reduce_noise"));
    if (!out) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_reduce_noise: out is null");
    CBV_ENTER(ctx);
    RC(ctx_set_bilateral(ctx, d, sigma_color, sigma_space));
    RC(upload(ctx, &ctx->in, bgr, w * 3, h, stride));
    Geom g = tight_geom(w, h);
    RC(dev_ensure(ctx, &ctx->a, g.frame_stride));
    RC(launch_bilateral(ctx, (const u8*)ctx->in.p, (u8*)ctx->a.p, g, 1));
    return download(ctx, ctx->a.p, out, w * 3, h, out_stride);
}

extern "C" int cbv_sharpen(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, const float* kernel9, uint8_t* out,
                           int out_stride)
{
    RC(check_img(ctx, bgr, w, h, stride, 3, "cbv_sharpen"));
    if (!out || !kernel9) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_sharpen: null argument");
    CBV_ENTER(ctx);
    RC(upload(ctx, &ctx->in, bgr, w * 3, h, stride));
    Geom g = tight_geom(w, h);
    RC(dev_ensure(ctx, &ctx->a, g.frame_stride));
    SmallLayout S;
    RC(small_layout(ctx, &ctx->small, 1, 1, &S));
    RC(aux_reset(ctx, S, 1, 1));
    RC(launch_sharpen(ctx, (const u8*)ctx->in.p, (u8*)ctx->a.p, S.aux, 1, g, kernel9, 1));
    return download(ctx, ctx->a.p, out, w * 3, h, out_stride);
}

int launch_minmax(cbv_ctx* ctx, const u8* src, u32* aux, int tiles, Geom g, int batch);

extern "C" int cbv_normalize_intensity(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, uint8_t* out,
                                       int out_stride)
{
    RC(check_img(ctx, bgr, w, h, stride, 3, "cbv_normalize_intensity"));
    if (!out) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_normalize_intensity: out is null");
    CBV_ENTER(ctx);
    RC(upload(ctx, &ctx->in, bgr, w * 3, h, stride));
    Geom g = tight_geom(w, h);
    RC(dev_ensure(ctx, &ctx->a, g.frame_stride));
    SmallLayout S;
    RC(small_layout(ctx, &ctx->small, 1, 1, &S));
    RC(aux_reset(ctx, S, 1, 1));
    RC(launch_minmax(ctx, (const u8*)ctx->in.p, S.aux, 1, g, 1));
    RC(launch_normalize(ctx, (const u8*)ctx->in.p, (u8*)ctx->a.p, norm_from_minmax(S.aux, 1), g, 1));
    return download(ctx, ctx->a.p, out, w * 3, h, out_stride);
}

extern "C" int cbv_prepare_analysis(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, uint8_t* gray,
                                    int gray_stride, uint8_t* binary, int binary_stride, int* otsu_threshold)
{
    RC(check_img(ctx, bgr, w, h, stride, 3, "cbv_prepare_analysis"));
    if (!gray || !binary) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_prepare_analysis: null output");
    CBV_ENTER(ctx);
    RC(upload(ctx, &ctx->in, bgr, w * 3, h, stride));
    Geom g = tight_geom(w, h);
    size_t plane = ((size_t)w * h + 255) & ~(size_t)255;
    RC(dev_ensure(ctx, &ctx->a, plane * 3));
    u8* dgray = (u8*)ctx->a.p;
    u8* dblur = dgray + plane;
    u8* dbin = dblur + plane;
    SmallLayout S;
    RC(small_layout(ctx, &ctx->small, 1, 1, &S));
    RC(aux_reset(ctx, S, 1, 1));
    RC(launch_gray_blur_hist(ctx, (const u8*)ctx->in.p, dgray, dblur, S.aux, 1, g, 1));
    RC(launch_otsu(ctx, S.aux, 1, w * h, 1));
    RC(launch_threshold(ctx, dblur, dbin, S.aux, 1, w, h, 1));
    RC(rows_d2h(ctx, gray, gray_stride, dgray, w, h));
    RC(rows_d2h(ctx, binary, binary_stride, dbin, w, h));
    u32 t = 0;
    CBV_HIP(ctx, hipMemcpyAsync(&t, S.aux + (size_t)1 * 256 + 2 + 256, 4, hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (otsu_threshold) *otsu_threshold = (int)t;
    return CBV_OK;
}

extern "C" int cbv_canny(cbv_ctx* ctx, const uint8_t* img, int w, int h, int stride, int cn, double threshold1, double threshold2,
                         uint8_t* edges, int edges_stride)
{
    RC(check_img(ctx, img, w, h, stride, cn, "cbv_canny"));
    if (!edges || (cn != 1 && cn != 3)) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_canny: null output or unsupported channel count");
    CBV_ENTER(ctx);
    RC(upload(ctx, &ctx->in, img, w * cn, h, stride));
    // cv::Canny: thresholds are ordered, then floored (L1 gradient)
    double lo = threshold1 < threshold2 ? threshold1 : threshold2, hi = threshold1 < threshold2 ? threshold2 : threshold1;
    const int low = (int)floor(lo), high = (int)floor(hi);
    size_t plane = ((size_t)w * h + 255) & ~(size_t)255;
    RC(dev_ensure(ctx, &ctx->a, plane));
    RC(launch_canny(ctx, (const u8*)ctx->in.p, w, h, w * cn, cn, low, high, (u8*)ctx->a.p, &ctx->b));
    RC(rows_d2h(ctx, edges, edges_stride, ctx->a.p, w, h));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

extern "C" int cbv_board_corners_from_edges(const uint8_t* edges, int w, int h, int stride, int32_t* pts8, int* n_contours)
{
    if (!edges || !pts8 || w <= 0 || h <= 0 || stride < w) return CBV_ERR_ARG;
    std::vector<u8> tight((size_t)w * h);
    for (int y = 0; y < h; y++) memcpy(tight.data() + (size_t)y * w, edges + (size_t)y * stride, (size_t)w);
    return board_corners_from_edges(tight.data(), w, h, pts8, n_contours);
}

extern "C" int cbv_largest_contour_polygon(const uint8_t* edges, int w, int h, int stride, double eps_frac, int32_t* pts, int cap,
                                           double* area, int* contour_len)
{
    if (!edges || !pts || w <= 0 || h <= 0 || stride < w || cap <= 0) return CBV_ERR_ARG;
    std::vector<u8> tight((size_t)w * h);
    for (int y = 0; y < h; y++) memcpy(tight.data() + (size_t)y * w, edges + (size_t)y * stride, (size_t)w);
    return largest_contour_polygon(tight.data(), w, h, eps_frac, pts, cap, area, contour_len);
}

extern "C" int cbv_find_chessboard_corners(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, int32_t* pts8,
                                           uint8_t* dilated_out, int dilated_stride)
{
    RC(check_img(ctx, bgr, w, h, stride, 3, "cbv_find_chessboard_corners"));
    if (!pts8) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_find_chessboard_corners: null output");
    CBV_ENTER(ctx);
    RC(upload(ctx, &ctx->in, bgr, w * 3, h, stride));
    const size_t plane = ((size_t)w * h + 255) & ~(size_t)255;
    RC(dev_ensure(ctx, &ctx->a, plane * 3 + 256));
    u8* blur = (u8*)ctx->a.p;
    u8* edges = blur + plane;
    u8* dil = edges + plane;
    int coef[16] = {0};
    build_gaussian_q8_sigma(7, 1.0, coef);                      // cv2.GaussianBlur(gray, (7, 7), 1)
    int* coef_dev = (int*)(dil + plane);
    CBV_HIP(ctx, hipMemcpyAsync(coef_dev, coef, sizeof(int) * 16, hipMemcpyHostToDevice, ctx->stream));
    RC(launch_gray_gauss(ctx, (const u8*)ctx->in.p, w, h, w * 3, 3, coef_dev, 7, blur));
    RC(launch_canny(ctx, blur, w, h, w, 1, 30, 100, edges, &ctx->b)); // cv2.Canny(blur, 30, 100)
    RC(launch_dilate_rect(ctx, edges, w, h, 6, dil));           // three 5x5 dilations = one 13x13
    std::vector<u8> host((size_t)w * h);
    CBV_HIP(ctx, hipMemcpyAsync(host.data(), dil, (size_t)w * h, hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (dilated_out)
        for (int y = 0; y < h; y++) memcpy(dilated_out + (size_t)y * dilated_stride, host.data() + (size_t)y * w, (size_t)w);
    int n_contours = 0;
    return board_corners_from_edges(host.data(), w, h, pts8, &n_contours);
}

// enhancement chain on device buffers: src -> (A, B ping-pong) ; result pointer returned.
// When `fold_norm` the final normalize pass is skipped and the caller applies S.norm_lut downstream.
static PxRect px_dilate(PxRect r, int d, Geom g)
{
    PxRect o = {r.x0 - d, r.y0 - d, r.x1 + d, r.y1 + d};
    o.x0 = o.x0 < 0 ? 0 : o.x0;
    o.y0 = o.y0 < 0 ? 0 : o.y0;
    o.x1 = o.x1 > g.w ? g.w : o.x1;
    o.y1 = o.y1 > g.h ? g.h : o.y1;
    return o;
}

// `region` (with a third buffer C, and only when the caller folds normalize into its own gather): the consumer samples
// just these pixels of the enhanced frame; see "Region-limited enhancement" in cbv_internal.h.
// `norm`: where the consumer of a fold_norm result finds cv2.normalize's byte map (NormSrc).
static int enhance_dev(cbv_ctx* ctx, const u8* src, u8* A, u8* B, Geom g, const cbv_enhance_params* P, SmallLayout S,
                       int batch, bool fold_norm, u8** result, NormSrc* norm, const PxRect* region = nullptr, u8* C = nullptr)
{
    ClaheGeom cg = clahe_geom(g.w, g.h, P->clahe_clip_limit, P->tiles_x, P->tiles_y);
    int tiles = P->tiles_x * P->tiles_y;
    if (!S.packed) return cbv_fail(ctx, CBV_ERR_STATE, "enhance_dev: the small-buffer layout lacks the packed CLAHE words");
    // histograms zero, [min, max] = [255, 0]: by a launch the first time this layout of the buffer is used, by the previous
    // pass's k_clahe_lut afterwards (SmallLayout::owner).  The tag stays 0 until every launch of this pass is enqueued.
    const unsigned long long clean = aux_clean_tag(tiles, batch);
    if (S.owner->tag != clean) RC(launch_reset_aux(ctx, S.aux, tiles, batch));
    S.owner->tag = 0;
    struct MarkClean {
        DevBuf* b;
        unsigned long long tag;
        bool ok = false;
        ~MarkClean() { if (ok) b->tag = tag; }
    } mark{S.owner, clean};
    RC(launch_color_lab_hist(ctx, src, A, S.aux, g, cg, batch, P->profile.enabled ? 1 : 0, 1));
    RC(launch_clahe_lut(ctx, S.aux, S.luts, cg, batch, S.packed, 1));
    // a run of a frame or two is launch latency: its consumer works the byte map out of [min, max] itself (NormSrc)
    const bool lut_launch = batch > 2;
    if (region && C && fold_norm && sharpen_region_ok(P->sharpen_kernel) && region->x1 > region->x0 && region->y1 > region->y0) {
        // what each stage must find complete in its input: the stage after it, rounded out to that stage's tiles, plus
        // that stage's halo (sharpen 1 px, bilateral 4 px as staged)
        const PxRect s_need = px_dilate(*region, 0, g);
        const PxRect b_need = px_dilate(sharpen_region_cover(g, s_need), 1, g);
        const PxRect c_need = px_dilate(bilateral_region_cover(ctx, g, batch, b_need), 4, g);
        EnhanceRegion ec = {c_need, 0, {nullptr, 0}}, eb = {b_need, 0, {nullptr, 0}}, es = {s_need, 0, {nullptr, 0}};
        // the sharpened frame goes to a THIRD buffer: the complement pass of the bilateral still reads CLAHE's output
        // (B) in a ring inside the region, and that of sharpen the bilateral's (A)
        RC(launch_clahe_apply(ctx, A, S.packed, B, g, cg, batch, &ec));
        RC(launch_bilateral(ctx, B, A, g, batch, &eb));
        RC(launch_sharpen(ctx, A, C, S.aux, tiles, g, P->sharpen_kernel, batch, &es));
        const SatGate gate = {S.aux + (size_t)tiles * 256, aux_words(tiles)}; // min, max of frame 0 (k_sharpen_box)
        ec.invert = eb.invert = es.invert = 1;
        ec.gate = eb.gate = es.gate = gate;
        RC(launch_clahe_apply(ctx, A, S.packed, B, g, cg, batch, &ec));
        RC(launch_bilateral(ctx, B, A, g, batch, &eb));
        RC(launch_sharpen(ctx, A, C, S.aux, tiles, g, P->sharpen_kernel, batch, &es));
        if (lut_launch) RC(launch_norm_lut(ctx, S.aux, tiles, S.norm_lut, batch));
        *norm = lut_launch ? norm_from_lut(S.norm_lut) : norm_from_minmax(S.aux, tiles);
        *result = C;
        mark.ok = true;
        return CBV_OK;
    }
    RC(launch_clahe_apply(ctx, A, S.packed, B, g, cg, batch));
    RC(launch_bilateral(ctx, B, A, g, batch));
    RC(launch_sharpen(ctx, A, B, S.aux, tiles, g, P->sharpen_kernel, batch));
    if (lut_launch) RC(launch_norm_lut(ctx, S.aux, tiles, S.norm_lut, batch));
    *norm = lut_launch ? norm_from_lut(S.norm_lut) : norm_from_minmax(S.aux, tiles);
    if (fold_norm) {
        *result = B;
        mark.ok = true;
        return CBV_OK;
    }
    RC(launch_normalize(ctx, B, A, *norm, g, batch));
    *result = A;
    mark.ok = true;
    return CBV_OK;
}

// Source pixels a dw x dh warp can sample: the image of the destination rectangle under Minv (a projective map keeps it
// convex while W > 0 on it, so its four corners bound it), + 3 px for the 1/32-px rounding and the bilinear taps, clipped
// to the frame.  False when the map is degenerate on the rectangle or the footprint is empty: callers then take the
// whole frame.
static bool warp_footprint(const double* Minv, int dw, int dh, int w, int h, PxRect* out)
{
    double lo[2] = {1e30, 1e30}, hi[2] = {-1e30, -1e30};
    for (int k = 0; k < 4; k++) {
        const double dx = (k & 1) ? dw - 1 : 0, dy = (k & 2) ? dh - 1 : 0;
        const double W = Minv[6] * dx + Minv[7] * dy + Minv[8];
        if (!(W > 1e-12)) return false;
        const double c[2] = {(Minv[0] * dx + Minv[1] * dy + Minv[2]) / W, (Minv[3] * dx + Minv[4] * dy + Minv[5]) / W};
        for (int a = 0; a < 2; a++) {
            lo[a] = std::min(lo[a], c[a]);
            hi[a] = std::max(hi[a], c[a]);
        }
    }
    if (!(lo[0] > -1e8 && lo[1] > -1e8 && hi[0] < 1e8 && hi[1] < 1e8)) return false; // (the casts below stay inside int)
    PxRect r = {(int)floor(lo[0]) - 3, (int)floor(lo[1]) - 3, (int)ceil(hi[0]) + 4, (int)ceil(hi[1]) + 4};
    r.x0 = std::max(r.x0, 0);
    r.y0 = std::max(r.y0, 0);
    r.x1 = std::min(r.x1, w);
    r.y1 = std::min(r.y1, h);
    if (r.x1 <= r.x0 || r.y1 <= r.y0) return false;
    *out = r;
    return true;
}

static int check_params(cbv_ctx* ctx, const cbv_enhance_params* P)
{
    if (!P) return cbv_fail(ctx, CBV_ERR_ARG, "enhance params are null");
    if (P->tiles_x <= 0 || P->tiles_y <= 0 || P->tiles_x > 64 || P->tiles_y > 64) return cbv_fail(ctx, CBV_ERR_ARG, "bad CLAHE tile grid");
    RC(ctx_set_profile(ctx, &P->profile));
    RC(ctx_set_bilateral(ctx, P->bilateral_d, P->sigma_color, P->sigma_space));
    return CBV_OK;
}

extern "C" int cbv_process_pipeline(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride,
                                    const cbv_enhance_params* params, uint8_t* out, int out_stride)
{
    RC(check_img(ctx, bgr, w, h, stride, 3, "cbv_process_pipeline"));
    if (!out) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_process_pipeline: out is null");
    CBV_ENTER(ctx);
    RC(check_params(ctx, params));
    RC(upload(ctx, &ctx->in, bgr, w * 3, h, stride));
    Geom g = tight_geom(w, h);
    RC(dev_ensure(ctx, &ctx->a, g.frame_stride));
    RC(dev_ensure(ctx, &ctx->b, g.frame_stride));
    SmallLayout S;
    RC(small_layout(ctx, &ctx->small, params->tiles_x * params->tiles_y, 1, &S, params->tiles_x, params->tiles_y));
    u8* res = nullptr;
    NormSrc norm;
    RC(enhance_dev(ctx, (const u8*)ctx->in.p, (u8*)ctx->a.p, (u8*)ctx->b.p, g, params, S, 1, false, &res, &norm));
    return download(ctx, res, out, w * 3, h, out_stride);
}

extern "C" int cbv_warp_perspective(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, const double* M9, int dw,
                                    int dh, int rot180, uint8_t* out, int out_stride)
{
    RC(check_img(ctx, bgr, w, h, stride, 3, "cbv_warp_perspective"));
    if (!out || !M9 || dw <= 0 || dh <= 0 || dw > 8192 || dh > 8192) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_warp_perspective: bad arguments");
    CBV_ENTER(ctx);
    double Minv[9];
    if (!host_invert3x3(M9, Minv)) memset(Minv, 0, sizeof(Minv)); // cv::invert returns a zero matrix when singular
    // Only the rows of the frame the warp can sample cross PCIe (the board quad's rows; whole rows, because a 2-D copy
    // with a wide pitch runs at a fifth of the rate of a contiguous one on this platform: tools/ubench_copy.hip).  The
    // rest of the device buffer is never read.
    PxRect fp;
    int y0 = 0, y1 = h;
    if (warp_footprint(Minv, dw, dh, w, h, &fp)) {
        y0 = fp.y0;
        y1 = fp.y1;
    }
    RC(dev_ensure(ctx, &ctx->in, (size_t)w * 3 * h + 256));
    if (ctx->debug_poison) CBV_HIP(ctx, hipMemsetAsync(ctx->in.p, 0xA5, (size_t)w * 3 * h, ctx->stream));
    static const bool tm = getenv("CBV_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = tm ? now() : 0;
    RC(rows_h2d(ctx, (u8*)ctx->in.p + (size_t)y0 * w * 3, bgr + (size_t)y0 * stride, stride, w * 3, y1 - y0));
    const double t1 = tm ? now() : 0;
    Geom g = tight_geom(w, h);
    RC(dev_ensure(ctx, &ctx->a, (size_t)dw * dh * 3 + 256));
    RC(launch_warp(ctx, (const u8*)ctx->in.p, g, Minv, dw, dh, rot180, (u8*)ctx->a.p, dw * 3, (size_t)dw * dh * 3, NormSrc(), 1));
    const double t2 = tm ? now() : 0;
    RC(rows_d2h(ctx, out, out_stride, ctx->a.p, dw * 3, dh));
    const double t3 = tm ? now() : 0;
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (tm) fprintf(stderr, "warp: h2d %.1f launch %.1f d2h %.1f sync %.1f us (rows %d..%d)\n", t1 - t0, t2 - t1, t3 - t2, now() - t3, y0, y1);
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// per-square detector state
// ---------------------------------------------------------------------------
struct cbv_squares {
    cbv_ctx* ctx = nullptr;
    int n = 0;
    int blur_k = 0;
    std::vector<SquareDesc> descs;
    size_t plane_total = 0, mask_total = 0;
    DevBuf d_descs, d_masks, d_gray, d_ref, d_mean, d_var, d_stats, d_select, d_coef, d_stage, d_hough, d_retry;
    DevBuf d_fast;  // one-call entry points: worklist (1 + 64 u32) | retry list (1 + 64 u32) | dflags (64 B) | second statistics set
    DevBuf d_gray5; // cbv_squares_detect_changes with blur_k != 5: the squares as PieceDetector preprocesses them (k = 5)
    std::vector<SquareDesc> descs_on_dev; // what d_descs holds (a host-image load re-uploads only when it differs)
    std::vector<u8> stage;
    bool has_ref = false, has_model = false;
    int coef_k = -1;
};

extern "C" int cbv_squares_create(cbv_ctx* ctx, cbv_squares** out)
{
    if (!ctx || !out) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_squares_create: null argument");
    cbv_squares* s = new cbv_squares();
    s->ctx = ctx;
    *out = s;
    return CBV_OK;
}

extern "C" void cbv_squares_destroy(cbv_squares* s)
{
    if (!s) return;
    {
        std::lock_guard<std::recursive_mutex> lock(s->ctx->mu);
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
    }
    DevBuf* bufs[] = {&s->d_descs, &s->d_masks, &s->d_gray, &s->d_ref, &s->d_mean, &s->d_var, &s->d_stats, &s->d_select, &s->d_coef, &s->d_stage, &s->d_hough, &s->d_retry,
                      &s->d_fast, &s->d_gray5};
    for (auto b : bufs) dev_free(b);
    delete s;
}

// np.sqrt(var) of every pixel, kept beside the variance plane by the kernels that write it
static float* sq_sd(cbv_squares* s) { return (float*)s->d_var.p + s->plane_total; }

static int squares_set_coef(cbv_squares* s, int blur_k)
{
    cbv_ctx* ctx = s->ctx;
    if (blur_k < 1) blur_k = 1;
    blur_k |= 1;
    if (blur_k > 31) return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "blur kernel %d too large (max 31)", blur_k);
    if (s->coef_k != blur_k) {
        int coef[32] = {0};
        build_gaussian_q8(blur_k, coef);
        RC(dev_ensure(ctx, &s->d_coef, sizeof(coef)));
        CBV_HIP(ctx, hipMemcpyAsync(s->d_coef.p, coef, sizeof(coef), hipMemcpyHostToDevice, ctx->stream));
        CBV_HIP(ctx, hipStreamSynchronize(ctx->stream)); // coef is a stack array
        s->coef_k = blur_k;
    }
    s->blur_k = blur_k;
    return CBV_OK;
}

// (re)build geometry-dependent tables when the set of square shapes changes
static int squares_set_geometry(cbv_squares* s, const int* ws, const int* hs, int n)
{
    cbv_ctx* ctx = s->ctx;
    bool same = (n == s->n);
    for (int i = 0; same && i < n; i++) same = s->descs[i].w == ws[i] && s->descs[i].h == hs[i];
    if (same) return CBV_OK;
    for (int i = 0; i < n; i++) // validate before any state changes: a rejected load leaves the set as it was
        if (ws[i] <= 0 || hs[i] <= 0 || ws[i] > CBV_MAX_SQUARE_DIM || hs[i] > CBV_MAX_SQUARE_DIM)
            return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "square %d is %dx%d; supported up to %dx%d", i, ws[i], hs[i], CBV_MAX_SQUARE_DIM, CBV_MAX_SQUARE_DIM);
    s->descs.assign(n, SquareDesc());
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        s->descs[i].w = ws[i];
        s->descs[i].h = hs[i];
        s->descs[i].plane_off = (int)off;
        s->descs[i].mask_off = (int)off;
        off += ((size_t)ws[i] * hs[i] + 15) & ~(size_t)15;
    }
    s->n = n;
    s->plane_total = off;
    s->has_ref = s->has_model = false;
    std::vector<u8> masks(off, 0);
    for (int i = 0; i < n; i++) {
        build_piece_mask(ws[i], hs[i], masks.data() + s->descs[i].mask_off);
        square_region_counts(masks.data() + s->descs[i].mask_off, ws[i] * hs[i], s->descs[i].cnt);
    }
    RC(dev_ensure(ctx, &s->d_masks, off));
    RC(dev_ensure(ctx, &s->d_gray, off));
    RC(dev_ensure(ctx, &s->d_ref, off));
    RC(dev_ensure(ctx, &s->d_mean, off * 4));
    RC(dev_ensure(ctx, &s->d_var, off * 8)); // variance plane, then its square root (sq_sd)
    RC(dev_ensure(ctx, &s->d_stats, sizeof(cbv_sq_stats) * n));
    RC(dev_ensure(ctx, &s->d_select, CBV_MAX_SQUARES * 4));
    RC(dev_ensure(ctx, &s->d_descs, sizeof(SquareDesc) * n));
    CBV_HIP(ctx, hipMemcpy(s->d_masks.p, masks.data(), off, hipMemcpyHostToDevice));
    return CBV_OK;
}

extern "C" int cbv_squares_load(cbv_squares* s, const cbv_square_view* views, int n, int blur_k)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    if (!views || n <= 0 || n > CBV_MAX_SQUARES) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_squares_load: bad arguments (n=%d)", n);
    CBV_ENTER(ctx);
    int ws[CBV_MAX_SQUARES], hs[CBV_MAX_SQUARES];
    for (int i = 0; i < n; i++) {
        if (!views[i].data) { // keep the current gray of this square (its geometry must already be known)
            if (i >= s->n) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_squares_load: view %d is null but the square is unknown", i);
            ws[i] = s->descs[i].w;
            hs[i] = s->descs[i].h;
            continue;
        }
        if ((views[i].cn != 1 && views[i].cn != 3) || views[i].stride < views[i].w * views[i].cn)
            return cbv_fail(ctx, CBV_ERR_ARG, "cbv_squares_load: bad view %d", i);
        ws[i] = views[i].w;
        hs[i] = views[i].h;
    }
    RC(squares_set_geometry(s, ws, hs, n));
    RC(squares_set_coef(s, blur_k));
    // pack the views tightly into one staging buffer (the caller's arrays may be scattered)
    size_t total = 0;
    for (int i = 0; i < n; i++) {
        if (!views[i].data) {
            s->descs[i].cn = 0; // skipped by the kernel
            continue;
        }
        s->descs[i].cn = views[i].cn;
        s->descs[i].stride = views[i].w * views[i].cn;
        s->descs[i].src_off = (int)total;
        total += ((size_t)views[i].w * views[i].cn * views[i].h + 15) & ~(size_t)15;
    }
    if (total == 0) total = 16;
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream)); // staging may still be in flight from the previous call
    s->stage.resize(total);
    for (int i = 0; i < n; i++) {
        if (!views[i].data) continue;
        const int rb = views[i].w * views[i].cn;
        for (int y = 0; y < views[i].h; y++)
            memcpy(s->stage.data() + s->descs[i].src_off + (size_t)y * rb, views[i].data + (size_t)y * views[i].stride, rb);
    }
    RC(dev_ensure(ctx, &s->d_stage, total));
    CBV_HIP(ctx, hipMemcpyAsync(s->d_stage.p, s->stage.data(), total, hipMemcpyHostToDevice, ctx->stream));
    CBV_HIP(ctx, hipMemcpyAsync(s->d_descs.p, s->descs.data(), sizeof(SquareDesc) * n, hipMemcpyHostToDevice, ctx->stream));
    s->descs_on_dev.clear();
    RC(launch_squares_preprocess(ctx, (const u8*)s->d_stage.p, 0, (const SquareDesc*)s->d_descs.p, n, (const int*)s->d_coef.p,
                                 s->blur_k, (u8*)s->d_gray.p, 0, 1));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

extern "C" int cbv_squares_load_dev(cbv_squares* s, const void* dev_img, int w, int h, int stride, int cn,
                                    const cbv_roi* rois, int n, int blur_k)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    if (!dev_img || !rois || n <= 0 || n > CBV_MAX_SQUARES || (cn != 1 && cn != 3)) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_squares_load_dev: bad arguments");
    CBV_ENTER(ctx);
    int ws[CBV_MAX_SQUARES], hs[CBV_MAX_SQUARES];
    for (int i = 0; i < n; i++) {
        if (rois[i].x0 < 0 || rois[i].y0 < 0 || rois[i].x0 + rois[i].w > w || rois[i].y0 + rois[i].h > h)
            return cbv_fail(ctx, CBV_ERR_ARG, "cbv_squares_load_dev: roi %d outside the image", i);
        ws[i] = rois[i].w;
        hs[i] = rois[i].h;
    }
    RC(squares_set_geometry(s, ws, hs, n));
    RC(squares_set_coef(s, blur_k));
    for (int i = 0; i < n; i++) {
        s->descs[i].cn = cn;
        s->descs[i].stride = stride;
        s->descs[i].src_off = rois[i].y0 * stride + rois[i].x0 * cn;
    }
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CBV_HIP(ctx, hipMemcpyAsync(s->d_descs.p, s->descs.data(), sizeof(SquareDesc) * n, hipMemcpyHostToDevice, ctx->stream));
    s->descs_on_dev.clear();
    RC(launch_squares_preprocess(ctx, (const u8*)dev_img, 0, (const SquareDesc*)s->d_descs.p, n, (const int*)s->d_coef.p, s->blur_k,
                                 (u8*)s->d_gray.p, 0, 1));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

static int squares_select(cbv_squares* s, const uint8_t* select, const u8** dev)
{
    *dev = nullptr;
    if (!select) return CBV_OK;
    cbv_ctx* ctx = s->ctx;
    CBV_HIP(ctx, hipMemcpyAsync(s->d_select.p, select, s->n, hipMemcpyHostToDevice, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *dev = (const u8*)s->d_select.p;
    return CBV_OK;
}

extern "C" int cbv_squares_calibrate(cbv_squares* s, double initial_variance, const uint8_t* select)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    CBV_ENTER(ctx);
    if (s->n == 0) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_squares_calibrate: no squares loaded");
    const u8* sel;
    RC(squares_select(s, select, &sel));
    RC(launch_squares_calibrate(ctx, (const SquareDesc*)s->d_descs.p, s->n, (const u8*)s->d_gray.p, (float*)s->d_mean.p, (float*)s->d_var.p,
                                sq_sd(s), (float)initial_variance, sel));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->has_model = true;
    return CBV_OK;
}

extern "C" int cbv_squares_ema(cbv_squares* s, double alpha, const uint8_t* select)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    CBV_ENTER(ctx);
    if (!s->has_model) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_squares_ema: not calibrated");
    const u8* sel;
    RC(squares_select(s, select, &sel));
    RC(launch_squares_ema(ctx, (const SquareDesc*)s->d_descs.p, s->n, (const u8*)s->d_gray.p, (float*)s->d_mean.p, (float*)s->d_var.p, sq_sd(s), alpha, sel));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

extern "C" int cbv_squares_set_ref(cbv_squares* s, const uint8_t* select)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    CBV_ENTER(ctx);
    if (s->n == 0) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_squares_set_ref: no squares loaded");
    const u8* sel;
    RC(squares_select(s, select, &sel));
    RC(launch_squares_set_ref(ctx, (const SquareDesc*)s->d_descs.p, s->n, (const u8*)s->d_gray.p, (u8*)s->d_ref.p, sel));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->has_ref = true;
    return CBV_OK;
}

extern "C" int cbv_squares_stats(cbv_squares* s, int use_ref, int use_model, double z_threshold, cbv_sq_stats* out)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    CBV_ENTER(ctx);
    if (!out || s->n == 0) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_squares_stats: no squares loaded or null output");
    if (use_model && !s->has_model) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_squares_stats: model requested but not calibrated");
    RC(launch_squares_stats(ctx, (const SquareDesc*)s->d_descs.p, s->n, (const u8*)s->d_gray.p, 0, use_ref ? (const u8*)s->d_ref.p : nullptr,
                            use_model ? (const float*)s->d_mean.p : nullptr, use_model ? (const float*)sq_sd(s) : nullptr,
                            (const u8*)s->d_masks.p, (float)z_threshold, (cbv_sq_stats*)s->d_stats.p, 1));
    CBV_HIP(ctx, hipMemcpyAsync(out, s->d_stats.p, sizeof(cbv_sq_stats) * s->n, hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

// the checks of hough_cfg that need no square geometry (entry points run them before they change any state)
static int hough_params_check(cbv_ctx* ctx, const cbv_hough_params* prm)
{
    if (!prm || !(prm->dp > 0) || prm->param1 < 0 || prm->param2 < 0 || !(prm->max_radius_ratio >= 0) || !(prm->min_radius_ratio >= 0))
        return cbv_fail(ctx, CBV_ERR_ARG, "HoughCircles parameters are invalid");
    if (prm->max_radius_ratio > 4.0 || prm->min_radius_ratio > 4.0)
        return cbv_fail(ctx, CBV_ERR_ARG, "HoughCircles radius ratios above 4 squares are not supported (got %g, %g)", prm->min_radius_ratio,
                        prm->max_radius_ratio);
    return CBV_OK;
}

static int hough_cfg(cbv_ctx* ctx, const cbv_hough_params* prm, const std::vector<SquareDesc>& descs, HoughCfg* hc)
{
    RC(hough_params_check(ctx, prm));
    memset(hc, 0, sizeof(*hc));
    hc->dp = (float)prm->dp < 1.f ? 1.f : (float)prm->dp;
    hc->canny_thr = (int)nearbyint(prm->param1);
    hc->acc_thr = (int)nearbyint(prm->param2);
    hc->min_ratio = prm->min_radius_ratio;
    hc->max_ratio = prm->max_radius_ratio;
    for (const SquareDesc& d : descs) {
        hc->maxw = std::max(hc->maxw, d.w);
        hc->maxh = std::max(hc->maxh, d.h);
    }
    return CBV_OK;
}

extern "C" int cbv_squares_hough(cbv_squares* s, const cbv_hough_params* prm, cbv_hough_result* out)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    CBV_ENTER(ctx);
    if (!out || s->n == 0) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_squares_hough: no squares loaded or null output");
    HoughCfg hc;
    RC(hough_cfg(ctx, prm, s->descs, &hc));
    RC(dev_ensure(ctx, &s->d_hough, sizeof(cbv_hough_result) * CBV_MAX_SQUARES));
    RC(dev_ensure(ctx, &s->d_retry, sizeof(u32) * (1 + CBV_MAX_SQUARES)));
    CBV_HIP(ctx, hipMemsetAsync(s->d_retry.p, 0, sizeof(u32), ctx->stream));
    RC(launch_hough(ctx, (const SquareDesc*)s->d_descs.p, s->n, (const u8*)s->d_gray.p, 0, hc, (cbv_hough_result*)s->d_hough.p, nullptr, nullptr, 1,
                    (u32*)s->d_retry.p, 0));
    RC(launch_hough_second(ctx, (const SquareDesc*)s->d_descs.p, s->n, (const u8*)s->d_gray.p, 0, hc, (cbv_hough_result*)s->d_hough.p, nullptr,
                           (const u32*)s->d_retry.p, s->n));
    CBV_HIP(ctx, hipMemcpyAsync(out, s->d_hough.p, sizeof(cbv_hough_result) * s->n, hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// one call per frame for the reference's own call pattern (game_session.py:124-161, calibrate_sensitivity.py:142-157):
// the 64 squares are views of ONE host image (split_board of the warped board), so the image's bounding rows are
// uploaded with one copy AT CALL TIME (nothing stale, no host pointer kept) and everything up to the per-square records
// runs behind it with one wait at the end.
// ---------------------------------------------------------------------------
extern "C" int cbv_debug_poison(cbv_ctx* ctx, int on)
{
    if (!ctx) return CBV_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    ctx->debug_poison = on ? 1 : 0;
    return CBV_OK;
}

static bool same_desc(const SquareDesc& a, const SquareDesc& b)
{
    return a.w == b.w && a.h == b.h && a.src_off == b.src_off && a.stride == b.stride && a.cn == b.cn && a.plane_off == b.plane_off &&
           a.mask_off == b.mask_off;
}

// geometry + upload of the rows of `img` the ROIs cover + descriptors; leaves the stream with d_stage / d_descs ready.
// `hst` (pinned, >= sizeof(SquareDesc) * n) stages the descriptors when they changed.
static int squares_stage_host_image(cbv_squares* s, const cbv_host_image* img, const cbv_roi* rois, int n, int blur_k, u8* hst)
{
    cbv_ctx* ctx = s->ctx;
    if (!img || !img->data || !rois || n <= 0 || n > CBV_MAX_SQUARES || (img->cn != 1 && img->cn != 3) || img->w <= 0 || img->h <= 0 ||
        img->stride < img->w * img->cn)
        return cbv_fail(ctx, CBV_ERR_ARG, "host image / roi arguments are invalid (n=%d)", n);
    int ws[CBV_MAX_SQUARES], hs[CBV_MAX_SQUARES];
    int y0 = img->h, y1 = 0;
    for (int i = 0; i < n; i++) {
        const cbv_roi& r = rois[i];
        if (r.w <= 0 || r.h <= 0 || r.x0 < 0 || r.y0 < 0 || r.x0 + r.w > img->w || r.y0 + r.h > img->h)
            return cbv_fail(ctx, CBV_ERR_ARG, "roi %d (%d,%d %dx%d) is outside the %dx%d image", i, r.x0, r.y0, r.w, r.h, img->w, img->h);
        ws[i] = r.w;
        hs[i] = r.h;
        y0 = std::min(y0, r.y0);
        y1 = std::max(y1, r.y0 + r.h);
    }
    RC(squares_set_geometry(s, ws, hs, n));
    RC(squares_set_coef(s, blur_k));
    // One contiguous copy of the rows the ROIs cover.  When the image's rows are padded or it is a crop of a wider frame,
    // the bytes between its rows travel too as long as that at most doubles the copy (they lie inside the same
    // allocation: between the first and the last byte of the image); a much wider pitch is gathered by a 2-D copy,
    // which is several times slower per byte on this platform (tools/ubench_copy.hip).
    const int rowb = img->w * img->cn;
    const bool one_d = img->stride <= 2 * rowb;
    const int pitch = one_d ? img->stride : rowb;
    for (int i = 0; i < n; i++) {
        s->descs[i].cn = img->cn;
        s->descs[i].stride = pitch;
        s->descs[i].src_off = (rois[i].y0 - y0) * pitch + rois[i].x0 * img->cn;
    }
    const size_t bytes = (size_t)pitch * (y1 - y0 - 1) + rowb;
    RC(dev_ensure(ctx, &s->d_stage, bytes + 16)); // (+16: the 12-byte loads of the last pixels of the last row)
    if (ctx->debug_poison) CBV_HIP(ctx, hipMemsetAsync(s->d_stage.p, 0xA5, bytes + 16, ctx->stream));
    const u8* src = img->data + (size_t)y0 * img->stride;
    if (one_d) CBV_HIP(ctx, hipMemcpyAsync(s->d_stage.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    else CBV_HIP(ctx, hipMemcpy2DAsync(s->d_stage.p, rowb, src, img->stride, rowb, y1 - y0, hipMemcpyHostToDevice, ctx->stream));
    bool same = (int)s->descs_on_dev.size() == n;
    for (int i = 0; same && i < n; i++) same = same_desc(s->descs_on_dev[i], s->descs[i]);
    if (!same) {
        memcpy(hst, s->descs.data(), sizeof(SquareDesc) * n);
        CBV_HIP(ctx, hipMemcpyAsync(s->d_descs.p, hst, sizeof(SquareDesc) * n, hipMemcpyHostToDevice, ctx->stream));
        s->descs_on_dev = s->descs;
    }
    return CBV_OK;
}

static int max_px_of(const cbv_squares* s)
{
    int m = 0;
    for (const SquareDesc& d : s->descs) m = std::max(m, d.w * d.h);
    return m;
}

extern "C" int cbv_squares_load_image(cbv_squares* s, const cbv_host_image* img, const cbv_roi* rois, int n, int blur_k)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    CBV_ENTER(ctx);
    u8* hst;
    RC(ctx_hstage(ctx, 65536, &hst));
    RC(squares_stage_host_image(s, img, rois, n, blur_k, hst));
    RC(launch_squares_preprocess(ctx, (const u8*)s->d_stage.p, 0, (const SquareDesc*)s->d_descs.p, n, (const int*)s->d_coef.p, s->blur_k,
                                 (u8*)s->d_gray.p, 0, 1, max_px_of(s)));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the host image (and the pinned descriptors) are free again
    return CBV_OK;
}

extern "C" int cbv_squares_set_ref_mask(cbv_squares* s, uint64_t mask)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    CBV_ENTER(ctx);
    if (s->n == 0) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_squares_set_ref_mask: no squares loaded");
    RC(launch_squares_set_ref_mask(ctx, (const SquareDesc*)s->d_descs.p, s->n, (const u8*)s->d_gray.p, (u8*)s->d_ref.p, mask));
    s->has_ref = true;
    return CBV_OK; // asynchronous: later calls on this context are ordered behind it
}

// PieceDetector.detect_piece (piece_detector.py:289-345) for one square from its statistics and its HoughCircles
// record, in the reference's arithmetic: np.std as an exact integer test, means and differences as float64
// quotients, np.var of the ring means summed left to right like numpy does for fewer than eight elements.
extern "C" int cbv_decide_piece(const cbv_sq_stats* st, const cbv_hough_result* hg, int w, int h, double circle_threshold,
                                cbv_piece_result* out)
{
    if (!st || !out) return CBV_ERR_ARG;
    out->has_piece = 0;
    out->method = CBV_METHOD_NONE;
    out->cx = out->cy = out->radius = 0;
    out->confidence = 0.0;
    out->center_border_diff = 0.0;
    const long long n = st->n, sm = st->sum;
    if (n * (long long)st->sumsq - sm * sm < 225ll * n * n) return CBV_OK; // np.std(gray) < 15: nothing else is tried
    if (hg && (hg->flags & CBV_HOUGH_OVERFLOW)) return CBV_ERR_UNSUPPORTED; // never passed on as HoughCircles' answer
    if (hg && hg->found) {
        out->has_piece = 1;
        out->method = hg->kind == 2 ? CBV_METHOD_TOWER_TOP : CBV_METHOD_HOUGH;
        out->cx = (int)hg->cx; // int(np.float32): toward zero
        out->cy = (int)hg->cy;
        out->radius = (int)hg->r;
        out->confidence = hg->kind == 2 ? 0.75 : 0.9;
        return CBV_OK;
    }
    const double zero = 0.0;
    const double cm = st->center_cnt ? (double)st->center_sum / (double)st->center_cnt : zero / zero; // np.mean of nothing is nan
    const double bm = st->border_cnt ? (double)st->border_sum / (double)st->border_cnt : zero / zero;
    const double diff = fabs(cm - bm);
    out->center_border_diff = diff;
    const int md = w < h ? w : h;
    if (diff > 40) {
        out->has_piece = 1;
        out->method = CBV_METHOD_CENTER_DIFF;
        out->cx = w / 2;
        out->cy = h / 2;
        out->radius = md / 3;
        out->confidence = diff / 80 < 1.0 ? diff / 80 : 1.0;
        return CBV_OK;
    }
    double rm[4];
    int nr = 0;
    for (int k = 0; k < 4; k++)
        if (st->ring_cnt[k] > 0) rm[nr++] = (double)st->ring_sum[k] / (double)st->ring_cnt[k];
    double symmetry = 0.0;
    if (nr >= 2) {
        double sum = 0;
        for (int k = 0; k < nr; k++) sum = sum + rm[k];
        const double mean = sum / nr;
        double sq = 0;
        for (int k = 0; k < nr; k++) {
            const double x = rm[k] - mean;
            sq = sq + x * x;
        }
        const double var = sq / nr;
        symmetry = var / 500 < 1.0 ? var / 500 : 1.0;
    }
    if (symmetry > circle_threshold) {
        out->has_piece = 1;
        out->method = CBV_METHOD_SYMMETRY;
        out->cx = w / 2;
        out->cy = h / 2;
        out->radius = md / 3;
        out->confidence = symmetry;
    }
    return CBV_OK;
}

// device scratch of the one-call entry points: worklist | retry list | then everything that travels back to the host as
// ONE copy: gate flags | statistics | second statistics set (k = 5 planes) | HoughCircles records
struct FastLayout {
    u32* work;
    u32* retry;
    u8* out;       // start of the block that is copied back
    u8* dflags;
    cbv_sq_stats* stats;
    cbv_sq_stats* stats5;
    cbv_hough_result* hough;
    size_t o_flags, o_stats, o_stats5, o_hough, out_bytes; // offsets inside the block
};
static int fast_layout(cbv_squares* s, FastLayout* L)
{
    const size_t o_retry = 512, o_out = 1024; // (cbv_squares_detect_all copies [retry, out + out_bytes) back as one block)
    // the block is copied back to offset 8192 of the 64 KB pinned staging area, whose tail (from 40960) holds the host-built worklist
    static_assert(8192 + 64 + (2 * sizeof(cbv_sq_stats) + sizeof(cbv_hough_result)) * CBV_MAX_SQUARES <= 40960, "result block overruns the staging area");
    static_assert(sizeof(SquareDesc) * CBV_MAX_SQUARES <= 8192 && 40960 + 4 * (1 + CBV_MAX_SQUARES) <= 65536, "staging layout");
    L->o_flags = 0;
    L->o_stats = 64;
    L->o_stats5 = L->o_stats + sizeof(cbv_sq_stats) * CBV_MAX_SQUARES;
    L->o_hough = L->o_stats5 + sizeof(cbv_sq_stats) * CBV_MAX_SQUARES;
    L->out_bytes = L->o_hough + sizeof(cbv_hough_result) * CBV_MAX_SQUARES;
    RC(dev_ensure(s->ctx, &s->d_fast, o_out + L->out_bytes));
    u8* b = (u8*)s->d_fast.p;
    L->work = (u32*)b;
    L->retry = (u32*)(b + o_retry);
    L->out = b + o_out;
    L->dflags = L->out + L->o_flags;
    L->stats = (cbv_sq_stats*)(L->out + L->o_stats);
    L->stats5 = (cbv_sq_stats*)(L->out + L->o_stats5);
    L->hough = (cbv_hough_result*)(L->out + L->o_hough);
    return CBV_OK;
}

extern "C" int cbv_squares_detect_all(cbv_squares* s, const cbv_host_image* img, const cbv_roi* rois, int n,
                                      const cbv_detect_params* prm, cbv_piece_result* out)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    if (!prm || !out) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_squares_detect_all: null argument");
    CBV_ENTER(ctx);
    RC(hough_params_check(ctx, &prm->hough)); // before anything of the set changes
    u8* hst;
    const size_t o_back = 8192; // the descriptors (64 x 64 B) are staged in front of it
    RC(ctx_hstage(ctx, 65536, &hst));
    FastLayout L;
    RC(fast_layout(s, &L));
    // worklist and retry counters; in front of the upload, whose host-blocking copy would otherwise leave the GPU idle
    // while this launch is submitted
    CBV_HIP(ctx, hipMemsetAsync(s->d_fast.p, 0, 1024, ctx->stream));
    RC(squares_stage_host_image(s, img, rois, n, 5, hst));
    HoughCfg hc;
    RC(hough_cfg(ctx, &prm->hough, s->descs, &hc));
    DetectMasks dm;
    dm.has_ref = s->has_ref ? prm->has_ref : 0;
    dm.cached = prm->cached;
    dm.check = prm->check;
    dm.check_given = prm->check_given;
    dm.use_delta = prm->use_delta;
    dm.change_threshold = prm->change_threshold;
    dm.dflags = L.dflags;
    // preprocess + statistics (+ |gray - reference|) + the gate, one launch; HoughCircles over the squares it listed
    RC(launch_squares_pre5_stats(ctx, (const u8*)s->d_stage.p, 0, (const SquareDesc*)s->d_descs.p, n, (u8*)s->d_gray.p, 0, nullptr, nullptr,
                                 (const u8*)s->d_masks.p, 0.f, L.stats, 1, nullptr, 1, L.work, L.hough, max_px_of(s), (const u8*)s->d_ref.p, &dm));
    RC(launch_hough(ctx, (const SquareDesc*)s->d_descs.p, n, (const u8*)s->d_gray.p, 0, hc, L.hough, nullptr, L.work, 1, L.retry, 0));
    // the second-pass list travels back in front of the results (it sits 512 bytes before them): normally it is empty
    // and the second pass, a launch of its own, is never enqueued
    const size_t o_retry_back = o_back - 512;
    CBV_HIP(ctx, hipMemcpyAsync(hst + o_retry_back, L.retry, 512 + L.out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*(const u32*)(hst + o_retry_back) != 0) {
        RC(launch_hough_second(ctx, (const SquareDesc*)s->d_descs.p, n, (const u8*)s->d_gray.p, 0, hc, L.hough, nullptr, L.retry, n));
        CBV_HIP(ctx, hipMemcpyAsync(hst + o_back + L.o_hough, L.hough, sizeof(cbv_hough_result) * CBV_MAX_SQUARES, hipMemcpyDeviceToHost, ctx->stream));
        CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const cbv_sq_stats* st = (const cbv_sq_stats*)(hst + o_back + L.o_stats);
    const cbv_hough_result* hg = (const cbv_hough_result*)(hst + o_back + L.o_hough);
    const u8* fl = hst + o_back + L.o_flags;
    for (int i = 0; i < n; i++) {
        cbv_piece_result r;
        memset(&r, 0, sizeof(r));
        if (fl[i] & 4) { // detect_piece is evaluated for this square
            const bool hough_ran = (fl[i] & 8) != 0;
            if (cbv_decide_piece(&st[i], hough_ran ? &hg[i] : nullptr, s->descs[i].w, s->descs[i].h, prm->circle_threshold, &r) != CBV_OK)
                return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "HoughCircles candidate list overflowed (more accumulator maxima than the second pass holds in a "
                                "%dx%d square)", s->descs[i].w, s->descs[i].h);
        }
        r.changed = fl[i] & 1;
        r.should_process = (fl[i] >> 1) & 1;
        r.evaluated = (fl[i] >> 2) & 1;
        out[i] = r;
    }
    return CBV_OK;
}

extern "C" int cbv_squares_detect_changes(cbv_squares* s, const cbv_host_image* img, const cbv_roi* rois, int n, int blur_k,
                                          const cbv_change_params* prm, cbv_change_result* out)
{
    if (!s) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    if (!prm || !out) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_squares_detect_changes: null argument");
    CBV_ENTER(ctx);
    if (!s->has_model) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_squares_detect_changes: not calibrated");
    RC(hough_params_check(ctx, &prm->hough));
    u8* hst;
    const size_t o_back = 8192, o_wk = 40960;
    RC(ctx_hstage(ctx, 65536, &hst));
    const int old_n = s->n;
    RC(squares_stage_host_image(s, img, rois, n, blur_k, hst));
    if (!s->has_model || s->n != old_n) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_squares_detect_changes: the squares' geometry changed since calibrate");
    FastLayout L;
    RC(fast_layout(s, &L));
    const int mpx = max_px_of(s);
    const bool five = s->blur_k == 5;
    if (five) {
        RC(launch_squares_pre5_stats(ctx, (const u8*)s->d_stage.p, 0, (const SquareDesc*)s->d_descs.p, n, (u8*)s->d_gray.p, 0, (const float*)s->d_mean.p,
                                     (const float*)sq_sd(s), (const u8*)s->d_masks.p, (float)prm->z_threshold, L.stats, 1, nullptr, 0, nullptr, nullptr,
                                     mpx));
    } else {
        // the detector's own blur for the background model, and the squares as PieceDetector preprocesses them (k = 5)
        // for the is_circular test of the squares that changed
        RC(launch_squares_preprocess(ctx, (const u8*)s->d_stage.p, 0, (const SquareDesc*)s->d_descs.p, n, (const int*)s->d_coef.p, s->blur_k,
                                     (u8*)s->d_gray.p, 0, 1, mpx));
        RC(launch_squares_stats(ctx, (const SquareDesc*)s->d_descs.p, n, (const u8*)s->d_gray.p, 0, nullptr, (const float*)s->d_mean.p,
                                (const float*)sq_sd(s), (const u8*)s->d_masks.p, (float)prm->z_threshold, L.stats, 1));
        RC(dev_ensure(ctx, &s->d_gray5, s->plane_total));
        RC(launch_squares_pre5_stats(ctx, (const u8*)s->d_stage.p, 0, (const SquareDesc*)s->d_descs.p, n, (u8*)s->d_gray5.p, 0, nullptr, nullptr,
                                     (const u8*)s->d_masks.p, 0.f, L.stats5, 1, nullptr, 0, nullptr, nullptr, mpx));
    }
    CBV_HIP(ctx, hipMemcpyAsync(hst + o_back + L.o_stats, L.stats, sizeof(cbv_sq_stats) * CBV_MAX_SQUARES * (five ? 1 : 2), hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const cbv_sq_stats* st = (const cbv_sq_stats*)(hst + o_back + L.o_stats);
    const cbv_sq_stats* st5 = five ? st : (const cbv_sq_stats*)(hst + o_back + L.o_stats5);
    // change_detector.py:124-150: squares of the selection whose share of changed pixels reaches 5 %
    u32* work = (u32*)(hst + o_wk);
    int nwork = 0;
    for (int i = 0; i < n; i++) {
        cbv_change_result r;
        memset(&r, 0, sizeof(r));
        r.z_max = st[i].z_max;
        r.z_count = st[i].z_count;
        r.n = st[i].n;
        if ((prm->select >> i) & 1ull) {
            const double pct = ((double)st[i].z_count / (double)st[i].n) * 100.0;
            if (!(pct < 5.0)) {
                r.in_result = 1;
                r.intensity = pct > 75.0 ? 3 : (pct > 15.0 ? 2 : 1);
                const long long nn = st5[i].n, sm = st5[i].sum;
                if (!(nn * (long long)st5[i].sumsq - sm * sm < 225ll * nn * nn)) work[1 + nwork++] = (u32)i; // HoughCircles runs on it
            }
        }
        out[i] = r;
    }
    const cbv_hough_result* hg = nullptr;
    if (nwork) {
        HoughCfg hc;
        RC(hough_cfg(ctx, &prm->hough, s->descs, &hc));
        work[0] = (u32)nwork;
        CBV_HIP(ctx, hipMemcpyAsync(L.work, work, sizeof(u32) * (1 + nwork), hipMemcpyHostToDevice, ctx->stream));
        CBV_HIP(ctx, hipMemsetAsync(L.retry, 0, sizeof(u32), ctx->stream));
        const u8* g5 = five ? (const u8*)s->d_gray.p : (const u8*)s->d_gray5.p;
        RC(launch_hough(ctx, (const SquareDesc*)s->d_descs.p, n, g5, 0, hc, L.hough, nullptr, L.work, 1, L.retry, 0));
        u32* retry_back = (u32*)(hst + o_wk + 1024);
        CBV_HIP(ctx, hipMemcpyAsync(retry_back, L.retry, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
        CBV_HIP(ctx, hipMemcpyAsync(hst + o_back + L.o_hough, L.hough, sizeof(cbv_hough_result) * n, hipMemcpyDeviceToHost, ctx->stream));
        CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (*retry_back != 0) { // (normally empty: the second pass is a launch of its own)
            RC(launch_hough_second(ctx, (const SquareDesc*)s->d_descs.p, n, g5, 0, hc, L.hough, nullptr, L.retry, n));
            CBV_HIP(ctx, hipMemcpyAsync(hst + o_back + L.o_hough, L.hough, sizeof(cbv_hough_result) * n, hipMemcpyDeviceToHost, ctx->stream));
            CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        hg = (const cbv_hough_result*)(hst + o_back + L.o_hough);
    }
    for (int i = 0; i < n; i++) {
        if (!out[i].in_result) continue;
        bool ran = false;
        for (int k = 0; k < nwork && !ran; k++) ran = work[1 + k] == (u32)i;
        cbv_piece_result pr;
        if (cbv_decide_piece(&st5[i], ran ? &hg[i] : nullptr, s->descs[i].w, s->descs[i].h, prm->circle_threshold, &pr) != CBV_OK)
            return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "HoughCircles candidate list overflowed in a %dx%d square", s->descs[i].w, s->descs[i].h);
        out[i].is_circular = pr.has_piece;
    }
    return CBV_OK;
}

static int squares_plane(cbv_squares* s, int which, int index, void** p, size_t* bytes)
{
    cbv_ctx* ctx = s->ctx;
    if (index < 0 || index >= s->n) return cbv_fail(ctx, CBV_ERR_ARG, "square index %d out of range", index);
    const SquareDesc& d = s->descs[index];
    size_t n = (size_t)d.w * d.h;
    switch (which) {
    case 0: *p = (u8*)s->d_gray.p + d.plane_off; *bytes = n; break;
    case 1: *p = (u8*)s->d_ref.p + d.plane_off; *bytes = n; break;
    case 2: *p = (float*)s->d_mean.p + d.plane_off; *bytes = n * 4; break;
    case 3: *p = (float*)s->d_var.p + d.plane_off; *bytes = n * 4; break;
    case 4: *p = sq_sd(s) + d.plane_off; *bytes = n * 4; break; // sqrt(var) as the statistics kernels read it (inspection)
    default: return cbv_fail(ctx, CBV_ERR_ARG, "bad plane selector %d", which);
    }
    return CBV_OK;
}

extern "C" int cbv_squares_get(cbv_squares* s, int which, int index, void* out)
{
    if (!s || !out) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    void* p;
    size_t bytes;
    CBV_ENTER(ctx);
    RC(squares_plane(s, which, index, &p, &bytes));
    CBV_HIP(ctx, hipMemcpyAsync(out, p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

extern "C" int cbv_squares_set(cbv_squares* s, int which, int index, const void* in)
{
    if (!s || !in) return CBV_ERR_ARG;
    cbv_ctx* ctx = s->ctx;
    void* p;
    size_t bytes;
    CBV_ENTER(ctx);
    RC(squares_plane(s, which, index, &p, &bytes));
    CBV_HIP(ctx, hipMemcpyAsync(p, in, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (which == 3) RC(launch_squares_refresh_sd(ctx, (const SquareDesc*)s->d_descs.p, (const float*)s->d_var.p, sq_sd(s), index));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (which == 1) s->has_ref = true;
    return CBV_OK;
}

extern "C" int cbv_squares_geometry(cbv_squares* s, int index, int* w, int* h)
{
    if (!s || index < 0 || index >= s->n) return CBV_ERR_ARG;
    if (w) *w = s->descs[index].w;
    if (h) *h = s->descs[index].h;
    return CBV_OK;
}

// ---------------------------------------------------------------------------
// device-resident batched pipeline
// ---------------------------------------------------------------------------
struct cbv_pipeline {
    cbv_ctx* ctx = nullptr;
    int w = 0, h = 0, max_frames = 0;
    Geom g;
    cbv_pipeline_config cfg;
    bool configured = false;
    int chunk = 8;
    double Minv[9];
    u8* frames = nullptr;
    // Lanes: chunk c runs on lane c % n_lanes, each lane with its own HIP stream and scratch, so a
    // VALU-bound bilateral launch of one chunk overlaps the memory-latency-bound kernels of another.
    enum { MAX_LANES = 4 };
    int n_lanes = 1;
    hipStream_t lane_stream[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_done[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t start_ev = nullptr;
    u8* A[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    u8* B[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    u8* C[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr}; // third scratch frame set: region-limited enhancement only
    bool use_region = false;                                  // cfg.enhance_region, keep_enhanced == 0, a usable footprint
    PxRect region = {0, 0, 0, 0};                             // source pixels the warp samples (+ margin), clipped
    DevBuf lane_small[MAX_LANES];
    DevBuf lane_work[MAX_LANES]; // HoughCircles worklist of the lane's current chunk: count, then frame << 8 | square
    u8* enhanced = nullptr; // [max_frames] when keep_enhanced
    u8* warped = nullptr;   // [max_frames][S][S][3]
    size_t warped_stride = 0;
    DevBuf d_descs, d_masks, d_gray, d_stats, d_ref, d_state, d_results, d_flags, d_dec, d_coef, d_synth, d_mean, d_var, d_noise, d_noise_state, d_hough, d_check, d_hough_over;
    bool has_check = false; // squares_to_check masks were set
    HoughCfg hough_cfg;
    bool calibrated = false;
    std::vector<SquareDesc> descs;
    size_t plane_total = 0;
    // ingest: pinned host mirror of the frame ring, filled by the capture side and copied on its own stream
    u8* host_ring = nullptr;
    u8* h_stage = nullptr; // pinned mirror of d_results ([max_frames] records, then the HoughCircles overflow word): written by
                           // the last kernel of a SHORT run (ResultMirror), by a copy otherwise; read by cbv_pipeline_results
    size_t h_stage_bytes = 0;
    std::vector<u8> slot_mirrored; // per slot: the mirror holds the slot's newest record (once its run has finished)
    hipStream_t copy_stream = nullptr;
    struct CopyRec {
        int s0, cnt;
        hipEvent_t ev;
        bool pending;
    };
    std::vector<CopyRec> copies;
    // The temporal scan (+ NoiseHandler) of a run goes to its own stream behind the lanes' events, so the next run's
    // enhancement of OTHER slots overlaps it; scans of successive runs stay ordered on that stream.  (Runs of one or
    // two frames keep their scan on the caller's stream, after waiting for every run in flight: see cbv_pipeline_run.)
    hipStream_t scan_stream = nullptr;
    hipEvent_t main_done = nullptr;
    // Every run that may still be executing: its slot range and two events on the (in-order) scan stream,
    // `lanes_ev` = all lanes have read the input frames and written the per-slot buffers, `scan_ev` = the scan has
    // read them.  A later run (or ingest copy) that touches overlapping slots waits on the NEWEST overlapping
    // record, which covers the older ones because the scan stream is in order.  Records are recycled once their
    // scan event has completed.
    struct RunRec {
        int s0, cnt;
        unsigned long long seq;
        hipEvent_t lanes_ev, scan_ev;
        bool live;
        bool one_event; // a run of a frame or two, all on the caller's stream: only scan_ev is recorded (an event between two
                        // kernels is a ~5 us bubble in a 150 us chain), and it stands for lanes_ev too
        DevBuf retry; // HoughCircles second-pass list of this run (HoughCfg::retry), frames numbered from the run's slot0
    };
    std::vector<RunRec> runs;
    unsigned long long run_seq = 0;         // sequence number of the newest run
    unsigned long long joined_seq = 0;      // runs up to this one are ordered before later work on `joined_stream`
    hipStream_t joined_stream = nullptr;
    int max_px = 0; // pixels of the largest square
    bool keep_enhanced = false;
};

// the HoughCircles overflow word of the pinned result mirror (behind its max_frames records)
static u32* pipeline_over_word(cbv_pipeline* p) { return (u32*)(p->h_stage + ((sizeof(cbv_frame_result) * (size_t)p->max_frames + 7) & ~(size_t)7)); }

static bool ranges_overlap(int a0, int an, int b0, int bn) { return a0 < b0 + bn && b0 < a0 + an; }

static void retire_runs(cbv_pipeline* p)
{
    for (auto& r : p->runs)
        if (r.live && hipEventQuery(r.scan_ev) == hipSuccess) r.live = false;
}

// newest record that is still in flight, not yet ordered before the context's stream, and overlaps the slots
// (cnt <= 0: any slots)
static cbv_pipeline::RunRec* newest_unjoined(cbv_pipeline* p, int s0, int cnt)
{
    if (p->joined_stream != p->ctx->stream) { // the caller switched streams: nothing is ordered before the new one
        p->joined_stream = p->ctx->stream;
        p->joined_seq = 0;
    }
    cbv_pipeline::RunRec* best = nullptr;
    for (auto& r : p->runs)
        if (r.live && r.seq > p->joined_seq && (cnt <= 0 || ranges_overlap(s0, cnt, r.s0, r.cnt)) && (!best || r.seq > best->seq)) best = &r;
    return best;
}

// make the context's stream wait for every run that is still in flight (lanes and scans)
static int join_scan(cbv_pipeline* p)
{
    cbv_ctx* ctx = p->ctx;
    if (cbv_pipeline::RunRec* r = newest_unjoined(p, 0, 0)) {
        CBV_HIP(ctx, hipStreamWaitEvent(ctx->stream, r->scan_ev, 0));
        p->joined_seq = r->seq;
    }
    return CBV_OK;
}

// make the context's stream wait for the runs in flight that touch these slots (older scans of the same slots
// may still be queued: the newest overlapping record covers them)
static int join_slots(cbv_pipeline* p, int s0, int cnt)
{
    cbv_ctx* ctx = p->ctx;
    retire_runs(p);
    if (cbv_pipeline::RunRec* r = newest_unjoined(p, s0, cnt)) {
        CBV_HIP(ctx, hipStreamWaitEvent(ctx->stream, r->scan_ev, 0));
        // everything up to r is ordered now; records between joined_seq and r.seq that do not overlap are too
        p->joined_seq = std::max(p->joined_seq, r->seq);
    }
    return CBV_OK;
}

extern "C" int cbv_pipeline_create(cbv_ctx* ctx, int w, int h, int max_frames, cbv_pipeline** out)
{
    if (!ctx || !out || w <= 0 || h <= 0 || max_frames <= 0) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_pipeline_create: bad arguments");
    CBV_ENTER(ctx);
    cbv_pipeline* p = new cbv_pipeline();
    p->ctx = ctx;
    p->w = w;
    p->h = h;
    p->max_frames = max_frames;
    p->g = tight_geom(w, h);
    hipError_t e = hipMalloc((void**)&p->frames, p->g.frame_stride * max_frames + 256);
    if (e != hipSuccess) {
        const size_t want = p->g.frame_stride * max_frames;
        delete p;
        return cbv_fail(ctx, CBV_ERR_HIP, "hipMalloc of %zu bytes for the frame ring failed: %s", want, hipGetErrorString(e));
    }
    *out = p;
    return CBV_OK;
}

extern "C" void cbv_pipeline_destroy(cbv_pipeline* p)
{
    if (!p) return;
    std::lock_guard<std::recursive_mutex> lock(p->ctx->mu);
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    for (int l = 0; l < cbv_pipeline::MAX_LANES; l++) {
        if (p->lane_stream[l]) (void)hipStreamSynchronize(p->lane_stream[l]);
        if (p->A[l]) (void)hipFree(p->A[l]);
        if (p->B[l]) (void)hipFree(p->B[l]);
        if (p->C[l]) (void)hipFree(p->C[l]);
        dev_free(&p->lane_small[l]);
        dev_free(&p->lane_work[l]);
        if (p->lane_done[l]) (void)hipEventDestroy(p->lane_done[l]);
    }
    if (p->start_ev) (void)hipEventDestroy(p->start_ev);
    if (p->scan_stream) (void)hipStreamSynchronize(p->scan_stream); // (the worker streams belong to the context)
    for (auto& r : p->runs) {
        (void)hipEventDestroy(r.lanes_ev);
        (void)hipEventDestroy(r.scan_ev);
        dev_free(&r.retry);
    }
    if (p->main_done) (void)hipEventDestroy(p->main_done);
    if (p->copy_stream) (void)hipStreamSynchronize(p->copy_stream);
    for (auto& c : p->copies) (void)hipEventDestroy(c.ev);
    if (p->host_ring) (void)hipHostFree(p->host_ring);
    if (p->h_stage) (void)hipHostFree(p->h_stage);
    if (p->frames) (void)hipFree(p->frames);
    if (p->enhanced) (void)hipFree(p->enhanced);
    if (p->warped) (void)hipFree(p->warped);
    DevBuf* bufs[] = {&p->d_descs, &p->d_masks, &p->d_gray, &p->d_stats, &p->d_ref, &p->d_state, &p->d_results, &p->d_flags, &p->d_dec, &p->d_noise, &p->d_noise_state, &p->d_coef, &p->d_synth, &p->d_mean, &p->d_var, &p->d_hough, &p->d_check, &p->d_hough_over};
    for (auto b : bufs) dev_free(b);
    delete p;
}

extern "C" void* cbv_pipeline_frames_dev(cbv_pipeline* p) { return p ? p->frames : nullptr; }

extern "C" int cbv_pipeline_configure(cbv_pipeline* p, const cbv_pipeline_config* cfg)
{
    if (!p || !cfg) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    if (cfg->n_rois <= 0 || cfg->n_rois > CBV_MAX_SQUARES || cfg->board_size <= 0 || cfg->board_size > 4096)
        return cbv_fail(ctx, CBV_ERR_ARG, "cbv_pipeline_configure: bad board/roi configuration");
    if (cfg->history_size < 1 || cfg->history_size > 7) return cbv_fail(ctx, CBV_ERR_ARG, "history_size must be in 1..7");
    RC(check_params(ctx, &cfg->enhance));
    for (int i = 0; i < cfg->n_rois; i++) {
        const cbv_roi& r = cfg->rois[i];
        if (r.w <= 0 || r.h <= 0 || r.w > CBV_MAX_SQUARE_DIM || r.h > CBV_MAX_SQUARE_DIM || r.x0 < 0 || r.y0 < 0 ||
            r.x0 + r.w > cfg->board_size || r.y0 + r.h > cfg->board_size)
            return cbv_fail(ctx, CBV_ERR_ARG, "roi %d is invalid for a %dx%d board", i, cfg->board_size, cfg->board_size);
    }
    // every argument check that needs no state is done; from here on a failure leaves the pipeline UNconfigured
    // (run / results / ... return CBV_ERR_STATE) instead of half reconfigured
    p->configured = false;
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    p->cfg = *cfg;
    p->keep_enhanced = cfg->keep_enhanced != 0;
    int chunk = cfg->chunk;
    if (chunk <= 0) chunk = 32;
    if (chunk > p->max_frames) chunk = p->max_frames;
    p->chunk = chunk;
    if (!host_invert3x3(cfg->M, p->Minv)) memset(p->Minv, 0, sizeof(p->Minv));
    const int S = cfg->board_size;
    p->warped_stride = ((size_t)S * S * 3 + 255) & ~(size_t)255;
    // (re)allocate
    int lanes = cfg->lanes <= 0 ? 2 : cfg->lanes;
    if (lanes > cbv_pipeline::MAX_LANES) lanes = cbv_pipeline::MAX_LANES;
    if ((p->max_frames + chunk - 1) / chunk < lanes) lanes = (p->max_frames + chunk - 1) / chunk;
    p->n_lanes = lanes;
    for (int l = 0; l < cbv_pipeline::MAX_LANES; l++) {
        if (p->A[l]) (void)hipFree(p->A[l]);
        if (p->B[l]) (void)hipFree(p->B[l]);
        if (p->C[l]) (void)hipFree(p->C[l]);
        p->A[l] = p->B[l] = p->C[l] = nullptr;
    }
    // region-limited enhancement: the source footprint of the S x S warp = the image of the destination square under
    // Minv (a projective map keeps the square convex while W > 0 on it: its four corners bound it), + 3 px for the
    // 1/32-px rounding and the bilinear taps
    p->use_region = false;
    if (cfg->enhance_region && !p->keep_enhanced && sharpen_region_ok(cfg->enhance.sharpen_kernel))
        p->use_region = warp_footprint(p->Minv, S, S, p->w, p->h, &p->region);
    if (p->warped) (void)hipFree(p->warped);
    if (p->enhanced) (void)hipFree(p->enhanced);
    p->warped = p->enhanced = nullptr;
    if (!p->start_ev) CBV_HIP(ctx, hipEventCreateWithFlags(&p->start_ev, hipEventDisableTiming));
    for (int l = 0; l < lanes; l++) {
        CBV_HIP(ctx, hipMalloc((void**)&p->A[l], p->g.frame_stride * chunk + 256));
        CBV_HIP(ctx, hipMalloc((void**)&p->B[l], p->g.frame_stride * chunk + 256));
        if (p->use_region) CBV_HIP(ctx, hipMalloc((void**)&p->C[l], p->g.frame_stride * chunk + 256));
        SmallLayout SL;
        RC(small_layout(ctx, &p->lane_small[l], cfg->enhance.tiles_x * cfg->enhance.tiles_y, chunk, &SL, cfg->enhance.tiles_x, cfg->enhance.tiles_y));
        RC(dev_ensure(ctx, &p->lane_work[l], sizeof(u32) * (1 + (size_t)CBV_MAX_SQUARES * chunk)));
        if (l > 0) RC(ctx_worker_stream(ctx, &ctx->lane_streams[l], &p->lane_stream[l]));
        if (!p->lane_done[l]) CBV_HIP(ctx, hipEventCreateWithFlags(&p->lane_done[l], hipEventDisableTiming));
    }
    CBV_HIP(ctx, hipMalloc((void**)&p->warped, p->warped_stride * p->max_frames));
    if (p->keep_enhanced) CBV_HIP(ctx, hipMalloc((void**)&p->enhanced, p->g.frame_stride * p->max_frames + 256));
    // squares
    const int n = cfg->n_rois;
    p->descs.assign(n, SquareDesc());
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        const cbv_roi& r = cfg->rois[i];
        SquareDesc& d = p->descs[i];
        d.w = r.w;
        d.h = r.h;
        d.cn = 3;
        d.stride = S * 3;
        d.src_off = r.y0 * S * 3 + r.x0 * 3;
        d.plane_off = (int)off;
        d.mask_off = (int)off;
        off += ((size_t)r.w * r.h + 15) & ~(size_t)15;
    }
    p->plane_total = off;
    p->max_px = 0;
    for (int i = 0; i < n; i++) p->max_px = std::max(p->max_px, p->descs[i].w * p->descs[i].h);
    std::vector<u8> masks(off, 0);
    for (int i = 0; i < n; i++) {
        build_piece_mask(p->descs[i].w, p->descs[i].h, masks.data() + p->descs[i].mask_off);
        square_region_counts(masks.data() + p->descs[i].mask_off, p->descs[i].w * p->descs[i].h, p->descs[i].cnt);
    }
    RC(dev_ensure(ctx, &p->d_descs, sizeof(SquareDesc) * n));
    RC(dev_ensure(ctx, &p->d_masks, off));
    RC(dev_ensure(ctx, &p->d_gray, off * p->max_frames));
    RC(dev_ensure(ctx, &p->d_stats, sizeof(cbv_sq_stats) * n * p->max_frames));
    RC(dev_ensure(ctx, &p->d_ref, off));
    RC(dev_ensure(ctx, &p->d_mean, off * 4));
    RC(dev_ensure(ctx, &p->d_var, off * 8)); // variance plane, then its square root
    p->calibrated = false;
    RC(dev_ensure(ctx, &p->d_state, sizeof(ScanState) * n));
    RC(dev_ensure(ctx, &p->d_results, sizeof(cbv_frame_result) * p->max_frames));
    {
        const size_t want = sizeof(cbv_frame_result) * (size_t)p->max_frames + 16;
        if (p->h_stage_bytes < want) {
            if (p->h_stage) (void)hipHostFree(p->h_stage);
            p->h_stage = nullptr;
            p->h_stage_bytes = 0;
            CBV_HIP(ctx, hipHostMalloc((void**)&p->h_stage, want, hipHostMallocDefault));
            p->h_stage_bytes = want;
        }
        memset(p->h_stage, 0, p->h_stage_bytes);
        p->slot_mirrored.assign((size_t)p->max_frames, 0);
    }
    RC(dev_ensure(ctx, &p->d_flags, (size_t)CBV_MAX_SQUARES * p->max_frames));
    RC(dev_ensure(ctx, &p->d_dec, (size_t)CBV_MAX_SQUARES * p->max_frames));
    if (cfg->use_hough) {
        RC(hough_cfg(ctx, &cfg->hough, p->descs, &p->hough_cfg));
        RC(dev_ensure(ctx, &p->d_hough, sizeof(cbv_hough_result) * CBV_MAX_SQUARES * p->max_frames));
        RC(dev_ensure(ctx, &p->d_hough_over, 256));
        CBV_HIP(ctx, hipMemset(p->d_hough_over.p, 0, 256));
        p->hough_cfg.overflow_count = (u32*)p->d_hough_over.p;
    }
    RC(dev_ensure(ctx, &p->d_noise, sizeof(cbv_noise_result) * p->max_frames));
    RC(dev_ensure(ctx, &p->d_noise_state, sizeof(cbv_noise_state)));
    RC(dev_ensure(ctx, &p->d_check, sizeof(u64) * p->max_frames));
    CBV_HIP(ctx, hipMemset(p->d_check.p, 0, sizeof(u64) * p->max_frames));
    p->has_check = false;
    CBV_HIP(ctx, hipMemset(p->d_noise_state.p, 0, sizeof(cbv_noise_state)));
    int coef[32] = {0};
    build_gaussian_q8(5, coef);
    RC(dev_ensure(ctx, &p->d_coef, sizeof(coef)));
    CBV_HIP(ctx, hipMemcpy(p->d_coef.p, coef, sizeof(coef), hipMemcpyHostToDevice));
    CBV_HIP(ctx, hipMemcpy(p->d_descs.p, p->descs.data(), sizeof(SquareDesc) * n, hipMemcpyHostToDevice));
    CBV_HIP(ctx, hipMemcpy(p->d_masks.p, masks.data(), off, hipMemcpyHostToDevice));
    CBV_HIP(ctx, hipMemset(p->d_state.p, 0, sizeof(ScanState) * n));
    // plane padding (planes are rounded to 16 B) must read as zero in every frame: k_scan compares whole vectors
    CBV_HIP(ctx, hipMemset(p->d_gray.p, 0, off * p->max_frames));
    CBV_HIP(ctx, hipMemset(p->d_ref.p, 0, off));
    p->configured = true;
    return CBV_OK;
}

extern "C" int cbv_pipeline_reset_state(cbv_pipeline* p)
{
    if (!p || !p->configured) return CBV_ERR_STATE;
    cbv_ctx* ctx = p->ctx;
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    CBV_HIP(ctx, hipMemsetAsync(p->d_state.p, 0, sizeof(ScanState) * p->cfg.n_rois, ctx->stream));
    CBV_HIP(ctx, hipMemsetAsync(p->d_noise_state.p, 0, sizeof(cbv_noise_state), ctx->stream));
    if (p->d_hough_over.p) {
        CBV_HIP(ctx, hipMemsetAsync(p->d_hough_over.p, 0, 4, ctx->stream));
        CBV_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the runs that could still write the mirror's copy are behind us
        *pipeline_over_word(p) = 0;
    }
    return CBV_OK;
}

extern "C" int cbv_pipeline_calibrate(cbv_pipeline* p, int slot)
{
    if (!p || !p->configured) return CBV_ERR_STATE;
    cbv_ctx* ctx = p->ctx;
    if (slot < 0 || slot >= p->max_frames) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_pipeline_calibrate: bad slot");
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    RC(launch_squares_calibrate(ctx, (const SquareDesc*)p->d_descs.p, p->cfg.n_rois, (const u8*)p->d_gray.p + p->plane_total * slot,
                                (float*)p->d_mean.p, (float*)p->d_var.p, (float*)p->d_var.p + p->plane_total, (float)p->cfg.initial_variance, nullptr));
    p->calibrated = true;
    return CBV_OK;
}

extern "C" int cbv_pipeline_update_references(cbv_pipeline* p, int slot, int reset_noise)
{
    if (!p || !p->configured) return CBV_ERR_STATE;
    cbv_ctx* ctx = p->ctx;
    if (slot < 0 || slot >= p->max_frames) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_pipeline_update_references: bad slot");
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    RC(launch_scan_update_refs(ctx, (const SquareDesc*)p->d_descs.p, p->cfg.n_rois, (const u8*)p->d_gray.p + p->plane_total * slot,
                               (u8*)p->d_ref.p, (ScanState*)p->d_state.p));
    if (reset_noise) CBV_HIP(ctx, hipMemsetAsync(p->d_noise_state.p, 0, sizeof(cbv_noise_state), ctx->stream)); // NoiseHandler.reset()
    return CBV_OK;
}

extern "C" int cbv_pipeline_upload(cbv_pipeline* p, int slot, const uint8_t* bgr, int stride)
{
    if (!p || !bgr || slot < 0 || slot >= p->max_frames || stride < p->w * 3) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    RC(rows_h2d(ctx, p->frames + p->g.frame_stride * slot, bgr, stride, p->w * 3, p->h));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

extern "C" uint8_t* cbv_pipeline_host_ring(cbv_pipeline* p)
{
    if (!p) return nullptr;
    cbv_ctx* ctx = p->ctx;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (!p->host_ring) {
        if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
        if (hipHostMalloc((void**)&p->host_ring, p->g.frame_stride * p->max_frames, hipHostMallocDefault) != hipSuccess) {
            cbv_fail(ctx, CBV_ERR_HIP, "pinned host ring of %zu bytes could not be allocated", p->g.frame_stride * p->max_frames);
            p->host_ring = nullptr;
        }
    }
    return p->host_ring;
}

extern "C" int cbv_pipeline_submit(cbv_pipeline* p, int slot0, int count)
{
    if (!p) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    if (slot0 < 0 || count <= 0 || slot0 + count > p->max_frames) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_pipeline_submit: bad slot range");
    if (!p->host_ring) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_pipeline_submit: cbv_pipeline_host_ring() was never called");
    CBV_ENTER(ctx);
    if (!p->copy_stream) RC(ctx_worker_stream(ctx, &ctx->copy_stream, &p->copy_stream));
    // do not overwrite device slots a run that is still in flight reads: ANY such run, not only the last one
    retire_runs(p);
    {
        cbv_pipeline::RunRec* best = nullptr;
        for (auto& r : p->runs)
            if (r.live && ranges_overlap(slot0, count, r.s0, r.cnt) && (!best || r.seq > best->seq)) best = &r;
        if (best) CBV_HIP(ctx, hipStreamWaitEvent(p->copy_stream, best->one_event ? best->scan_ev : best->lanes_ev, 0));
    }
    CBV_HIP(ctx, hipMemcpyAsync(p->frames + p->g.frame_stride * slot0, p->host_ring + p->g.frame_stride * slot0,
                                p->g.frame_stride * count, hipMemcpyHostToDevice, p->copy_stream));
    cbv_pipeline::CopyRec* rec = nullptr;
    for (auto& c : p->copies)
        if (!c.pending) {
            rec = &c;
            break;
        }
    if (!rec) {
        cbv_pipeline::CopyRec c{0, 0, nullptr, false};
        CBV_HIP(ctx, hipEventCreateWithFlags(&c.ev, hipEventDisableTiming));
        p->copies.push_back(c);
        rec = &p->copies.back();
    }
    rec->s0 = slot0;
    rec->cnt = count;
    rec->pending = true;
    CBV_HIP(ctx, hipEventRecord(rec->ev, p->copy_stream));
    return CBV_OK;
}

extern "C" int cbv_pipeline_wait_submitted(cbv_pipeline* p)
{
    if (!p) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    CBV_ENTER(ctx);
    if (p->copy_stream) CBV_HIP(ctx, hipStreamSynchronize(p->copy_stream));
    return CBV_OK;
}

extern "C" int cbv_pipeline_synth(cbv_pipeline* p, int slot0, int count, const uint64_t* seeds, const double* Hinv9,
                                  const uint8_t* boards, const cbv_scene* scene)
{
    if (!p || !seeds || !Hinv9 || !boards || !scene || slot0 < 0 || count <= 0 || slot0 + count > p->max_frames) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    size_t o_seeds = 0, o_h = (size_t)count * 8, o_b = o_h + 72, o_s = (o_b + (size_t)count * 64 + 15) & ~(size_t)15;
    size_t total = o_s + sizeof(cbv_scene);
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RC(dev_ensure(ctx, &p->d_synth, total));
    std::vector<u8> host(total, 0);
    memcpy(host.data() + o_seeds, seeds, (size_t)count * 8);
    memcpy(host.data() + o_h, Hinv9, 72);
    memcpy(host.data() + o_b, boards, (size_t)count * 64);
    memcpy(host.data() + o_s, scene, sizeof(cbv_scene));
    CBV_HIP(ctx, hipMemcpy(p->d_synth.p, host.data(), total, hipMemcpyHostToDevice));
    u8* base = (u8*)p->d_synth.p;
    RC(launch_synth(ctx, p->frames + p->g.frame_stride * slot0, p->g, (const u64*)(base + o_seeds), (const double*)(base + o_h),
                    base + o_b, (const cbv_scene*)(base + o_s), count));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

// second half of cbv_pipeline_run: join the lanes on the scan's stream, HoughCircles second pass, temporal scan, run record
static int pipeline_run_tail(cbv_pipeline* p, cbv_pipeline::RunRec* rec, int slot0, int count, bool inline_scan, const bool* lane_used,
                             hipStream_t main_stream)
{
    cbv_ctx* ctx = p->ctx;
    const cbv_pipeline_config& cfg = p->cfg;
    const int n = cfg.n_rois;
    if (!p->scan_stream) {
        RC(ctx_worker_stream(ctx, &ctx->scan_stream, &p->scan_stream));
        CBV_HIP(ctx, hipEventCreateWithFlags(&p->main_done, hipEventDisableTiming));
    }
    hipStream_t scan_on = inline_scan ? main_stream : p->scan_stream;
    if (!inline_scan) {
        CBV_HIP(ctx, hipEventRecord(p->main_done, main_stream));
        CBV_HIP(ctx, hipStreamWaitEvent(scan_on, p->main_done, 0));
    }
    // every forked lane is joined, in the inline case too (chunk = 1 puts the second frame of a two-frame run on lane 1)
    for (int l = 1; l < p->n_lanes; l++)
        if (lane_used[l]) {
            CBV_HIP(ctx, hipEventRecord(p->lane_done[l], p->lane_stream[l]));
            CBV_HIP(ctx, hipStreamWaitEvent(scan_on, p->lane_done[l], 0));
        }
    // every lane has read its frames: a later cbv_pipeline_submit may overwrite these slots after this event
    rec->one_event = inline_scan;
    if (!inline_scan) CBV_HIP(ctx, hipEventRecord(rec->lanes_ev, scan_on));
    ctx->stream = scan_on;
    struct Restore {
        cbv_ctx* c;
        hipStream_t s;
        ~Restore() { c->stream = s; }
    } restore{ctx, main_stream};
    if (cfg.use_hough) // squares whose first HoughCircles pass overflowed (normally none), before the scan reads the decisions
        RC(launch_hough_second(ctx, (const SquareDesc*)p->d_descs.p, n, (const u8*)p->d_gray.p + p->plane_total * slot0, p->plane_total,
                               p->hough_cfg, (cbv_hough_result*)p->d_hough.p + (size_t)CBV_MAX_SQUARES * slot0,
                               (u8*)p->d_dec.p + (size_t)CBV_MAX_SQUARES * slot0, (const u32*)rec->retry.p, n * count));
    ScanParams sp;
    sp.n = n;
    sp.history_size = cfg.history_size;
    sp.min_presence = cfg.min_presence;
    sp.change_threshold = cfg.change_threshold;
    sp.with_model = p->calibrated ? 1 : 0;
    sp.stable_table = 0;
    for (int len = 1; len <= 7 && len <= cfg.history_size; len++)
        for (int sum = 0; sum <= len; sum++)
            if ((double)sum / (double)len >= cfg.min_presence) sp.stable_table |= 1ull << (len * 8 + sum);
    sp.thr_is_int = (cfg.change_threshold == (double)(int)cfg.change_threshold && cfg.change_threshold >= 0 && cfg.change_threshold < 256) ? 1 : 0;
    sp.thr_int = (int)cfg.change_threshold;
    // A short run (the live-camera case) writes its records to the pinned mirror too: reading them back is then a wait and a
    // host copy instead of two more launches.  Not the long runs: their records would cross PCIe as thousands of 8-byte
    // writes inside the scan stream's critical path (512-frame steps: -0.5 % frames/s, alternating A/B runs); they are
    // fetched with one copy when asked for.
    ResultMirror mir;
    const bool mirrored = count <= 4;
    if (mirrored) {
        mir.records = (cbv_frame_result*)p->h_stage + slot0;
        mir.over_src = cfg.use_hough ? (const u32*)p->d_hough_over.p : nullptr;
        mir.over_dst = pipeline_over_word(p);
    }
    for (int t = 0; t < count; t++) p->slot_mirrored[(size_t)slot0 + t] = mirrored ? 1 : 0;
    // + NoiseHandler on the frames' visual_changes sets (game_session.py:165)
    RC(launch_scan(ctx, (const SquareDesc*)p->d_descs.p, sp, (const u8*)p->d_gray.p + p->plane_total * slot0, p->plane_total,
                   (const u8*)p->d_dec.p + (size_t)CBV_MAX_SQUARES * slot0, (u8*)p->d_ref.p, (ScanState*)p->d_state.p,
                   (u8*)p->d_flags.p + (size_t)CBV_MAX_SQUARES * slot0, (cbv_frame_result*)p->d_results.p + slot0, count,
                   p->has_check ? (const u64*)p->d_check.p + slot0 : nullptr, (cbv_noise_state*)p->d_noise_state.p,
                   (cbv_noise_result*)p->d_noise.p + slot0, mir));
    CBV_HIP(ctx, hipEventRecord(rec->scan_ev, scan_on));
    rec->s0 = slot0;
    rec->cnt = count;
    rec->seq = ++p->run_seq;
    rec->live = true;
    return CBV_OK;
}

extern "C" int cbv_pipeline_run(cbv_pipeline* p, int slot0, int count)
{
    if (!p || !p->configured) return CBV_ERR_STATE;
    cbv_ctx* ctx = p->ctx;
    if (slot0 < 0 || count <= 0 || slot0 + count > p->max_frames) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_pipeline_run: bad slot range");
    CBV_ENTER(ctx);
    const cbv_pipeline_config& cfg = p->cfg;
    const int S = cfg.board_size, n = cfg.n_rois;
    // Lane 0 is the context's stream; lanes 1.. are worker streams forked from it and joined before
    // the temporal scan (which needs every frame's statistics, in order).
    hipStream_t main_stream = ctx->stream;
    // A run of one or two frames (the live-camera case) is latency, not throughput: its scan is a few microseconds, less
    // than the hop to the scan stream and back, so everything stays on the caller's stream, behind every run in flight
    // (the scans' state is sequential over runs).
    const bool inline_scan = count <= 2;
    const int chunks = (count + p->chunk - 1) / p->chunk;
    // Chunks go round the lanes, and the round continues from run to run and from pipeline to pipeline of this context
    // (ctx->lane_rr): K camera streams whose runs are one chunk each would otherwise all pile on lane 0 and lose the
    // overlap of the lanes.  Short (latency) runs start on the caller's stream.
    const int lane_base = inline_scan ? 0 : ctx->lane_rr % p->n_lanes;
    if (!inline_scan) ctx->lane_rr = (ctx->lane_rr + chunks) % (12 * 1024);
    bool lane_used[cbv_pipeline::MAX_LANES] = {false, false, false, false};
    for (int c = 0; c < chunks && c < p->n_lanes; c++) lane_used[(lane_base + c) % p->n_lanes] = true;
    if (inline_scan) {
        retire_runs(p);
        RC(join_scan(p));
    } else RC(join_slots(p, slot0, count)); // scans in flight that still read these slots' planes, however many runs back
    cbv_pipeline::RunRec* rec = nullptr; // the record (and second-pass list) of this run
    for (auto& r : p->runs)
        if (!r.live) {
            rec = &r;
            break;
        }
    if (!rec) {
        cbv_pipeline::RunRec r{0, 0, 0, nullptr, nullptr, false, false, DevBuf()};
        CBV_HIP(ctx, hipEventCreateWithFlags(&r.lanes_ev, hipEventDisableTiming));
        CBV_HIP(ctx, hipEventCreateWithFlags(&r.scan_ev, hipEventDisableTiming));
        p->runs.push_back(r);
        rec = &p->runs.back();
    }
    // the second-pass list's counter: zeroed before the lanes fork from this stream, or, when the run is ONE chunk, by that
    // chunk's k_warp (a memset is a launch of its own, ~13 us with its bubble in front of a 150 us chain)
    const bool retry_zero_in_warp = chunks == 1;
    if (cfg.use_hough) {
        RC(dev_ensure(ctx, &rec->retry, sizeof(u32) * (1 + (size_t)CBV_MAX_SQUARES * p->max_frames)));
        if (!retry_zero_in_warp) CBV_HIP(ctx, hipMemsetAsync(rec->retry.p, 0, sizeof(u32), main_stream));
    }
    for (auto& c : p->copies) // ingest copies of these slots must have landed
        if (c.pending && ranges_overlap(slot0, count, c.s0, c.cnt)) {
            CBV_HIP(ctx, hipStreamWaitEvent(main_stream, c.ev, 0));
            c.pending = false;
        }
    bool forked = false;
    for (int l = 1; l < p->n_lanes; l++) forked = forked || lane_used[l];
    if (forked) {
        CBV_HIP(ctx, hipEventRecord(p->start_ev, main_stream));
        for (int l = 1; l < p->n_lanes; l++)
            if (lane_used[l]) CBV_HIP(ctx, hipStreamWaitEvent(p->lane_stream[l], p->start_ev, 0));
    }
    int ci = 0, rc_all = CBV_OK;
    for (int s0 = slot0; s0 < slot0 + count && rc_all == CBV_OK; s0 += p->chunk, ci++) {
        const int lane = (lane_base + ci) % p->n_lanes;
        ctx->stream = lane == 0 ? main_stream : p->lane_stream[lane];
        SmallLayout SL;
        rc_all = small_layout(ctx, &p->lane_small[lane], cfg.enhance.tiles_x * cfg.enhance.tiles_y, p->chunk, &SL, cfg.enhance.tiles_x, cfg.enhance.tiles_y);
        if (rc_all) break;
        const int b = std::min(p->chunk, slot0 + count - s0);
        const u8* src = p->frames + p->g.frame_stride * s0;
        u8* res = nullptr;
        NormSrc norm;
        rc_all = enhance_dev(ctx, src, p->A[lane], p->B[lane], p->g, &cfg.enhance, SL, b, !p->keep_enhanced, &res, &norm,
                             p->use_region ? &p->region : nullptr, p->C[lane]);
        if (rc_all) break;
        u8* wdst = p->warped + p->warped_stride * s0;
        u32* work = cfg.use_hough ? (u32*)p->lane_work[lane].p : nullptr; // worklist counter: zeroed by k_warp
        u32* retry0 = cfg.use_hough && retry_zero_in_warp ? (u32*)rec->retry.p : nullptr;
        if (p->keep_enhanced) {
            if (hipMemcpyAsync(p->enhanced + p->g.frame_stride * s0, res, p->g.frame_stride * b, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
                rc_all = cbv_fail(ctx, CBV_ERR_HIP, "copy of the enhanced frames failed");
                break;
            }
            rc_all = launch_warp(ctx, res, p->g, p->Minv, S, S, cfg.rot180, wdst, S * 3, p->warped_stride, NormSrc(), b, work, retry0);
        } else {
            rc_all = launch_warp(ctx, res, p->g, p->Minv, S, S, cfg.rot180, wdst, S * 3, p->warped_stride, norm, b, work, retry0);
        }
        if (rc_all) break;
        u8* dec = (u8*)p->d_dec.p + (size_t)CBV_MAX_SQUARES * s0;
        cbv_hough_result* hres = cfg.use_hough ? (cbv_hough_result*)p->d_hough.p + (size_t)CBV_MAX_SQUARES * s0 : nullptr;
        rc_all = launch_squares_pre5_stats(ctx, wdst, p->warped_stride, (const SquareDesc*)p->d_descs.p, n,
                                           (u8*)p->d_gray.p + p->plane_total * s0, p->plane_total,
                                           p->calibrated ? (const float*)p->d_mean.p : nullptr, p->calibrated ? (const float*)p->d_var.p + p->plane_total : nullptr,
                                           (const u8*)p->d_masks.p, (float)cfg.z_threshold, (cbv_sq_stats*)p->d_stats.p + (size_t)n * s0, b,
                                           dec, cfg.use_hough, work, hres, p->max_px);
        if (rc_all) break;
        if (cfg.use_hough)
            rc_all = launch_hough(ctx, (const SquareDesc*)p->d_descs.p, n, (const u8*)p->d_gray.p + p->plane_total * s0, p->plane_total,
                                  p->hough_cfg, hres, dec, work, b, (u32*)rec->retry.p, s0 - slot0);
    }
    ctx->stream = main_stream;
    // A failure after lanes were forked: whatever they already enqueued on these slots and scratch buffers must not outlive
    // the call unordered (no RunRec goes live for a failed run), whether a lane's launch failed or the join / scan below did.
    auto drain = [&](int rc) {
        ctx->stream = main_stream;
        for (int l = 1; l < p->n_lanes; l++)
            if (lane_used[l]) (void)hipStreamSynchronize(p->lane_stream[l]);
        if (p->scan_stream) (void)hipStreamSynchronize(p->scan_stream);
        (void)hipStreamSynchronize(main_stream);
        return rc;
    };
    if (rc_all) return drain(rc_all);
    const int rc_tail = pipeline_run_tail(p, rec, slot0, count, inline_scan, lane_used, main_stream);
    return rc_tail == CBV_OK ? CBV_OK : drain(rc_tail);
}

extern "C" int cbv_pipeline_set_check_squares(cbv_pipeline* p, int slot0, int count, const uint64_t* roi_masks)
{
    if (!p || !p->configured || slot0 < 0 || count <= 0 || slot0 + count > p->max_frames) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    CBV_ENTER(ctx);
    RC(join_scan(p)); // the last run's scan may still read the masks
    if (roi_masks) {
        CBV_HIP(ctx, hipMemcpyAsync((u64*)p->d_check.p + slot0, roi_masks, sizeof(u64) * count, hipMemcpyHostToDevice, ctx->stream));
        CBV_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the caller's buffer may go away
        p->has_check = true;
    } else CBV_HIP(ctx, hipMemsetAsync((u64*)p->d_check.p + slot0, 0, sizeof(u64) * count, ctx->stream));
    return CBV_OK;
}

extern "C" int cbv_pipeline_results(cbv_pipeline* p, int slot0, int count, cbv_frame_result* out)
{
    if (!p || !out || slot0 < 0 || count <= 0 || slot0 + count > p->max_frames) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    if (!p->configured || !p->h_stage) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_pipeline_results: the pipeline is not configured");
    // short runs left their records in pinned host memory (ResultMirror): wait for the runs, copy on the host; the others
    // are fetched into the same place first (through pinned memory in any case: a copy into the caller's pageable buffer
    // would be staged by the runtime, one blocking copy at a time)
    const size_t bytes = sizeof(cbv_frame_result) * (size_t)count;
    u32* over_h = pipeline_over_word(p);
    bool have = true;
    for (int t = 0; t < count; t++) have = have && p->slot_mirrored[(size_t)slot0 + t];
    if (!have) {
        CBV_HIP(ctx, hipMemcpyAsync((cbv_frame_result*)p->h_stage + slot0, (cbv_frame_result*)p->d_results.p + slot0, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (p->cfg.use_hough && p->d_hough_over.p) CBV_HIP(ctx, hipMemcpyAsync(over_h, p->d_hough_over.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!have)
        for (int t = 0; t < count; t++) p->slot_mirrored[(size_t)slot0 + t] = 1;
    memcpy(out, (const cbv_frame_result*)p->h_stage + slot0, bytes);
    const u32 over = *over_h;
    if (over) {
        // A truncated candidate list may change has_piece: never hand that over as if it were HoughCircles' answer.  The
        // counter is cleared on read, so the error is reported ONCE, by the first results call after the runs it
        // happened in, and later frames are not poisoned; `out` is filled and valid except for the flagged squares.
        CBV_HIP(ctx, hipMemsetAsync(p->d_hough_over.p, 0, 4, ctx->stream));
        *over_h = 0; // (nothing is in flight: the next run's last kernel writes the word again)
        return cbv_fail(ctx, CBV_ERR_UNSUPPORTED, "HoughCircles: the candidate list overflowed even the second pass on %u square(s) since the "
                        "previous cbv_pipeline_results; those occupancy bits are not HoughCircles' (cbv_pipeline_hough flags name the squares)", over);
    }
    return CBV_OK;
}

extern "C" int cbv_pipeline_noise_results(cbv_pipeline* p, int slot0, int count, cbv_noise_result* out)
{
    if (!p || !out || !p->configured || slot0 < 0 || count <= 0 || slot0 + count > p->max_frames) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    CBV_HIP(ctx, hipMemcpyAsync(out, (cbv_noise_result*)p->d_noise.p + slot0, sizeof(cbv_noise_result) * count, hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

extern "C" int cbv_noise_run(cbv_ctx* ctx, const uint64_t* changes, int n, cbv_noise_state* state, cbv_noise_result* out)
{
    if (!ctx || !changes || !state || !out || n <= 0) return cbv_fail(ctx, CBV_ERR_ARG, "cbv_noise_run: bad arguments");
    CBV_ENTER(ctx);
    size_t b_in = ((size_t)n * 8 + 255) & ~(size_t)255, b_out = ((size_t)n * sizeof(cbv_noise_result) + 255) & ~(size_t)255;
    RC(dev_ensure(ctx, &ctx->c, b_in + b_out + 256));
    u8* base = (u8*)ctx->c.p;
    CBV_HIP(ctx, hipMemcpyAsync(base, changes, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    CBV_HIP(ctx, hipMemcpyAsync(base + b_in + b_out, state, sizeof(cbv_noise_state), hipMemcpyHostToDevice, ctx->stream));
    RC(launch_noise(ctx, (const u64*)base, 1, n, (cbv_noise_state*)(base + b_in + b_out), (cbv_noise_result*)(base + b_in)));
    CBV_HIP(ctx, hipMemcpyAsync(out, base + b_in, (size_t)n * sizeof(cbv_noise_result), hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipMemcpyAsync(state, base + b_in + b_out, sizeof(cbv_noise_state), hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

extern "C" int cbv_pipeline_download(cbv_pipeline* p, int which, int slot, uint8_t* out)
{
    if (!p || !out || slot < 0 || slot >= p->max_frames) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    const u8* src;
    size_t bytes;
    if (which == 0) {
        src = p->frames + p->g.frame_stride * slot;
        bytes = (size_t)p->w * p->h * 3;
    } else if (which == 1) {
        if (!p->enhanced) return cbv_fail(ctx, CBV_ERR_STATE, "enhanced frames are not kept (configure with keep_enhanced = 1)");
        src = p->enhanced + p->g.frame_stride * slot;
        bytes = (size_t)p->w * p->h * 3;
    } else if (which == 2) {
        if (!p->warped) return cbv_fail(ctx, CBV_ERR_STATE, "pipeline not configured");
        src = p->warped + p->warped_stride * slot;
        bytes = (size_t)p->cfg.board_size * p->cfg.board_size * 3;
    } else
        return cbv_fail(ctx, CBV_ERR_ARG, "bad buffer selector %d", which);
    CBV_HIP(ctx, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

extern "C" int cbv_pipeline_hough(cbv_pipeline* p, int slot, cbv_hough_result* out)
{
    if (!p || !out || slot < 0 || slot >= p->max_frames) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    if (!p->configured || !p->cfg.use_hough) return cbv_fail(ctx, CBV_ERR_STATE, "cbv_pipeline_hough: the HoughCircles stage is not configured");
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    CBV_HIP(ctx, hipMemcpyAsync(out, (const cbv_hough_result*)p->d_hough.p + (size_t)CBV_MAX_SQUARES * slot,
                                sizeof(cbv_hough_result) * CBV_MAX_SQUARES, hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}

extern "C" int cbv_pipeline_square_stats(cbv_pipeline* p, int slot, cbv_sq_stats* out)
{
    if (!p || !out || !p->configured || slot < 0 || slot >= p->max_frames) return CBV_ERR_ARG;
    cbv_ctx* ctx = p->ctx;
    CBV_ENTER(ctx);
    RC(join_scan(p)); // lanes and scan of the last run
    CBV_HIP(ctx, hipMemcpyAsync(out, (cbv_sq_stats*)p->d_stats.p + (size_t)p->cfg.n_rois * slot, sizeof(cbv_sq_stats) * p->cfg.n_rois,
                                hipMemcpyDeviceToHost, ctx->stream));
    CBV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CBV_OK;
}
