// Internal declarations shared by the HIP translation units of libcbv_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cbv.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------
// constant tables (device copies live in cbv_ctx::tabs)
// ---------------------------------------------------------------------------
enum {
    LAB_SHIFT = 12, GAMMA_SHIFT = 3, LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT,
    LAB_CBRT_TAB_SIZE_B = 256 * 3 / 2 * (1 << GAMMA_SHIFT),
    INV_GAMMA_SHIFT = 12, INV_GAMMA_TAB_SIZE = 1 << INV_GAMMA_SHIFT,
    LAB_BASE_SHIFT = 14, LAB_BASE = 1 << LAB_BASE_SHIFT
};

struct StaticTabs {
    int sdiv[256];                       // RGB2HSV_b saturation divisors
    int hdiv[256];                       // RGB2HSV_b hue divisors (hrange 180)
    u16 gamma[256];                      // sRGBGammaTab_b
    u16 cbrt[LAB_CBRT_TAB_SIZE_B];       // LabCbrtTab_b
    u16 inv_gamma[INV_GAMMA_TAB_SIZE];   // sRGBInvGammaTab_b
    u16 lab_yf[512];                     // LabToYF_b (y, ify) pairs
    int fwd[9];                          // RGB2Lab_b coefficients (BGR order)
    int inv[9];                          // Lab2RGBinteger coefficients
};

// apply_color_profile reduced to byte->byte tables (frame_enhancer.py:71-97)
struct ProfileTabs {
    u8 csa[256];      // convertScaleAbs(alpha=contrast, beta=brightness)
    u8 hmap[256];     // clip((h + hue_shift) % 180, 0, 179) -> uint8
    u8 vmap[256];     // clip(v * val_scale, 0, 255) -> uint8
    u8 smap[2][256];  // [radical mask][s]: clip(s * {1, 0.5, 2} * sat_scale, 0, 255) -> uint8
    u8 hmask[256];    // radical-mode window membership of h
    int enabled;
    int radical;      // radical_mode of the profile
    // HSV2RGB_b inputs per byte, evaluated on the host in float32 exactly as the device used to:
    u32 hsel[256];    // v_perm selector of (b, g, r) among {v, v(1-s), v(1-s f), v(1-s(1-f))} for hue byte h
    float hfr[256];   // f  = frac(hmap[h] * 6/180)
    float s_f[2][256]; // smap[m][s] * (1/255)
    float v_f[256];    // vmap[v] * (1/255)
};

#define CBV_BL_MAXCLS 10  // distinct dy^2 + dx^2 inside a disc of radius <= 4
struct BilateralTabs {
    float color_w[768];
    float space_w[128];
    signed char dy[128], dx[128];
    int maxk, radius;
    // w = space_w * color_w as ONE table lookup: taps with the same dy^2 + dx^2 share a space weight ("class"), and
    // folded[c][i] = space_w(class c) * color_w[i] is the very float product the filter would form (one IEEE multiply)
    float folded[CBV_BL_MAXCLS][768];
    int tap_off[9][16];  // [dy + radius][dx + radius]: class * 768 (word offset into folded), -1 outside the disc
    int ncls;
};

void build_static_tabs(StaticTabs* t);
void build_profile_tabs(const cbv_color_profile* p, ProfileTabs* t);
int build_bilateral_tabs(int d, double sigma_color, double sigma_space, BilateralTabs* t);
void build_gaussian_q8(int k, int* coef);
void build_piece_mask(int w, int h, u8* mask);
int host_invert3x3(const double* S, double* D);

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    unsigned long long tag = 0; // what the owner knows about the contents; 0 after every (re)allocation
};

struct ProfSlot {
    hipEvent_t a, b;
    int kid;
};

struct cbv_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    char devname[256] = {0};
    int num_cus = 256;

    StaticTabs* tabs = nullptr;          // device
    ProfileTabs* ptabs = nullptr;        // device, current profile
    cbv_color_profile ptabs_key;
    bool ptabs_valid = false;
    BilateralTabs* btabs = nullptr;      // device
    BilateralTabs btabs_host;
    int b_d = -1;
    double b_sc = 0, b_ss = 0;

    // scratch for the host-buffer entry points
    DevBuf in, a, b, c, small;

    // profiling
    int prof_kid = -2;
    std::vector<ProfSlot> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[CBV_K_COUNT] = {0};
    long long prof_n[CBV_K_COUNT] = {0};

    // One context = one queue of work: its scratch buffers and `stream` are shared by every object created on it, and
    // cbv_pipeline_run points `stream` at its lanes while it enqueues.  Entry points that touch the GPU hold this lock
    // for their whole call (CBV_ENTER), so calls from several threads on one context serialise instead of corrupting it.
    std::recursive_mutex mu;
    bool hough_lds_raised = false; // k_hough's dynamic-LDS limit was raised for this device

    // pinned staging for the small records the one-frame-per-call (class API) entry points move in either direction
    // (square descriptors, worklists, statistics, HoughCircles results): a copy from / to pinned memory is one DMA
    // (12 us for 16 KB against 26 us through the runtime's pageable path); guarded by `mu`, and every call that uses
    // it synchronises before it returns
    u8* h_stage = nullptr;
    size_t h_stage_cap = 0;
    int debug_poison = 0; // tests: fill partially uploaded staging buffers with 0xA5 first (cbv_debug_poison)

    // Worker streams of the pipelines created on this context (lanes 1.., temporal scan, ingest copies).  They belong to
    // the context, not to a pipeline: a HIP process has four hardware queues by default and streams beyond them share
    // a queue and serialise, so K camera streams on one GPU (K pipelines) must not bring K sets of streams.  All
    // pipelines of a context enqueue from under its lock, in order, so sharing changes no dependency.
    hipStream_t lane_streams[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t scan_stream = nullptr, copy_stream = nullptr;
    int lane_rr = 0; // next lane of the round the pipelines' chunks are dealt on (cbv_pipeline_run)
};
int ctx_worker_stream(cbv_ctx* ctx, hipStream_t* slot, hipStream_t* out);
int ctx_hstage(cbv_ctx* ctx, size_t bytes, u8** p);

extern thread_local std::string g_cbv_err;
int cbv_fail(cbv_ctx* ctx, int code, const char* fmt, ...);

#define CBV_HIP(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e__ = (call);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return cbv_fail(ctx, CBV_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                                    \
    } while (0)

// first statement of every entry point that uses the GPU through `ctx`
#define CBV_ENTER(ctx)                                   \
    std::lock_guard<std::recursive_mutex> lock__((ctx)->mu); \
    CBV_HIP(ctx, hipSetDevice((ctx)->device))

int dev_ensure(cbv_ctx* ctx, DevBuf* b, size_t bytes);
void dev_free(DevBuf* b);
int ctx_set_profile(cbv_ctx* ctx, const cbv_color_profile* p);
int ctx_set_bilateral(cbv_ctx* ctx, int d, double sc, double ss);

// RAII-less helpers to bracket a launch with profiling events
void prof_begin(cbv_ctx* ctx, int kid);
void prof_end(cbv_ctx* ctx, int kid);

// ---------------------------------------------------------------------------
// frame geometry for batched launches: `batch` frames, each frame_stride bytes
// apart, rows `stride` bytes apart.
// ---------------------------------------------------------------------------
struct Geom {
    int w, h;
    int stride;          // bytes per row
    size_t frame_stride; // bytes between consecutive frames of a batch
};

// ---------------------------------------------------------------------------
// Region-limited enhancement (cbv_pipeline_config::enhance_region).  The pipeline's only consumer of the enhanced frame
// is the warp, which samples the board quad; CLAHE apply, the bilateral and the sharpen + min/max pass therefore first
// run on the part of the frame the quad (+ their stencil halos) needs.  normalize needs the min / max of the WHOLE
// sharpened frame, but bytes cannot leave [0, 255]: if the region already holds a 0 and a 255 they ARE the global
// extremes and the rest of the frame cannot change any output.  A second pass over the complement of each kernel's
// region follows; its workgroups leave at once for frames whose region saturated (SatGate) and do the work otherwise,
// so results are identical to whole-frame enhancement in every case.
// ---------------------------------------------------------------------------
struct PxRect {
    int x0, y0, x1, y1; // pixels [x0, x1) x [y0, y1)
};

// up to four rectangles of a kernel's own tiles (one = a region, four = its complement), enumerated as ONE index space
struct TileSet {
    int n;
    int x0[4], y0[4], w[4];
    int cum[5]; // cum[k] = tiles in the rectangles before k; cum[n] = tiles per frame
};

// (constant indices and selects only: the set lives in kernel-argument SGPRs, a dynamic index would turn every call into
// scalar loads from the argument segment with an s_waitcnt behind them)
__host__ __device__ static inline void tileset_at(const TileSet& T, int r, int& tx, int& ty)
{
    int x0 = T.x0[0], y0 = T.y0[0], w = T.w[0], c = 0;
    if (T.n > 1 && r >= T.cum[1]) { x0 = T.x0[1]; y0 = T.y0[1]; w = T.w[1]; c = T.cum[1]; }
    if (T.n > 2 && r >= T.cum[2]) { x0 = T.x0[2]; y0 = T.y0[2]; w = T.w[2]; c = T.cum[2]; }
    if (T.n > 3 && r >= T.cum[3]) { x0 = T.x0[3]; y0 = T.y0[3]; w = T.w[3]; c = T.cum[3]; }
    const int l = r - c;
    const int row = l / w;
    ty = y0 + row;
    tx = x0 + (l - row * w);
}

static inline void tileset_add(TileSet& T, int x0, int y0, int x1, int y1)
{
    if (x1 <= x0 || y1 <= y0) return;
    T.x0[T.n] = x0;
    T.y0[T.n] = y0;
    T.w[T.n] = x1 - x0;
    T.cum[T.n + 1] = T.cum[T.n] + (x1 - x0) * (y1 - y0);
    T.n++;
}

// tiles [x0, x1) x [y0, y1) of a txn x tyn grid, or (invert) every other tile: strips above, below, left, right
static inline TileSet tileset_make(int txn, int tyn, int x0, int y0, int x1, int y1, bool invert)
{
    TileSet T;
    memset(&T, 0, sizeof(T));
    x0 = x0 < 0 ? 0 : x0;
    y0 = y0 < 0 ? 0 : y0;
    x1 = x1 > txn ? txn : x1;
    y1 = y1 > tyn ? tyn : y1;
    if (x1 <= x0 || y1 <= y0) x0 = x1 = y0 = y1 = 0; // empty region: its complement is everything
    if (!invert) tileset_add(T, x0, y0, x1, y1);
    else {
        tileset_add(T, 0, 0, txn, y0);
        tileset_add(T, 0, y1, txn, tyn);
        tileset_add(T, 0, y0, x0, y1);
        tileset_add(T, x1, y0, txn, y1);
    }
    if (T.n == 0) T.w[0] = 1; // no tiles: cum[0] = 0, never dereferenced with r < 0
    return T;
}

// frames whose min / max words (aux, after the region pass of sharpen) already read 0 / 255 are skipped
struct SatGate {
    const u32* mm;       // min word of frame 0 (max follows), or null = no gate
    size_t stride_words; // between frames
};
__device__ static inline bool sat_gate_closed(const SatGate& G, int frame)
{
    return G.mm && G.mm[(size_t)frame * G.stride_words] == 0u && G.mm[(size_t)frame * G.stride_words + 1] == 255u;
}

// k_sharpen_box's region: bytes [b0, b1) of the rows (multiples of 16: a lane's chunk) x rows [y0, y1) (multiples of the
// tile height); invert = everything else
struct ShRegion {
    int b0, b1, y0, y1, invert;
};

struct EnhanceRegion {
    PxRect px;   // what the consumer samples (clipped to the frame)
    int invert;  // 0: the region pass, 1: the complement pass
    SatGate gate;
};

struct ClaheGeom {
    int tiles_x, tiles_y, tw, th; // tile size of the (possibly extended) image
    int clip;                     // integer clip limit (0 = none)
    float lut_scale;
    int divisible;                // image dims divisible by the tile grid
};
ClaheGeom clahe_geom(int w, int h, double clip_limit, int tiles_x, int tiles_y);

// per-frame small device state of the enhancement chain
struct FrameAux {
    u32 hist[1];  // layout helper only
};
// layout of the per-frame aux block (u32 units)
//   [0 .. T*256)         CLAHE histograms
//   then 2               min, max of the sharpen output
//   then 256             otsu histogram
//   then 1               otsu threshold
__host__ __device__ static inline size_t aux_words(int tiles) { return (size_t)tiles * 256 + 2 + 256 + 2; }

// ---------------------------------------------------------------------------
// kernel launchers (device pointers; asynchronous on ctx->stream)
// ---------------------------------------------------------------------------
int launch_reset_aux(cbv_ctx* ctx, u32* aux, int tiles, int batch);
int launch_color_lab_hist(cbv_ctx* ctx, const u8* src, u8* lab, u32* aux, Geom g, ClaheGeom cg, int batch,
                          int do_profile, int do_lab);
// self_clean: the histograms are zeroed once read and [min, max] set to 255, 0, i.e. the kernel leaves the frame's words as
// k_reset_aux would, for the NEXT pass over the same buffer (k_sharpen's min / max come later in this pass)
int launch_clahe_lut(cbv_ctx* ctx, u32* aux, u8* luts, ClaheGeom cg, int batch, u32* packed, int self_clean = 0);
int launch_clahe_apply(cbv_ctx* ctx, const u8* lab, const u32* packed, u8* dst, Geom g, ClaheGeom cg, int batch, const EnhanceRegion* er = nullptr);
int launch_clahe_gray(cbv_ctx* ctx, const u8* src, int w, int h, int stride, ClaheGeom cg, u32* aux, u8* luts, u8* dst);
int launch_bilateral(cbv_ctx* ctx, const u8* src, u8* dst, Geom g, int batch, const EnhanceRegion* er = nullptr);
int launch_sharpen(cbv_ctx* ctx, const u8* src, u8* dst, u32* aux, int tiles, Geom g, const float* k9, int batch, const EnhanceRegion* er = nullptr);
// pixel rectangles the region pass of a kernel covers / must find complete in its input (tile rounding, halos); `out` of
// one stage dilated by its halo is the `need` of the stage before it
bool sharpen_region_ok(const float* k9);
PxRect sharpen_region_cover(Geom g, PxRect need);
PxRect bilateral_region_cover(cbv_ctx* ctx, Geom g, int batch, PxRect need);
PxRect clahe_region_cover(Geom g, PxRect need);
int launch_norm_lut(cbv_ctx* ctx, const u32* aux, int tiles, u8* norm_lut, int batch);
// Where a kernel takes cv2.normalize's byte map from: a table k_norm_lut built ([frame][256]), or the frames' [min, max]
// words themselves (`minmax` = frame 0's, `mm_stride` words apart), from which every workgroup works out the table it needs:
// a launch of a frame or two is launch latency, and that way the chain has one launch fewer.  Neither: no mapping.
struct NormSrc {
    const u8* lut = nullptr;
    const u32* minmax = nullptr;
    size_t mm_stride = 0;
};
static inline NormSrc norm_from_lut(const u8* lut)
{
    NormSrc n;
    n.lut = lut;
    return n;
}
static inline NormSrc norm_from_minmax(const u32* aux, int tiles)
{
    NormSrc n;
    n.minmax = aux + (size_t)tiles * 256;
    n.mm_stride = aux_words(tiles);
    return n;
}
int launch_normalize(cbv_ctx* ctx, const u8* src, u8* dst, NormSrc norm, Geom g, int batch);
int launch_warp(cbv_ctx* ctx, const u8* src, Geom g, const double* Minv9, int dw, int dh, int rot180, u8* dst,
                int dst_stride, size_t dst_frame_stride, NormSrc norm, int batch, u32* zero_word = nullptr, u32* zero_word2 = nullptr);
int launch_gray_blur_hist(cbv_ctx* ctx, const u8* src, u8* gray, u8* blur, u32* aux, int tiles, Geom g, int batch);
int launch_otsu(cbv_ctx* ctx, u32* aux, int tiles, int total, int batch);
int launch_threshold(cbv_ctx* ctx, const u8* blur, u8* binary, const u32* aux, int tiles, int w, int h, int batch);
int launch_synth(cbv_ctx* ctx, u8* dst, Geom g, const u64* seeds_dev, const double* hinv_dev, const u8* boards_dev,
                 const cbv_scene* scene_dev, int batch);

void build_gaussian_q8_sigma(int k, double sigma, int* coef);
int launch_gray_gauss(cbv_ctx* ctx, const u8* src, int w, int h, int stride, int cn, const int* coef_dev, int k, u8* dst);
int launch_dilate_rect(cbv_ctx* ctx, const u8* src, int w, int h, int r, u8* dst);
int largest_contour_polygon(const uint8_t* img, int w, int h, double eps_frac, int32_t* pts, int cap, double* area_out, int* contour_len);
int board_corners_from_edges(const uint8_t* dilated, int w, int h, int32_t pts[8], int* n_contours);
int launch_canny(cbv_ctx* ctx, const u8* src, int w, int h, int stride, int cn, int low, int high, u8* edges, DevBuf* scratch);

// squares
struct SquareDesc {
    int w, h;
    int src_off;   // byte offset of the ROI origin inside the source image / staging
    int stride;    // source row stride in bytes
    int cn;
    int plane_off; // element offset of this square's planes (gray/ref/mean/var)
    int mask_off;  // byte offset into the mask table
    int pad;
    // pixels of each region of the square's mask (centre disc, corners, four rings): they depend on the square's shape
    // only, so the host counts them once (square_region_counts) and the statistics kernels do not count them per frame
    u32 cnt[6];
    int pad2[2];
};
void square_region_counts(const u8* mask, int n, u32 cnt[6]);
// detect_all_pieces' per-square gate for the class API (one frame), evaluated by thread 0 of the statistics kernels:
// bit i of the sets = square i.  dflags == null: not a class-API launch.
struct DetectMasks {
    u64 has_ref = 0;     // pos in reference_squares
    u64 cached = 0;      // pos in cached_results
    u64 check = 0;       // pos in squares_to_check
    int check_given = 0; // squares_to_check is not None
    int use_delta = 1;
    double change_threshold = 25.0;
    u8* dflags = nullptr; // out, per square: 1 has_changed_visual, 2 should_process, 4 detect_piece is evaluated, 8 std >= 15
};
int launch_squares_preprocess(cbv_ctx* ctx, const u8* src, size_t src_frame_stride, const SquareDesc* descs, int n,
                              const int* coef_dev, int blur_k, u8* gray, size_t gray_frame_stride, int batch, int max_px = 0);
int launch_squares_stats(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, size_t gray_frame_stride,
                         const u8* ref, const float* mean, const float* var, const u8* masks, float z_thresh,
                         cbv_sq_stats* out, int batch, u8* decisions = nullptr, int want_hough = 0, u32* hough_work = nullptr,
                         cbv_hough_result* hough_out = nullptr, const DetectMasks* dm = nullptr);

// HoughCircles per square (k_hough.hip).  off_* / max_* are filled by launch_hough.
struct HoughCfg {
    float dp;
    int canny_thr, acc_thr;
    double min_ratio, max_ratio;
    int maxw, maxh; // largest square of the set
    int gs, mw, mag_bytes; // padded row strides of the gray/map planes (bytes) and the magnitude plane (u16)
    int off_map, off_mag, off_acc, off_centres, off_bins, off_order, max_bins;
    u32* overflow_count; // device counter of squares whose candidate list overflowed (CBV_HOUGH_OVERFLOW), or null
    int maxc;            // accumulator maxima / candidate circles kept per square (set by launch_hough)
    u32* retry;          // first pass: squares with more than `maxc` maxima are listed here (count, then frame << 8 | square)
                         // and done again by a second, rarely needed pass with room for every possible maximum
    int retry_frame_base; // added to the frame index of a listed square (the second pass may cover several first passes)
};
// `work` (may be null = every square of every frame): work[0] = number of items, work[1 + i] = frame << 8 | square,
// as k_squares_stats lists them; a found circle sets bit 0 of the square's `decisions` byte.
// `retry`: device list (count, then items) the first pass appends overflowing squares to (see HoughCfg::retry); the
// caller zeroes its count before the first pass that feeds it and runs launch_hough_second over it afterwards, with
// gray / out / decisions pointing at frame 0 of the list's frame numbering and `max_items` = its capacity.
int launch_hough(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, size_t gray_frame_stride, HoughCfg cfg,
                 cbv_hough_result* out, u8* decisions, const u32* work, int batch, u32* retry, int retry_frame_base);
int launch_hough_second(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, size_t gray_frame_stride, HoughCfg cfg,
                        cbv_hough_result* out, u8* decisions, const u32* retry, int max_items);
int launch_squares_pre5_stats(cbv_ctx* ctx, const u8* src, size_t src_frame_stride, const SquareDesc* descs, int n, u8* gray,
                              size_t gray_frame_stride, const float* mean, const float* var, const u8* masks, float z_thresh,
                              cbv_sq_stats* out, int batch, u8* decisions, int want_hough, u32* hough_work,
                              cbv_hough_result* hough_out, int max_px, const u8* ref = nullptr, const DetectMasks* dm = nullptr);
// (the statistics launchers' `var` argument is the SD plane: sqrt(var), written by these three next to the variance)
int launch_squares_calibrate(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, float* mean, float* var, float* sd,
                             float init_var, const u8* select);
int launch_squares_ema(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, float* mean, float* var, float* sd,
                       double alpha, const u8* select);
int launch_squares_refresh_sd(cbv_ctx* ctx, const SquareDesc* descs, const float* var, float* sd, int index);
int launch_squares_set_ref(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, u8* ref, const u8* select);
int launch_squares_set_ref_mask(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, u8* ref, u64 mask);

struct ScanParams {
    int n;               // squares
    int history_size;
    double min_presence;
    double change_threshold;
    int with_model; // z_count of the statistics is valid: classify LEVE / PARCIAL / TOTAL
    // _get_stable_detection's `sum(history) / len(history) >= min_presence` evaluated on the host in double for
    // every (len, sum): bit len * 8 + sum
    u64 stable_table;
    // mean(|diff|) > change_threshold as an exact integer test when the threshold is integral: sad > thr * n
    int thr_is_int;
    int thr_int;
};
struct ScanState {       // per square, device resident
    u32 has_ref, has_cache, cached_raw, hist_len, hist_bits;
};
int launch_scan_update_refs(cbv_ctx* ctx, const SquareDesc* descs, int n, const u8* gray, u8* ref, ScanState* state);
int launch_noise(cbv_ctx* ctx, const u64* changes, size_t stride_words, int count, cbv_noise_state* state, cbv_noise_result* out);
// Pinned host copy of a run's result records (`records` = the run's first slot there) and of HoughCircles' overflow
// counter, written by the run's last kernel beside the device copy: cbv_pipeline_results then waits and copies on the
// host instead of launching two device-to-host copies (~15 us of a 160 us frame).  All null: no mirror.
struct ResultMirror {
    cbv_frame_result* records = nullptr;
    const u32* over_src = nullptr;
    u32* over_dst = nullptr;
};
int launch_scan(cbv_ctx* ctx, const SquareDesc* descs, ScanParams sp, const u8* gray, size_t gray_frame_stride,
                const u8* decisions, u8* ref, ScanState* state, u8* flags, cbv_frame_result* results, int count,
                const u64* check = nullptr, cbv_noise_state* noise_state = nullptr, cbv_noise_result* noise_out = nullptr,
                ResultMirror mir = ResultMirror());
