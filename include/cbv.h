/*
 * cbv.h — C-ABI of libcbv_hip.so: the MI355X (gfx950) implementation of the
 * per-frame digitisation path of hericmr/chessboard-vision.
 *
 * This is the drop-in boundary.  Every entry point replaces one cv2/numpy
 * call sequence of the reference (cited as reference file:line); the Python
 * classes in chessboard-vision_amd/ bind them with ctypes and mirror the
 * reference's class surface (ImageEnhancer, warp_image, ChangeDetector,
 * PieceDetector).  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every function returns CBV_OK (0) or a negative CBV_ERR_* code; the
 *     message is available from cbv_last_error().
 *   - images are uint8, HWC, BGR, row stride in bytes given explicitly
 *     (numpy frames from cv2.VideoCapture; views into them are fine).
 *   - "host" entry points copy in, run the kernels, copy out and synchronise;
 *     the library keeps no host pointer after returning.
 *   - "dev" entry points take device pointers, enqueue on the context's
 *     stream and do not synchronise.
 *   - one cbv_ctx per GPU is the intended use.  A ctx is ONE queue of work (shared scratch buffers, one current
 *     stream): every entry point that touches the GPU holds the context's lock for its whole call, so calls from
 *     several threads on one ctx (and on the cbv_squares / cbv_pipeline objects created on it) serialise; they never
 *     run concurrently.  Use one ctx per thread for concurrency.  Host buffers passed to a call must stay valid and
 *     unmodified until it returns; the pinned host ring (cbv_pipeline_host_ring) until cbv_pipeline_wait_submitted.
 *   - there is no CPU fallback: without a usable gfx950 device
 *     cbv_ctx_create() fails with CBV_ERR_NODEV.
 */
#ifndef CBV_H
#define CBV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define CBV_API __attribute__((visibility("default")))
#else
#define CBV_API
#endif

#define CBV_OK 0
#define CBV_ERR_ARG (-1)
#define CBV_ERR_HIP (-2)
#define CBV_ERR_NODEV (-3)
#define CBV_ERR_STATE (-4)
#define CBV_ERR_UNSUPPORTED (-5)

#define CBV_MAX_SQUARES 64
#define CBV_MAX_SQUARE_DIM 128

typedef struct cbv_ctx cbv_ctx;
typedef struct cbv_squares cbv_squares;
typedef struct cbv_pipeline cbv_pipeline;

/* color_profile.json as read by ImageEnhancer.load_profile
 * (frame_enhancer.py:46-54,61-68).  enabled = 0 is the `{}` profile. */
typedef struct {
    double hue_shift, sat_scale, val_scale, contrast, brightness;
    int32_t radical_mode;
    double target_hue, hue_window;
    int32_t enabled;
} cbv_color_profile;

/* Parameters of ImageEnhancer.process_pipeline (frame_enhancer.py:28-44,161-181). */
typedef struct {
    cbv_color_profile profile;
    double clahe_clip_limit;       /* 3.0 */
    int32_t tiles_x, tiles_y;      /* (8, 8) */
    int32_t bilateral_d;           /* 9 */
    double sigma_color, sigma_space; /* 75, 75 */
    float sharpen_kernel[9];       /* [[-1,-1,-1],[-1,9,-1],[-1,-1,-1]] */
} cbv_enhance_params;

typedef struct {
    int32_t x0, y0, w, h;
} cbv_roi;

/* One square handed over by the host: a (possibly strided) view, 1 or 3 channels. */
typedef struct {
    const uint8_t* data;
    int32_t w, h, stride, cn;
} cbv_square_view;

/* Integer statistics of one preprocessed (gray + Gaussian-blurred) square.
 * Everything the host decision chains of PieceDetector.detect_piece
 * (piece_detector.py:272-345) and ChangeDetector.detect_changes_detailed
 * (change_detector.py:105-167) need. */
typedef struct {
    uint32_t n;                 /* pixels */
    uint32_t sum, sumsq;        /* sum g, sum g^2  -> np.std (piece_detector.py:305) */
    uint32_t sad_ref;           /* sum |g - ref|   -> _has_changed (piece_detector.py:90-93) */
    uint32_t center_sum, center_cnt, border_sum, border_cnt; /* piece_detector.py:177-207 */
    uint32_t ring_sum[4], ring_cnt[4];                       /* piece_detector.py:141-175 */
    uint32_t z_count;           /* #(z > z_threshold)  (change_detector.py:136-137) */
    float z_max;                /* np.max(z)           (change_detector.py:160) */
} cbv_sq_stats;

/* Synthetic scene (bench / tests only; see chessboard-vision_amd/synth.py). */
typedef struct {
    uint8_t bg_lo, bg_span;
    uint8_t light[3], dark[3], white[3], black[3];
    uint8_t noise;
    uint8_t pad[3];
    double radius;
} cbv_scene;

/* ------------------------------------------------------------------ */
/* context                                                             */
/* ------------------------------------------------------------------ */
CBV_API int cbv_device_count(void);
CBV_API int cbv_ctx_create(int device_id, cbv_ctx** out);
CBV_API void cbv_ctx_destroy(cbv_ctx* ctx);
/* ctx may be NULL: last error of a failed cbv_ctx_create / host-only call. */
CBV_API const char* cbv_last_error(const cbv_ctx* ctx);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL
 * restores the context's own stream. */
CBV_API int cbv_ctx_set_stream(cbv_ctx* ctx, void* hip_stream);
CBV_API int cbv_ctx_synchronize(cbv_ctx* ctx);
CBV_API const char* cbv_device_name(const cbv_ctx* ctx);

/* Per-kernel timing with HIP events on the launch stream.  While enabled,
 * every launch of kernel `kid` (CBV_K_*) is bracketed by an event pair;
 * cbv_profile_read synchronises and returns the accumulated time. */
enum {
    CBV_K_COLOR_LAB_HIST = 0, CBV_K_CLAHE_LUT, CBV_K_CLAHE_APPLY, CBV_K_BILATERAL, CBV_K_SHARPEN,
    CBV_K_NORM_LUT, CBV_K_NORMALIZE, CBV_K_WARP, CBV_K_SQUARES, CBV_K_GRAY_BLUR, CBV_K_OTSU,
    CBV_K_THRESHOLD, CBV_K_SCAN, CBV_K_SYNTH, CBV_K_RESET, CBV_K_HOUGH, CBV_K_COUNT
};
CBV_API int cbv_profile_enable(cbv_ctx* ctx, int kid /* -1 = all, -2 = none */);
CBV_API int cbv_profile_read(cbv_ctx* ctx, int kid, double* total_ms, long long* launches);
CBV_API int cbv_profile_reset(cbv_ctx* ctx);
CBV_API const char* cbv_kernel_name(int kid);

/* ------------------------------------------------------------------ */
/* ImageEnhancer stages, host buffers (frame_enhancer.py)              */
/* ------------------------------------------------------------------ */
/* apply_color_profile, frame_enhancer.py:56-99 */
CBV_API int cbv_apply_color_profile(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride,
                            const cbv_color_profile* profile, uint8_t* out, int out_stride);
/* correct_lighting, frame_enhancer.py:101-120 */
CBV_API int cbv_correct_lighting(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, double clip_limit,
                         int tiles_x, int tiles_y, uint8_t* out, int out_stride);
/* cv2.CLAHE.apply on a single-channel 8-bit image: what `ImageEnhancer.clahe.apply(l)` does inside correct_lighting
 * (frame_enhancer.py:36,114), callable on its own because `clahe` is a public attribute of the class. */
CBV_API int cbv_clahe_apply(cbv_ctx* ctx, const uint8_t* gray, int w, int h, int stride, double clip_limit, int tiles_x,
                            int tiles_y, uint8_t* out, int out_stride);
/* reduce_noise, frame_enhancer.py:122-131 */
CBV_API int cbv_reduce_noise(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, int d, double sigma_color,
                     double sigma_space, uint8_t* out, int out_stride);
/* sharpen, frame_enhancer.py:133-138 (any 3x3 float kernel) */
CBV_API int cbv_sharpen(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, const float* kernel9, uint8_t* out,
                int out_stride);
/* normalize_intensity, frame_enhancer.py:140-146 */
CBV_API int cbv_normalize_intensity(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, uint8_t* out,
                            int out_stride);
/* prepare_analysis, frame_enhancer.py:148-159: returns unblurred gray and Otsu binary */
CBV_API int cbv_prepare_analysis(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, uint8_t* gray,
                         int gray_stride, uint8_t* binary, int binary_stride, int* otsu_threshold);
/* process_pipeline, frame_enhancer.py:161-181 */
CBV_API int cbv_process_pipeline(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride,
                         const cbv_enhance_params* params, uint8_t* out, int out_stride);

/* ------------------------------------------------------------------ */
/* warp (board_detection.py:61-71, game_session.py:124-126)            */
/* ------------------------------------------------------------------ */
/* cv2.getPerspectiveTransform: host-only, 4 (x,y) float32 pairs each. */
CBV_API int cbv_get_perspective_transform(const float* src8, const float* dst8, double* M9);
/* cv2.warpPerspective(img, M, (dw, dh)) INTER_LINEAR / BORDER_CONSTANT 0,
 * optionally followed by cv2.rotate(ROTATE_180). */
CBV_API int cbv_warp_perspective(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, const double* M9, int dw,
                         int dh, int rot180, uint8_t* out, int out_stride);

/* ------------------------------------------------------------------ */
/* per-square detector state (change_detector.py, piece_detector.py)   */
/* ------------------------------------------------------------------ */
CBV_API int cbv_squares_create(cbv_ctx* ctx, cbv_squares** out);
CBV_API void cbv_squares_destroy(cbv_squares* sq);
/* _preprocess / _preprocess_square of n squares (change_detector.py:49-56,
 * piece_detector.py:124-135): upload the views, BGR2GRAY when cn == 3,
 * GaussianBlur((k,k),0) per square.  Defines the squares' geometry; the
 * result becomes the "current" gray of each square on the device.  A view
 * whose data is NULL keeps that square's current gray (subset updates). */
CBV_API int cbv_squares_load(cbv_squares* sq, const cbv_square_view* views, int n, int blur_k);
/* Same, reading the ROIs from a device-resident image. */
CBV_API int cbv_squares_load_dev(cbv_squares* sq, const void* dev_img, int w, int h, int stride, int cn,
                         const cbv_roi* rois, int n, int blur_k);
/* calibrate (change_detector.py:36-47): mean = gray, var = initial_variance, for selected squares */
CBV_API int cbv_squares_calibrate(cbv_squares* sq, double initial_variance, const uint8_t* select /* n flags or NULL */);
/* update_all_references EMA (change_detector.py:73-92) */
CBV_API int cbv_squares_ema(cbv_squares* sq, double alpha, const uint8_t* select);
/* reference_squares[pos] = gray.copy() (piece_detector.py:95-97) */
CBV_API int cbv_squares_set_ref(cbv_squares* sq, const uint8_t* select);
/* statistics of the current gray of every square; use_ref / use_model say
 * whether sad_ref / z_* are wanted (they need set_ref / calibrate first). */
CBV_API int cbv_squares_stats(cbv_squares* sq, int use_ref, int use_model, double z_threshold, cbv_sq_stats* out);

/* PieceDetector._detect_circle_unified (piece_detector.py:210-270): cv2.HoughCircles(gray,
 * HOUGH_GRADIENT, dp, minDist = min_dim // 3, param1, param2, minRadius = int(min_dim * min_radius_ratio),
 * maxRadius = int(min_dim * max_radius_ratio)) on every loaded square's preprocessed gray, then the
 * circle nearest to (w // 2, h // 2) within 0.3 * min_dim.  The transform is restated from the
 * published OpenCV 4.x algorithm (parity unpinned). */
typedef struct {
    double dp;                 /* 1.2  piece_detector.py:234 */
    double param1, param2;     /* 100, 25  piece_detector.py:229-230 */
    double min_radius_ratio;   /* 0.20, or piece_detector_settings.json min_radius / 100 */
    double max_radius_ratio;   /* 0.55 */
} cbv_hough_params;
#define CBV_HOUGH_KEEP 6
#define CBV_HOUGH_OVERFLOW 1 /* more than 512 accumulator maxima, the rest were dropped */
#define CBV_HOUGH_SKIPPED 2  /* pipeline only: the statistics-based detectors already decided the square */
typedef struct {
    uint8_t found;             /* _detect_circle_unified's `found` */
    uint8_t kind;              /* 1 'hough', 2 'tower_top' */
    uint16_t n_circles;        /* circles HoughCircles returned for the square */
    float cx, cy, r;           /* the chosen circle (float32, before the reference's int()) */
    int32_t votes;
    uint32_t n_edges;          /* Canny edge pixels */
    uint16_t n_centres;        /* accumulator maxima above param2 */
    uint16_t flags;            /* CBV_HOUGH_* */
    float circles[CBV_HOUGH_KEEP][4]; /* first circles in HoughCircles' order: x, y, r, support */
} cbv_hough_result;
CBV_API int cbv_squares_hough(cbv_squares* sq, const cbv_hough_params* prm, cbv_hough_result* out);
/* download / upload per-square planes (tight w*h): which = 0 gray(u8) 1 ref(u8) 2 mean(f32) 3 var(f32) */
CBV_API int cbv_squares_get(cbv_squares* sq, int which, int index, void* out);
CBV_API int cbv_squares_set(cbv_squares* sq, int which, int index, const void* in);
CBV_API int cbv_squares_geometry(cbv_squares* sq, int index, int* w, int* h);

/* ------------------------------------------------------------------ */
/* one call per frame for the reference's own call pattern              */
/* (game_session.py:124-161, calibrate_sensitivity.py:142-157):         */
/* warp_image -> split_board -> detect_all_pieces / detect_changes      */
/* ------------------------------------------------------------------ */
/* The image all squares of a split_board() dict are views of (grid_extractor.py:46,153: img_warped[y:y+h, x:x+w]). */
typedef struct {
    const uint8_t* data;
    int32_t w, h, stride, cn;
} cbv_host_image;

/* _preprocess / _preprocess_square of the n ROIs of ONE host image: the rows of the image the ROIs cover are uploaded
 * with one copy at call time (no host pointer is kept, so pixels the caller drew on the board since warp_image are
 * seen), then as cbv_squares_load_dev.  Replaces cbv_squares_load's 64 packed view copies for split_board's case. */
CBV_API int cbv_squares_load_image(cbv_squares* sq, const cbv_host_image* img, const cbv_roi* rois, int n, int blur_k);
/* reference_squares[pos] = gray.copy() (piece_detector.py:95-97) for the squares of a 64-bit set (bit i = square i);
 * asynchronous: the set rides in the launch, nothing is read from host memory afterwards. */
CBV_API int cbv_squares_set_ref_mask(cbv_squares* sq, uint64_t mask);

#define CBV_METHOD_NONE 0
#define CBV_METHOD_HOUGH 1        /* 'hough'       confidence 0.9  (piece_detector.py:310-317) */
#define CBV_METHOD_TOWER_TOP 2    /* 'tower_top'   confidence 0.75 */
#define CBV_METHOD_CENTER_DIFF 3  /* 'center_diff' confidence min(1, diff / 80) (piece_detector.py:324-331) */
#define CBV_METHOD_SYMMETRY 4     /* 'symmetry'    confidence = score (piece_detector.py:337-343) */
/* detect_piece's result dict (piece_detector.py:289-299) + detect_all_pieces' per-square gate (piece_detector.py:367-395) */
typedef struct {
    uint8_t has_piece;        /* raw result of detect_piece (before the temporal smoothing, which stays with the caller) */
    uint8_t method;           /* CBV_METHOD_* */
    uint8_t changed;          /* has_changed_visual: no reference yet, or mean |gray - reference| > change_threshold */
    uint8_t should_process;   /* piece_detector.py:381-389 */
    uint8_t evaluated;        /* should_process or not cached: the fields of detect_piece below are fresh */
    uint8_t pad[3];
    int32_t cx, cy, radius;   /* result['center'], result['radius'] when has_piece */
    double confidence;
    double center_border_diff;
} cbv_piece_result;
/* detect_piece for one square from its statistics and its HoughCircles record (NULL = HoughCircles not run); host
 * only, no GPU.  CBV_ERR_UNSUPPORTED when the record carries CBV_HOUGH_OVERFLOW. */
CBV_API int cbv_decide_piece(const cbv_sq_stats* st, const cbv_hough_result* hg, int w, int h, double circle_threshold,
                             cbv_piece_result* out);

typedef struct {
    double change_threshold;   /* 25   piece_detector.py:50 */
    double circle_threshold;   /* 0.6  piece_detector.py:36 */
    cbv_hough_params hough;
    uint64_t has_ref;          /* bit i: square i is in reference_squares */
    uint64_t cached;           /* bit i: square i is in cached_results */
    uint64_t check;            /* squares_to_check (piece_detector.py:348), bit i */
    int32_t check_given;       /* squares_to_check is not None */
    int32_t use_delta;
} cbv_detect_params;
/* The device half of PieceDetector.detect_all_pieces (piece_detector.py:348-440) on the n squares of one host image, in
 * ONE call with one wait: upload, _preprocess_square, _has_changed against the device-resident references, the
 * should_process gate, HoughCircles on exactly the squares the reference would run it on (evaluated squares whose std
 * is >= 15), detect_piece's decision.  History, smoothing and the reference refresh (cbv_squares_set_ref_mask) stay
 * with the caller, which owns detection_history / cached_results like the reference's class does. */
CBV_API int cbv_squares_detect_all(cbv_squares* sq, const cbv_host_image* img, const cbv_roi* rois, int n,
                                   const cbv_detect_params* prm, cbv_piece_result* out /* n */);

typedef struct {
    double z_threshold;        /* change_detector.py:23 */
    uint64_t select;           /* squares to report on: focus_squares, or all, that are in `squares` and calibrated */
    double circle_threshold;   /* of the detector's own PieceDetector (change_detector.py:33) */
    cbv_hough_params hough;
} cbv_change_params;
typedef struct {
    uint8_t in_result;         /* pct_changed >= 5: the square is in detect_changes_detailed's dict */
    uint8_t intensity;         /* 1 LEVE, 2 PARCIAL, 3 TOTAL (change_detector.py:141-148) */
    uint8_t is_circular;       /* piece_detector.detect_piece(square)['has_piece'] (change_detector.py:152-154) */
    uint8_t pad;
    float z_max;               /* np.max(z_score) */
    uint32_t z_count, n;       /* pct_changed = z_count / n * 100 */
} cbv_change_result;
/* ChangeDetector.detect_changes_detailed (change_detector.py:105-167) on the n squares of one host image in one call:
 * upload, _preprocess with the detector's blur, z-score statistics against the device-resident model, and for the
 * squares that changed the circular test on the same pixels preprocessed the PieceDetector way (k = 5). */
CBV_API int cbv_squares_detect_changes(cbv_squares* sq, const cbv_host_image* img, const cbv_roi* rois, int n, int blur_k,
                                       const cbv_change_params* prm, cbv_change_result* out /* n */);
/* tests: fill partially uploaded staging buffers (cbv_warp_perspective's frame, the squares' image rows) with 0xA5
 * before the copy, so a read outside the uploaded part cannot go unnoticed */
CBV_API int cbv_debug_poison(cbv_ctx* ctx, int on);

/* cv2.Canny(img, threshold1, threshold2) with the default aperture 3 and L1 gradient, as
 * SmartGridExtractor.refine_grid (grid_extractor.py:66-121) uses it on the warped board at calibration time
 * (SURVEY §8 f3).  img: 1 or 3 channels (BGR is converted with BGR2GRAY first); edges: 0 / 255. */
CBV_API int cbv_canny(cbv_ctx* ctx, const uint8_t* img, int w, int h, int stride, int cn, double threshold1,
                      double threshold2, uint8_t* edges, int edges_stride);

/* board_detection.find_chessboard_corners (board_detection.py:4-46) up to, not including, reorder(): BGR2GRAY,
 * GaussianBlur((7,7), 1), Canny(30, 100), dilate(5x5, 3 iterations) on the GPU; findContours(RETR_EXTERNAL,
 * CHAIN_APPROX_NONE), contourArea > 100000, approxPolyDP(0.02 * arcLength) with four vertices, largest area on the
 * host.  Returns 1 and the polygon's four (x, y) vertices in pts8, 0 when no contour qualifies, negative on error.
 * dilated_out (optional) receives the dilated edge image.  Calibration-time code; parity unpinned. */
/* The host half of it alone (no GPU): external contours of a 0 / non-zero image -> the largest contour with
 * area > 100000 whose approxPolyDP(0.02 * arcLength) has four vertices.  n_contours (optional) = contours found. */
CBV_API int cbv_board_corners_from_edges(const uint8_t* edges, int w, int h, int stride, int32_t* pts8, int* n_contours);
/* Inspection helper: approxPolyDP (eps = eps_frac * arcLength) of the largest external contour; returns its vertex
 * count (the first `cap` are written), its area and its length in pixels. */
CBV_API int cbv_largest_contour_polygon(const uint8_t* edges, int w, int h, int stride, double eps_frac, int32_t* pts,
                                        int cap, double* area, int* contour_len);
CBV_API int cbv_find_chessboard_corners(cbv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, int32_t* pts8,
                                        uint8_t* dilated_out, int dilated_stride);

/* ------------------------------------------------------------------ */
/* device-resident batched pipeline: enhance -> warp -> 64-square detect */
/* over frames that stay in HBM (bench configs C2..C5)                  */
/* ------------------------------------------------------------------ */
typedef struct {
    cbv_enhance_params enhance;
    double M[9];                 /* getPerspectiveTransform result */
    int32_t board_size;          /* 620 */
    int32_t rot180;
    int32_t n_rois;              /* 64 */
    cbv_roi rois[CBV_MAX_SQUARES];   /* index = 8*row + col of the warped image (row 0 = rank 8) */
    int32_t history_size;        /* 5   piece_detector.py:40 */
    double min_presence;         /* 0.6 piece_detector.py:41 */
    double change_threshold;     /* 25  piece_detector.py:50 */
    int32_t chunk;               /* frames per kernel launch (0 = default) */
    int32_t lanes;               /* HIP streams the chunks are spread over (0 = default 2, max 4) */
    /* ChangeDetector stage on the same squares (change_detector.py:105-167), active once
     * cbv_pipeline_calibrate() has captured the background model: */
    double z_threshold;          /* 2.5  change_detector.py:23 */
    double initial_variance;     /* 100  change_detector.py:24 */
    int32_t keep_enhanced;       /* 1: materialise process_pipeline's output per frame (cbv_pipeline_download
                                    which = 1); 0: fold the final normalize into the warp gather */
    int32_t use_hough;           /* 1: detect_piece includes HoughCircles (piece_detector.py:308-317).  has_piece is
                                    an OR of three detectors, so the transform runs only on the squares the other
                                    two left undecided; 2: run it on every non-uniform square (inspection) */
    cbv_hough_params hough;
    int32_t enhance_region;      /* 1 (only with keep_enhanced = 0): CLAHE apply, bilateral and sharpen first run on the part
                                    of each frame the warp samples (+ stencil halos); the rest of the frame is processed
                                    only for frames whose region does not already hold both a 0 and a 255 after sharpen
                                    (normalize's global min / max are then 0 / 255 whatever the rest holds).  Every
                                    output is identical to whole-frame enhancement; 0 = always the whole frame */
} cbv_pipeline_config;

/* Per frame result of PieceDetector.detect_all_pieces(use_smoothing=True,
 * use_delta=True, squares_to_check=None) (piece_detector.py:348-440);
 * bit i = roi i. */
typedef struct {
    uint64_t raw_occupied;     /* raw has_piece per square (cached result when not processed) */
    uint64_t stable_occupied;  /* after 5-frame smoothing = results[pos]['has_piece'] */
    uint64_t visual_changes;   /* _has_changed */
    uint64_t processed;        /* should_process */
    /* ChangeDetector.detect_changes_detailed on the same frame (zero until calibrated): */
    uint64_t changed;          /* pct_changed >= 5: the square is in the result dict */
    uint64_t parcial;          /* 15 < pct_changed <= 75 */
    uint64_t total;            /* pct_changed > 75 */
    uint64_t circular;         /* detect_piece(current square)['has_piece'], evaluated fresh */
} cbv_frame_result;

/* NoiseHandler.process (noise_handler.py:49-213) evaluated on the device for every frame from its
 * visual_changes set.  `msg` selects the shape of the reference's data dict:
 *  0 waiting            1 hand_detected {changed_count}    2 detecting (from IDLE) {squares, lifted, stable, progress}
 *  3 noise_cleared      4 clearing {cooldown, progress}     5 stabilizing (NOISE) {changed_count}
 *  6 hand_active {changed_count}   7 detecting (from NOISE) {squares, stable}   8 interrupted_by_hand {changed_count}
 *  9 move_ready {squares, stable}  10 stabilizing (PENDING) {squares, stable, progress}
 * 11 stable_ready {squares, stable, progress}  12 counting {squares, lifted, stable, progress}  13 updated {...} */
typedef struct {
    uint8_t state;    /* returned state: 0 IDLE, 1 NOISE_ACTIVE, 2 MOVE_PENDING */
    uint8_t msg;
    uint8_t stable;   /* data["stable"] */
    int8_t lifted;    /* roi index of data["lifted"], -1 = None */
    uint16_t count;   /* changed_count, cooldown or stable_count (progress = count / 5 or / 12), by msg */
    uint16_t blocked; /* is_blocked() after the frame */
    uint64_t squares; /* data["squares"] */
} cbv_noise_result;

typedef struct {
    uint32_t state, stable_count, cooldown_count;
    int32_t lifted;   /* roi index + 1 of last_lifted_square, 0 = None */
    uint64_t pending;
} cbv_noise_state;

/* Run the state machine over `n` change sets (bit i = roi i); `state` is read and updated
 * (zero-initialised = a fresh NoiseHandler).  Host buffers. */
CBV_API int cbv_noise_run(cbv_ctx* ctx, const uint64_t* changes, int n, cbv_noise_state* state, cbv_noise_result* out);

CBV_API int cbv_pipeline_create(cbv_ctx* ctx, int w, int h, int max_frames, cbv_pipeline** out);
CBV_API void cbv_pipeline_destroy(cbv_pipeline* p);
CBV_API int cbv_pipeline_configure(cbv_pipeline* p, const cbv_pipeline_config* cfg);
/* device pointer of the input frame ring: [max_frames][h][w][3] uint8 */
CBV_API void* cbv_pipeline_frames_dev(cbv_pipeline* p);
CBV_API int cbv_pipeline_upload(cbv_pipeline* p, int slot, const uint8_t* bgr, int stride);
/* Ingest front end: a pinned host mirror of the frame ring ([max_frames][h][w][3], allocated on first call) that
 * the capture / decode side writes frames into, and an asynchronous copy of slots [slot0, slot0+count) to the
 * device ring on the pipeline's copy stream.  cbv_pipeline_run of those slots waits for their copy; a submit waits
 * for every run still in flight that reads the slots it would overwrite (however many runs back).  Submitting batch k+1 before running batch k
 * overlaps PCIe with compute. */
CBV_API uint8_t* cbv_pipeline_host_ring(cbv_pipeline* p);
CBV_API int cbv_pipeline_submit(cbv_pipeline* p, int slot0, int count);
/* Block until every submitted copy has left the pinned host ring (the copy stream only: runs stay in flight).
 * After it returns the capture side may overwrite any host-ring slot again. */
CBV_API int cbv_pipeline_wait_submitted(cbv_pipeline* p);
/* fill slots with synthetic frames generated on the device */
CBV_API int cbv_pipeline_synth(cbv_pipeline* p, int slot0, int count, const uint64_t* seeds, const double* Hinv9,
                       const uint8_t* boards /* count*64 */, const cbv_scene* scene);
/* reset the temporal detector state (reference squares, cache, history) */
CBV_API int cbv_pipeline_reset_state(cbv_pipeline* p);
/* ChangeDetector.calibrate (change_detector.py:36-47) from a slot that a previous cbv_pipeline_run
 * has processed: mean = its preprocessed squares, variance = cfg.initial_variance. */
CBV_API int cbv_pipeline_calibrate(cbv_pipeline* p, int slot);
/* detect_all_pieces' `squares_to_check` (piece_detector.py:348,381-389; game_session.py:130-152): per frame a set of
 * squares (bit = roi) that are processed even when unchanged and cached.  NULL clears the slots' sets (= None).  The
 * masks stay with the slots until changed. */
CBV_API int cbv_pipeline_set_check_squares(cbv_pipeline* p, int slot0, int count, const uint64_t* roi_masks);
/* What the session does after it accepted a move (game_session.py:219-223): PieceDetector.update_references with the
 * squares of an already processed slot (reference = that frame, cached results cleared, history kept) and, when
 * reset_noise, NoiseHandler.reset(). */
CBV_API int cbv_pipeline_update_references(cbv_pipeline* p, int slot, int reset_noise);
/* enqueue enhance -> warp -> detect for frames [slot0, slot0+count) in stream order; asynchronous */
CBV_API int cbv_pipeline_run(cbv_pipeline* p, int slot0, int count);
/* CBV_ERR_UNSUPPORTED (with `out` filled) when a HoughCircles candidate list overflowed even the second pass in a run
 * since the previous call: the counter is cleared on read, so later frames are not affected; cbv_pipeline_hough's flags
 * name the squares. */
CBV_API int cbv_pipeline_results(cbv_pipeline* p, int slot0, int count, cbv_frame_result* out);
/* NoiseHandler outputs of the same frames (fed by their visual_changes, like game_session.py:165) */
CBV_API int cbv_pipeline_noise_results(cbv_pipeline* p, int slot0, int count, cbv_noise_result* out);
/* download intermediates of one slot for parity checks: which = 0 input, 1 enhanced, 2 warped */
CBV_API int cbv_pipeline_download(cbv_pipeline* p, int which, int slot, uint8_t* out);
CBV_API int cbv_pipeline_square_stats(cbv_pipeline* p, int slot, cbv_sq_stats* out /* n_rois */);
/* HoughCircles outcome of one processed slot (cfg.use_hough): CBV_MAX_SQUARES entries, index = roi */
CBV_API int cbv_pipeline_hough(cbv_pipeline* p, int slot, cbv_hough_result* out);

#ifdef __cplusplus
}
#endif
#endif /* CBV_H */
