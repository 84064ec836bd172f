"""ctypes binding of libcbv_hip.so (include/cbv.h).

There is no CPU fallback.  `load()` raises ImportError when the shared library
is missing; `context()` raises ImportError when no gfx950 device can be
opened.  ImportError is what the reference's own plugin selector catches to
fall back to its Python classes (frame_enhancer.py:13-21,
change_detector.py:12-19), so a host without an MI355X keeps working exactly
as it does today with a missing Cython extension.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcbv_hip.so")

CBV_OK = 0
MAX_SQUARES = 64


class ColorProfile(C.Structure):
    _fields_ = [("hue_shift", C.c_double), ("sat_scale", C.c_double), ("val_scale", C.c_double),
                ("contrast", C.c_double), ("brightness", C.c_double), ("radical_mode", C.c_int32),
                ("target_hue", C.c_double), ("hue_window", C.c_double), ("enabled", C.c_int32)]

    @classmethod
    def from_dict(cls, d):
        """Defaults of ImageEnhancer.apply_color_profile (frame_enhancer.py:61-68);
        a falsy profile is the no-op `{}` case (frame_enhancer.py:57-58)."""
        p = cls()
        p.enabled = 1 if d else 0
        d = d or {}
        p.hue_shift = d.get("hue_shift", 0)
        p.sat_scale = d.get("sat_scale", 1.0)
        p.val_scale = d.get("val_scale", 1.0)
        p.contrast = d.get("contrast", 1.0)
        p.brightness = d.get("brightness", 0)
        p.radical_mode = 1 if d.get("radical_mode", 0) else 0
        p.target_hue = d.get("target_hue", 0)
        p.hue_window = d.get("hue_window", 20)
        return p


class EnhanceParams(C.Structure):
    _fields_ = [("profile", ColorProfile), ("clahe_clip_limit", C.c_double), ("tiles_x", C.c_int32),
                ("tiles_y", C.c_int32), ("bilateral_d", C.c_int32), ("sigma_color", C.c_double),
                ("sigma_space", C.c_double), ("sharpen_kernel", C.c_float * 9)]


class Roi(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


class SquareView(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("stride", C.c_int32), ("cn", C.c_int32)]


class SqStats(C.Structure):
    _fields_ = [("n", C.c_uint32), ("sum", C.c_uint32), ("sumsq", C.c_uint32), ("sad_ref", C.c_uint32),
                ("center_sum", C.c_uint32), ("center_cnt", C.c_uint32), ("border_sum", C.c_uint32),
                ("border_cnt", C.c_uint32), ("ring_sum", C.c_uint32 * 4), ("ring_cnt", C.c_uint32 * 4),
                ("z_count", C.c_uint32), ("z_max", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("bg_lo", C.c_uint8), ("bg_span", C.c_uint8), ("light", C.c_uint8 * 3), ("dark", C.c_uint8 * 3),
                ("white", C.c_uint8 * 3), ("black", C.c_uint8 * 3), ("noise", C.c_uint8), ("pad", C.c_uint8 * 3),
                ("radius", C.c_double)]

    @classmethod
    def from_dict(cls, d):
        s = cls()
        s.bg_lo, s.bg_span, s.noise, s.radius = d["bg_lo"], d["bg_span"], d["noise"], d["radius"]
        for k in ("light", "dark", "white", "black"):
            for i in range(3):
                getattr(s, k)[i] = d[k][i]
        return s


class HoughParams(C.Structure):
    _fields_ = [("dp", C.c_double), ("param1", C.c_double), ("param2", C.c_double), ("min_radius_ratio", C.c_double),
                ("max_radius_ratio", C.c_double)]


HOUGH_KEEP = 6
HOUGH_OVERFLOW, HOUGH_SKIPPED = 1, 2


class HoughResult(C.Structure):
    _fields_ = [("found", C.c_uint8), ("kind", C.c_uint8), ("n_circles", C.c_uint16), ("cx", C.c_float), ("cy", C.c_float),
                ("r", C.c_float), ("votes", C.c_int32), ("n_edges", C.c_uint32), ("n_centres", C.c_uint16),
                ("flags", C.c_uint16), ("circles", (C.c_float * 4) * HOUGH_KEEP)]


class HostImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("stride", C.c_int32), ("cn", C.c_int32)]


METHOD_NAMES = (None, "hough", "tower_top", "center_diff", "symmetry")


class PieceResult(C.Structure):
    _fields_ = [("has_piece", C.c_uint8), ("method", C.c_uint8), ("changed", C.c_uint8), ("should_process", C.c_uint8),
                ("evaluated", C.c_uint8), ("pad", C.c_uint8 * 3), ("cx", C.c_int32), ("cy", C.c_int32), ("radius", C.c_int32),
                ("confidence", C.c_double), ("center_border_diff", C.c_double)]


class DetectParams(C.Structure):
    _fields_ = [("change_threshold", C.c_double), ("circle_threshold", C.c_double), ("hough", HoughParams),
                ("has_ref", C.c_uint64), ("cached", C.c_uint64), ("check", C.c_uint64), ("check_given", C.c_int32),
                ("use_delta", C.c_int32)]


class ChangeParams(C.Structure):
    _fields_ = [("z_threshold", C.c_double), ("select", C.c_uint64), ("circle_threshold", C.c_double), ("hough", HoughParams)]


class ChangeResult(C.Structure):
    _fields_ = [("in_result", C.c_uint8), ("intensity", C.c_uint8), ("is_circular", C.c_uint8), ("pad", C.c_uint8),
                ("z_max", C.c_float), ("z_count", C.c_uint32), ("n", C.c_uint32)]


def record_dtype(struct, skip=("pad",)):
    """numpy dtype with the layout of a ctypes Structure (padding fields left out): an array of it is filled by the
    library in place and `.tolist()` turns all records into Python scalars in one call."""
    names = [n for n, _ in struct._fields_ if n not in skip]
    return np.dtype({"names": names, "formats": [np.dtype(dict(struct._fields_)[n]) for n in names],
                     "offsets": [getattr(struct, n).offset for n in names], "itemsize": C.sizeof(struct)})


class PipelineConfig(C.Structure):
    _fields_ = [("enhance", EnhanceParams), ("M", C.c_double * 9), ("board_size", C.c_int32), ("rot180", C.c_int32),
                ("n_rois", C.c_int32), ("rois", Roi * MAX_SQUARES), ("history_size", C.c_int32),
                ("min_presence", C.c_double), ("change_threshold", C.c_double), ("chunk", C.c_int32),
                ("lanes", C.c_int32), ("z_threshold", C.c_double), ("initial_variance", C.c_double), ("keep_enhanced", C.c_int32),
                ("use_hough", C.c_int32), ("hough", HoughParams), ("enhance_region", C.c_int32)]


class FrameResult(C.Structure):
    _fields_ = [("raw_occupied", C.c_uint64), ("stable_occupied", C.c_uint64), ("visual_changes", C.c_uint64),
                ("processed", C.c_uint64), ("changed", C.c_uint64), ("parcial", C.c_uint64), ("total", C.c_uint64),
                ("circular", C.c_uint64)]


class NoiseResult(C.Structure):
    _fields_ = [("state", C.c_uint8), ("msg", C.c_uint8), ("stable", C.c_uint8), ("lifted", C.c_int8),
                ("count", C.c_uint16), ("blocked", C.c_uint16), ("squares", C.c_uint64)]


class NoiseDevState(C.Structure):
    _fields_ = [("state", C.c_uint32), ("stable_count", C.c_uint32), ("cooldown_count", C.c_uint32),
                ("lifted", C.c_int32), ("pending", C.c_uint64)]


KERNEL_IDS = ["COLOR_LAB_HIST", "CLAHE_LUT", "CLAHE_APPLY", "BILATERAL", "SHARPEN", "NORM_LUT", "NORMALIZE", "WARP",
              "SQUARES", "GRAY_BLUR", "OTSU", "THRESHOLD", "SCAN", "SYNTH", "RESET", "HOUGH"]
K = {name: i for i, name in enumerate(KERNEL_IDS)}

_lib = None


def load():
    """dlopen libcbv_hip.so and declare prototypes.  ImportError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("chessboard_vision_amd: %s not built (run `python -m chessboard_vision_amd.build`); "
                          "there is no CPU fallback" % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. ROCm runtime missing
        raise ImportError("chessboard_vision_amd: cannot load %s: %s" % (LIB_PATH, e))
    vp, i32, dbl, u8p = C.c_void_p, C.c_int, C.c_double, C.c_void_p
    P = C.POINTER
    proto = {
        "cbv_device_count": (i32, []),
        "cbv_ctx_create": (i32, [i32, P(vp)]),
        "cbv_ctx_destroy": (None, [vp]),
        "cbv_last_error": (C.c_char_p, [vp]),
        "cbv_ctx_set_stream": (i32, [vp, vp]),
        "cbv_ctx_synchronize": (i32, [vp]),
        "cbv_device_name": (C.c_char_p, [vp]),
        "cbv_profile_enable": (i32, [vp, i32]),
        "cbv_profile_read": (i32, [vp, i32, P(dbl), P(C.c_longlong)]),
        "cbv_profile_reset": (i32, [vp]),
        "cbv_kernel_name": (C.c_char_p, [i32]),
        "cbv_apply_color_profile": (i32, [vp, u8p, i32, i32, i32, P(ColorProfile), u8p, i32]),
        "cbv_correct_lighting": (i32, [vp, u8p, i32, i32, i32, dbl, i32, i32, u8p, i32]),
        "cbv_clahe_apply": (i32, [vp, u8p, i32, i32, i32, dbl, i32, i32, u8p, i32]),
        "cbv_reduce_noise": (i32, [vp, u8p, i32, i32, i32, i32, dbl, dbl, u8p, i32]),
        "cbv_sharpen": (i32, [vp, u8p, i32, i32, i32, vp, u8p, i32]),
        "cbv_normalize_intensity": (i32, [vp, u8p, i32, i32, i32, u8p, i32]),
        "cbv_prepare_analysis": (i32, [vp, u8p, i32, i32, i32, u8p, i32, u8p, i32, P(i32)]),
        "cbv_process_pipeline": (i32, [vp, u8p, i32, i32, i32, P(EnhanceParams), u8p, i32]),
        "cbv_get_perspective_transform": (i32, [vp, vp, vp]),
        "cbv_warp_perspective": (i32, [vp, u8p, i32, i32, i32, vp, i32, i32, i32, u8p, i32]),
        "cbv_squares_create": (i32, [vp, P(vp)]),
        "cbv_squares_destroy": (None, [vp]),
        "cbv_squares_load": (i32, [vp, P(SquareView), i32, i32]),
        "cbv_squares_load_dev": (i32, [vp, vp, i32, i32, i32, i32, P(Roi), i32, i32]),
        "cbv_squares_calibrate": (i32, [vp, dbl, vp]),
        "cbv_squares_ema": (i32, [vp, dbl, vp]),
        "cbv_squares_set_ref": (i32, [vp, vp]),
        "cbv_squares_stats": (i32, [vp, i32, i32, dbl, P(SqStats)]),
        "cbv_board_corners_from_edges": (i32, [u8p, i32, i32, i32, C.POINTER(C.c_int32), C.POINTER(i32)]),
        "cbv_largest_contour_polygon": (i32, [u8p, i32, i32, i32, dbl, C.POINTER(C.c_int32), i32, C.POINTER(dbl), C.POINTER(i32)]),
        "cbv_find_chessboard_corners": (i32, [vp, u8p, i32, i32, i32, C.POINTER(C.c_int32), u8p, i32]),
        "cbv_canny": (i32, [vp, u8p, i32, i32, i32, i32, dbl, dbl, u8p, i32]),
        "cbv_squares_hough": (i32, [vp, P(HoughParams), P(HoughResult)]),
        "cbv_squares_load_image": (i32, [vp, P(HostImage), P(Roi), i32, i32]),
        "cbv_squares_set_ref_mask": (i32, [vp, C.c_uint64]),
        "cbv_decide_piece": (i32, [P(SqStats), P(HoughResult), i32, i32, dbl, P(PieceResult)]),
        "cbv_squares_detect_all": (i32, [vp, P(HostImage), P(Roi), i32, P(DetectParams), vp]),
        "cbv_squares_detect_changes": (i32, [vp, P(HostImage), P(Roi), i32, i32, P(ChangeParams), vp]),
        "cbv_debug_poison": (i32, [vp, i32]),
        "cbv_squares_get": (i32, [vp, i32, i32, vp]),
        "cbv_squares_set": (i32, [vp, i32, i32, vp]),
        "cbv_squares_geometry": (i32, [vp, i32, P(i32), P(i32)]),
        "cbv_pipeline_create": (i32, [vp, i32, i32, i32, P(vp)]),
        "cbv_pipeline_destroy": (None, [vp]),
        "cbv_pipeline_configure": (i32, [vp, P(PipelineConfig)]),
        "cbv_pipeline_frames_dev": (vp, [vp]),
        "cbv_pipeline_upload": (i32, [vp, i32, u8p, i32]),
        "cbv_pipeline_synth": (i32, [vp, i32, i32, vp, vp, vp, P(Scene)]),
        "cbv_pipeline_reset_state": (i32, [vp]),
        "cbv_pipeline_calibrate": (i32, [vp, i32]),
        "cbv_pipeline_run": (i32, [vp, i32, i32]),
        "cbv_pipeline_results": (i32, [vp, i32, i32, P(FrameResult)]),
        "cbv_pipeline_download": (i32, [vp, i32, i32, u8p]),
        "cbv_pipeline_noise_results": (i32, [vp, i32, i32, P(NoiseResult)]),
        "cbv_noise_run": (i32, [vp, vp, i32, P(NoiseDevState), P(NoiseResult)]),
        "cbv_pipeline_host_ring": (C.c_void_p, [vp]),
        "cbv_pipeline_submit": (i32, [vp, i32, i32]),
        "cbv_pipeline_wait_submitted": (i32, [vp]),
        "cbv_pipeline_update_references": (i32, [vp, i32, i32]),
        "cbv_pipeline_set_check_squares": (i32, [vp, i32, i32, vp]),
        "cbv_pipeline_square_stats": (i32, [vp, i32, P(SqStats)]),
        "cbv_pipeline_hough": (i32, [vp, i32, P(HoughResult)]),
    }
    for name, (res, args) in proto.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTS = None  # filled lazily by tests from include/cbv.h


class Context:
    """One cbv_ctx (one GPU)."""

    def __init__(self, device_id=0):
        lib = load()
        h = C.c_void_p()
        rc = lib.cbv_ctx_create(device_id, C.byref(h))
        if rc != CBV_OK:
            msg = lib.cbv_last_error(None).decode()
            # no device / wrong device: surface as ImportError so the reference's
            # selector falls back (see module docstring)
            raise ImportError("chessboard_vision_amd: cannot open GPU %d: %s" % (device_id, msg))
        self.h = h
        self.lib = lib
        self.device_id = device_id

    def check(self, rc):
        if rc != CBV_OK:
            raise RuntimeError("libcbv_hip: %s (code %d)" % (self.lib.cbv_last_error(self.h).decode(), rc))

    @property
    def name(self):
        return self.lib.cbv_device_name(self.h).decode()

    def set_stream(self, stream_ptr):
        self.check(self.lib.cbv_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self.check(self.lib.cbv_ctx_synchronize(self.h))

    def profile_enable(self, kid):
        self.check(self.lib.cbv_profile_enable(self.h, kid))

    def profile_reset(self):
        self.check(self.lib.cbv_profile_reset(self.h))

    def profile_read(self, kid):
        ms, n = C.c_double(), C.c_longlong()
        self.check(self.lib.cbv_profile_read(self.h, kid, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self.h:
            self.lib.cbv_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def context(device_id=None):
    """Process-wide context per device.  Device defaults to $CBV_DEVICE or
    $LOCAL_RANK (one process per GPU) or 0."""
    if device_id is None:
        device_id = int(os.environ.get("CBV_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device_id not in _default_ctx:
        _default_ctx[device_id] = Context(device_id)
    return _default_ctx[device_id]


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def as_bgr(frame):
    """uint8 HxWx3 array with contiguous pixels (row stride free)."""
    a = np.asarray(frame)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected a uint8 HxWx3 BGR frame, got %s %s" % (a.dtype, a.shape))
    if a.strides[2] != 1 or a.strides[1] != 3 or a.strides[0] < a.shape[1] * 3:
        a = np.ascontiguousarray(a)
    return a
