"""ctypes loader for the CPU oracle (oracle/cbv_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcbv_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "cbv_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcbv_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_prepare_analysis.restype = C.c_int
        _lib.orc_otsu_from_hist.restype = C.c_int
        _lib.orc_get_perspective_transform.restype = C.c_int
        _lib.orc_invert3x3.restype = C.c_int
        _lib.orc_bilateral_tables.restype = C.c_int
    return _lib


def set_threads(n):
    """OpenMP team size for later oracle calls from this thread (0 = only query); returns the size in effect."""
    lib().orc_set_threads.restype = C.c_int
    return lib().orc_set_threads(int(n))


class Profile(C.Structure):
    _fields_ = [("hue_shift", C.c_double), ("sat_scale", C.c_double), ("val_scale", C.c_double),
                ("contrast", C.c_double), ("brightness", C.c_double), ("radical_mode", C.c_int),
                ("target_hue", C.c_double), ("hue_window", C.c_double), ("enabled", C.c_int)]

    @classmethod
    def from_dict(cls, d):
        """Same defaults as frame_enhancer.py:61-68; {} -> disabled (no-op)."""
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


class Scene(C.Structure):
    _fields_ = [("bg_lo", C.c_uint8), ("bg_span", C.c_uint8), ("light", C.c_uint8 * 3), ("dark", C.c_uint8 * 3),
                ("white", C.c_uint8 * 3), ("black", C.c_uint8 * 3), ("noise", C.c_uint8), ("pad", C.c_uint8 * 3),
                ("radius", C.c_double)]


class SqStats(C.Structure):
    _fields_ = [("n", C.c_uint32), ("sum", C.c_uint32), ("sumsq", C.c_uint32), ("sad_ref", C.c_uint32),
                ("center_sum", C.c_uint32), ("center_cnt", C.c_uint32), ("border_sum", C.c_uint32),
                ("border_cnt", C.c_uint32), ("ring_sum", C.c_uint32 * 4), ("ring_cnt", C.c_uint32 * 4),
                ("z_count", C.c_uint32), ("z_max", C.c_float)]


def _u8(a):
    a = np.asarray(a)
    assert a.dtype == np.uint8
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _img_args(a):
    """(ptr, w, h, stride) for a 2-D/3-D uint8 array whose inner dims are contiguous."""
    assert a.strides[-1] == 1
    if a.ndim == 3:
        assert a.strides[1] == a.shape[2]
    return _p(a), a.shape[1], a.shape[0], a.strides[0]


SHARPEN_KERNEL = np.array([[-1, -1, -1], [-1, 9, -1], [-1, -1, -1]], dtype=np.float32)


def convert_scale_abs(img, alpha, beta):
    img = _u8(img)
    out = np.empty(img.shape, np.uint8)
    cn = img.shape[2] if img.ndim == 3 else 1
    ptr, w, h, st = _img_args(img)
    lib().orc_convert_scale_abs(ptr, w, h, st, cn, C.c_double(alpha), C.c_double(beta), _p(out), out.strides[0])
    return out


def _unary3(fn, img):
    img = _u8(img)
    out = np.empty(img.shape, np.uint8)
    ptr, w, h, st = _img_args(img)
    fn(ptr, w, h, st, _p(out), out.strides[0])
    return out


def bgr2hsv(img):
    return _unary3(lib().orc_bgr2hsv, img)


def hsv2bgr(img):
    return _unary3(lib().orc_hsv2bgr, img)


def bgr2lab(img):
    return _unary3(lib().orc_bgr2lab, img)


def lab2bgr(img):
    return _unary3(lib().orc_lab2bgr, img)


def apply_color_profile(img, profile_dict):
    img = _u8(img)
    out = np.empty(img.shape, np.uint8)
    prof = Profile.from_dict(profile_dict)
    ptr, w, h, st = _img_args(img)
    lib().orc_apply_color_profile(ptr, w, h, st, C.byref(prof), _p(out), out.strides[0])
    return out


def clahe(gray, clip_limit=3.0, tiles=(8, 8), return_lut=False):
    gray = _u8(gray)
    out = np.empty(gray.shape, np.uint8)
    lut = np.empty((tiles[1] * tiles[0], 256), np.uint8)
    ptr, w, h, st = _img_args(gray)
    lib().orc_clahe(ptr, w, h, st, C.c_double(clip_limit), tiles[0], tiles[1], _p(out), out.strides[0], _p(lut))
    return (out, lut) if return_lut else out


def correct_lighting(img, clip_limit=3.0, tiles=(8, 8)):
    img = _u8(img)
    out = np.empty(img.shape, np.uint8)
    ptr, w, h, st = _img_args(img)
    lib().orc_correct_lighting(ptr, w, h, st, C.c_double(clip_limit), tiles[0], tiles[1], _p(out), out.strides[0])
    return out


def bilateral(img, d=9, sigma_color=75.0, sigma_space=75.0):
    img = _u8(img)
    out = np.empty(img.shape, np.uint8)
    ptr, w, h, st = _img_args(img)
    lib().orc_bilateral(ptr, w, h, st, d, C.c_double(sigma_color), C.c_double(sigma_space), _p(out), out.strides[0])
    return out


def filter3x3(img, kernel=SHARPEN_KERNEL):
    img = _u8(img)
    k = np.ascontiguousarray(kernel, dtype=np.float32)
    out = np.empty(img.shape, np.uint8)
    cn = img.shape[2] if img.ndim == 3 else 1
    ptr, w, h, st = _img_args(img)
    lib().orc_filter3x3(ptr, w, h, st, cn, _p(k), _p(out), out.strides[0])
    return out


def normalize_minmax(img):
    img = _u8(img)
    out = np.empty(img.shape, np.uint8)
    cn = img.shape[2] if img.ndim == 3 else 1
    ptr, w, h, st = _img_args(img)
    lib().orc_normalize_minmax(ptr, w, h, st, cn, _p(out), out.strides[0])
    return out


def bgr2gray(img):
    img = _u8(img)
    out = np.empty(img.shape[:2], np.uint8)
    ptr, w, h, st = _img_args(img)
    lib().orc_bgr2gray(ptr, w, h, st, _p(out), out.strides[0])
    return out


def gaussian_blur(gray, k=5):
    gray = _u8(gray)
    out = np.empty(gray.shape, np.uint8)
    ptr, w, h, st = _img_args(gray)
    lib().orc_gaussian_blur(ptr, w, h, st, k, _p(out), out.strides[0])
    return out


def otsu_from_hist(hist):
    hist = np.ascontiguousarray(hist, dtype=np.int32)
    return lib().orc_otsu_from_hist(_p(hist), int(hist.sum()))


def prepare_analysis(img):
    img = _u8(img)
    gray = np.empty(img.shape[:2], np.uint8)
    binary = np.empty(img.shape[:2], np.uint8)
    ptr, w, h, st = _img_args(img)
    t = lib().orc_prepare_analysis(ptr, w, h, st, _p(gray), _p(binary))
    return gray, binary, t


def process_pipeline(img, profile_dict=None, clip_limit=3.0, tiles=(8, 8), kernel=SHARPEN_KERNEL):
    img = _u8(img)
    out = np.empty(img.shape, np.uint8)
    prof = Profile.from_dict(profile_dict)
    k = np.ascontiguousarray(kernel, dtype=np.float32)
    ptr, w, h, st = _img_args(img)
    lib().orc_process_pipeline(ptr, w, h, st, C.byref(prof), C.c_double(clip_limit), tiles[0], tiles[1], _p(k),
                               _p(out), out.strides[0])
    return out


def get_perspective_transform(src_pts, dst_pts):
    s = np.ascontiguousarray(np.asarray(src_pts, dtype=np.float32).reshape(4, 2))
    d = np.ascontiguousarray(np.asarray(dst_pts, dtype=np.float32).reshape(4, 2))
    M = np.empty((3, 3), np.float64)
    lib().orc_get_perspective_transform(_p(s), _p(d), _p(M))
    return M


def invert3x3(M):
    M = np.ascontiguousarray(M, dtype=np.float64)
    D = np.empty((3, 3), np.float64)
    ok = lib().orc_invert3x3(_p(M), _p(D))
    return D if ok else np.zeros((3, 3))


def warp_perspective(img, M, dsize):
    img = _u8(img)
    M = np.ascontiguousarray(M, dtype=np.float64)
    dw, dh = dsize
    out = np.empty((dh, dw, 3), np.uint8)
    ptr, w, h, st = _img_args(img)
    lib().orc_warp_perspective(ptr, w, h, st, _p(M), dw, dh, _p(out), out.strides[0])
    return out


def warp_image(img, points, display_size=(1280, 720), margin=100):
    """board_detection.warp_image (board_detection.py:61-71)."""
    board_size = min(display_size) - margin
    pts2 = [[0, 0], [board_size, 0], [0, board_size], [board_size, board_size]]
    M = get_perspective_transform(points, pts2)
    return warp_perspective(img, M, (board_size, board_size)), M, board_size


def rotate180(img):
    img = _u8(img)
    out = np.empty(img.shape, np.uint8)
    cn = img.shape[2] if img.ndim == 3 else 1
    ptr, w, h, st = _img_args(img)
    lib().orc_rotate180(ptr, w, h, st, cn, _p(out), out.strides[0])
    return out


def square_preprocess(roi, blur_k=5):
    roi = _u8(roi)
    cn = roi.shape[2] if roi.ndim == 3 else 1
    out = np.empty(roi.shape[:2], np.uint8)
    ptr, w, h, st = _img_args(roi)
    lib().orc_square_preprocess(ptr, w, h, st, cn, blur_k, _p(out))
    return out


def piece_masks(w, h):
    m = np.empty((h, w), np.uint8)
    lib().orc_piece_masks(w, h, _p(m))
    return m


def square_stats(gray, ref=None, mean=None, var=None, z_thresh=2.5):
    gray = np.ascontiguousarray(_u8(gray))
    h, w = gray.shape
    st = SqStats()
    ref_c = np.ascontiguousarray(ref, dtype=np.uint8) if ref is not None else None
    mean_c = np.ascontiguousarray(mean, dtype=np.float32) if mean is not None else None
    var_c = np.ascontiguousarray(var, dtype=np.float32) if var is not None else None
    lib().orc_square_stats(_p(gray), w, h, _p(ref_c) if ref_c is not None else None,
                           _p(mean_c) if mean_c is not None else None, _p(var_c) if var_c is not None else None,
                           C.c_double(z_thresh), C.byref(st))
    return st


def ema_update(gray, alpha, mean, var):
    """In-place on float32 mean/var (change_detector.py:77-92)."""
    gray = np.ascontiguousarray(_u8(gray))
    assert mean.dtype == np.float32 and var.dtype == np.float32 and mean.flags.c_contiguous and var.flags.c_contiguous
    lib().orc_ema_update(_p(gray), gray.size, C.c_double(alpha), _p(mean), _p(var))


def synth_frame(seed, w, h, Hinv, board, scene):
    Hinv = np.ascontiguousarray(Hinv, dtype=np.float64)
    board = np.ascontiguousarray(board, dtype=np.uint8).reshape(64)
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_synth_frame(C.c_uint64(seed), w, h, _p(Hinv), _p(board), C.byref(scene), _p(out), out.strides[0])
    return out


class Circle(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("r", C.c_float), ("votes", C.c_int)]


def hough_circles(gray, dp=1.2, min_dist=25, param1=100, param2=25, min_radius=19, max_radius=42, return_edges=False):
    """cv2.HoughCircles(gray, HOUGH_GRADIENT, ...) restated (parity unpinned).  Returns [(x, y, r, votes)]."""
    gray = np.ascontiguousarray(_u8(gray))
    h, w = gray.shape
    out = (Circle * 64)()
    edges = np.empty((h, w), np.uint8)
    lib().orc_hough_circles.restype = C.c_int
    n = lib().orc_hough_circles(_p(gray), w, h, C.c_double(dp), C.c_double(min_dist), C.c_double(param1), C.c_double(param2),
                                int(min_radius), int(max_radius), out, 64, _p(edges))
    res = [(out[i].x, out[i].y, out[i].r, out[i].votes) for i in range(min(n, 64))]
    return (res, edges) if return_edges else res


def canny(gray, t1, t2):
    """cv2.Canny(gray, t1, t2) restated (aperture 3, L1 gradient); parity unpinned."""
    gray = np.ascontiguousarray(_u8(gray))
    h, w = gray.shape
    edges = np.empty((h, w), np.uint8)
    lib().orc_canny_u8(_p(gray), w, h, C.c_double(t1), C.c_double(t2), _p(edges))
    return edges
