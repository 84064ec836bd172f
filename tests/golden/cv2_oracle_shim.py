"""A `cv2` stand-in for ONE purpose: letting the reference's own Python classes
run in the build container, where opencv-python is not installed, so that their
numpy-side arithmetic and control flow can be recorded as fixtures
(tests/golden/make_reference_runs.py).

Every function here forwards to the CPU oracle (oracle/cbv_oracle.c), i.e. to
this repository's restatement of the OpenCV algorithm.  The OpenCV-side numbers
in a fixture recorded through this module are therefore CIRCULAR (oracle in,
oracle out) and pin nothing about OpenCV; what such a fixture pins is everything
the reference does around those calls: call order and parameters, the float32
numpy section of apply_color_profile, EMA arithmetic with weak Python scalars,
z-score / percentage / threshold logic, the detect_all_pieces control flow
(forced / cached / delta / history / conditional refresh), dict shapes.

Only the calls the hot path makes are provided; anything
else raises AttributeError, exactly like the empty import gate used before.
Never imported by the product, by bench.py or by any test: only by the golden
generator, here, where /root/reference exists.
"""
import types

import numpy as np

from oracle import cbv_oracle as O

COLOR_BGR2GRAY, COLOR_BGR2HSV, COLOR_HSV2BGR, COLOR_BGR2LAB, COLOR_LAB2BGR = 6, 40, 54, 44, 56
NORM_MINMAX = 32
THRESH_BINARY, THRESH_OTSU = 0, 8
HOUGH_GRADIENT = 3
ROTATE_180 = 1
INTER_LINEAR = 1

CALLS = []  # (name, summary of the arguments) in call order: lets the generator assert the call sequence


def _log(name, **kw):
    CALLS.append((name, kw))


def _c(a):
    return np.ascontiguousarray(a)


def cvtColor(src, code):
    _log("cvtColor", code=code, shape=tuple(src.shape), dtype=str(src.dtype))
    src = _c(src)
    if src.dtype != np.uint8:
        raise TypeError("shim: cvtColor on %s" % src.dtype)
    fn = {COLOR_BGR2GRAY: O.bgr2gray, COLOR_BGR2HSV: O.bgr2hsv, COLOR_HSV2BGR: O.hsv2bgr,
          COLOR_BGR2LAB: O.bgr2lab, COLOR_LAB2BGR: O.lab2bgr}.get(code)
    if fn is None:
        raise AttributeError("shim: cvtColor code %r" % code)
    return fn(src)


def convertScaleAbs(src, alpha=1.0, beta=0.0):
    _log("convertScaleAbs", alpha=alpha, beta=beta)
    return O.convert_scale_abs(_c(src), alpha, beta)


def split(m):
    _log("split", dtype=str(m.dtype))
    return tuple(np.ascontiguousarray(m[..., i]) for i in range(m.shape[2]))


def merge(mv):
    mv = list(mv)
    _log("merge", dtypes=[str(x.dtype) for x in mv])
    if len({x.dtype for x in mv}) != 1:
        raise TypeError("shim: merge of mixed depths (cv2 raises too)")
    return np.stack(mv, axis=-1)


class _CLAHE:
    def __init__(self, clipLimit, tileGridSize):
        self.clipLimit, self.tileGridSize = float(clipLimit), (int(tileGridSize[0]), int(tileGridSize[1]))

    def apply(self, src):
        _log("CLAHE.apply", clipLimit=self.clipLimit, tileGridSize=self.tileGridSize)
        return O.clahe(_c(src), self.clipLimit, self.tileGridSize)

    def getClipLimit(self):
        return self.clipLimit

    def getTilesGridSize(self):
        return self.tileGridSize


def createCLAHE(clipLimit=40.0, tileGridSize=(8, 8)):
    return _CLAHE(clipLimit, tileGridSize)


def bilateralFilter(src, d, sigmaColor, sigmaSpace):
    _log("bilateralFilter", d=d, sigmaColor=sigmaColor, sigmaSpace=sigmaSpace)
    return O.bilateral(_c(src), d, sigmaColor, sigmaSpace)


def filter2D(src, ddepth, kernel):
    k = np.asarray(kernel)
    _log("filter2D", ddepth=ddepth, kernel_dtype=str(k.dtype), kernel=k.tolist())
    if ddepth != -1 or k.shape != (3, 3):
        raise AttributeError("shim: filter2D only as the reference calls it")
    return O.filter3x3(_c(src), k.astype(np.float32))


def normalize(src, dst, alpha=1.0, beta=0.0, norm_type=4):
    _log("normalize", alpha=alpha, beta=beta, norm_type=norm_type)
    if norm_type != NORM_MINMAX or alpha != 0 or beta != 255 or dst is not None:
        raise AttributeError("shim: normalize only as the reference calls it")
    return O.normalize_minmax(_c(src))


def GaussianBlur(src, ksize, sigmaX):
    _log("GaussianBlur", ksize=tuple(ksize), sigmaX=sigmaX, ndim=src.ndim)
    if sigmaX != 0 or ksize[0] != ksize[1] or src.ndim != 2:
        raise AttributeError("shim: GaussianBlur only as the reference's hot path calls it")
    return O.gaussian_blur(_c(src), int(ksize[0]))


def threshold(src, thresh, maxval, type):  # noqa: A002 (cv2's own parameter name)
    _log("threshold", thresh=thresh, maxval=maxval, type=type)
    if type != THRESH_BINARY + THRESH_OTSU or maxval != 255:
        raise AttributeError("shim: threshold only as the reference calls it")
    src = _c(src)
    t = O.otsu_from_hist(np.bincount(src.ravel(), minlength=256))
    return float(t), np.where(src > t, 255, 0).astype(np.uint8)


def absdiff(a, b):
    _log("absdiff")
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype != np.uint8 or b.dtype != np.uint8 or a.shape != b.shape:
        raise TypeError("shim: absdiff on uint8 images of one shape only")
    return np.abs(a.astype(np.int16) - b.astype(np.int16)).astype(np.uint8)


def HoughCircles(image, method, dp, minDist, param1=100, param2=100, minRadius=0, maxRadius=0):
    _log("HoughCircles", method=method, dp=dp, minDist=minDist, param1=param1, param2=param2,
         minRadius=minRadius, maxRadius=maxRadius)
    if method != HOUGH_GRADIENT:
        raise AttributeError("shim: HOUGH_GRADIENT only")
    res = O.hough_circles(_c(image), dp, minDist, param1, param2, minRadius, maxRadius)
    if not res:
        return None
    return np.array([[(x, y, r) for (x, y, r, _) in res]], dtype=np.float32)  # cv2's (1, N, 3) float32


def getPerspectiveTransform(src, dst):
    _log("getPerspectiveTransform", src_dtype=str(np.asarray(src).dtype))
    return O.get_perspective_transform(src, dst)


def warpPerspective(src, M, dsize):
    _log("warpPerspective", dsize=tuple(dsize))
    return O.warp_perspective(_c(src), M, dsize)


def rotate(src, code):
    _log("rotate", code=code)
    if code != ROTATE_180:
        raise AttributeError("shim: ROTATE_180 only")
    return O.rotate180(_c(src))


def Canny(image, threshold1, threshold2):
    _log("Canny", t1=threshold1, t2=threshold2)
    return O.canny(_c(image), threshold1, threshold2)


def as_module():
    m = types.ModuleType("cv2")
    for k, v in globals().items():
        if k[0] != "_" and k not in ("types", "np", "O", "as_module"):
            setattr(m, k, v)
    m.__doc__ = __doc__
    return m
