"""ImageEnhancer on MI355X — drop-in for the reference's frame_enhancer.py.

Same constructor, attributes and methods as ImageEnhancerPython /
ImageEnhancerCython (frame_enhancer.py:23-181, src/cython/frame_enhancer_cython.pyx);
every method is one call into libcbv_hip.so.  Frames are numpy uint8 HxWx3 BGR
arrays (what cv2.VideoCapture yields); outputs are new arrays.
"""
import json
import os

import numpy as np

from . import _native as N


class _ClaheParams:
    """Stand-in for the cv2.CLAHE handle kept in `self.clahe` (frame_enhancer.py:36): carries clipLimit /
    tileGridSize with cv2's accessor names, and `apply(gray)` runs CLAHE on a single-channel uint8 image on
    the device (cbv_clahe_apply), as `self.clahe.apply(l)` does inside correct_lighting (frame_enhancer.py:114)."""

    def __init__(self, owner, clip_limit, tile_grid_size):
        self._owner = owner
        self.clipLimit = float(clip_limit)
        self.tileGridSize = (int(tile_grid_size[0]), int(tile_grid_size[1]))

    def apply(self, src):
        a = np.asarray(src)
        if a.dtype != np.uint8 or a.ndim != 2:
            raise ValueError("CLAHE.apply on the HIP path takes a single-channel uint8 image, got %s %s" % (a.dtype, a.shape))
        if a.strides[1] != 1:
            a = np.ascontiguousarray(a)
        out = np.empty(a.shape, np.uint8)
        c = self._owner._ctx
        tx, ty = self.tileGridSize
        c.check(c.lib.cbv_clahe_apply(c.h, N.ptr(a), a.shape[1], a.shape[0], a.strides[0], self.clipLimit, tx, ty, N.ptr(out), out.strides[0]))
        return out

    def getClipLimit(self):
        return self.clipLimit

    def setClipLimit(self, v):
        self.clipLimit = float(v)

    def getTilesGridSize(self):
        return self.tileGridSize

    def setTilesGridSize(self, v):
        self.tileGridSize = (int(v[0]), int(v[1]))


class ImageEnhancerHIP:
    def __init__(self, clahe_clip_limit=3.0, tile_grid_size=(8, 8)):
        self._ctx = N.context()
        self.clahe = _ClaheParams(self, clahe_clip_limit, tile_grid_size)
        self.sharpen_kernel = np.array([[-1, -1, -1], [-1, 9, -1], [-1, -1, -1]])
        self.profile = self.load_profile()

    def load_profile(self):
        """color_profile.json from the working directory, `{}` if absent or
        unreadable (frame_enhancer.py:46-54)."""
        try:
            if os.path.exists("color_profile.json"):
                with open("color_profile.json", "r") as f:
                    prof = json.load(f)
                print("Loaded color profile")
                return prof
        except Exception as e:  # same tolerance as the reference
            print(f"Error loading profile: {e}")
        return {}

    # -- helpers ------------------------------------------------------------
    def _kernel9(self):
        k = np.asarray(self.sharpen_kernel, dtype=np.float32)
        if k.shape != (3, 3):
            raise ValueError("sharpen_kernel must be 3x3 on the HIP path, got %s" % (k.shape,))
        return np.ascontiguousarray(k)

    def _params(self):
        p = N.EnhanceParams()
        p.profile = N.ColorProfile.from_dict(self.profile)
        p.clahe_clip_limit = self.clahe.clipLimit
        p.tiles_x, p.tiles_y = self.clahe.tileGridSize
        p.bilateral_d, p.sigma_color, p.sigma_space = 9, 75.0, 75.0
        k = self._kernel9().reshape(9)
        for i in range(9):
            p.sharpen_kernel[i] = float(k[i])
        return p

    # -- stages (frame_enhancer.py:56-159) -----------------------------------
    def apply_color_profile(self, frame):
        if not self.profile:
            return frame
        f = N.as_bgr(frame)
        out = np.empty(f.shape, np.uint8)
        prof = N.ColorProfile.from_dict(self.profile)
        lib, c = self._ctx.lib, self._ctx
        c.check(lib.cbv_apply_color_profile(c.h, N.ptr(f), f.shape[1], f.shape[0], f.strides[0], prof, N.ptr(out), out.strides[0]))
        return out

    def correct_lighting(self, frame):
        f = N.as_bgr(frame)
        out = np.empty(f.shape, np.uint8)
        lib, c = self._ctx.lib, self._ctx
        tx, ty = self.clahe.tileGridSize
        c.check(lib.cbv_correct_lighting(c.h, N.ptr(f), f.shape[1], f.shape[0], f.strides[0], self.clahe.clipLimit, tx, ty,
                                         N.ptr(out), out.strides[0]))
        return out

    def reduce_noise(self, frame):
        f = N.as_bgr(frame)
        out = np.empty(f.shape, np.uint8)
        lib, c = self._ctx.lib, self._ctx
        c.check(lib.cbv_reduce_noise(c.h, N.ptr(f), f.shape[1], f.shape[0], f.strides[0], 9, 75.0, 75.0, N.ptr(out), out.strides[0]))
        return out

    def sharpen(self, frame):
        f = N.as_bgr(frame)
        out = np.empty(f.shape, np.uint8)
        k = self._kernel9()
        lib, c = self._ctx.lib, self._ctx
        c.check(lib.cbv_sharpen(c.h, N.ptr(f), f.shape[1], f.shape[0], f.strides[0], N.ptr(k), N.ptr(out), out.strides[0]))
        return out

    def normalize_intensity(self, frame):
        f = N.as_bgr(frame)
        out = np.empty(f.shape, np.uint8)
        lib, c = self._ctx.lib, self._ctx
        c.check(lib.cbv_normalize_intensity(c.h, N.ptr(f), f.shape[1], f.shape[0], f.strides[0], N.ptr(out), out.strides[0]))
        return out

    def prepare_analysis(self, frame):
        f = N.as_bgr(frame)
        gray = np.empty(f.shape[:2], np.uint8)
        binary = np.empty(f.shape[:2], np.uint8)
        lib, c = self._ctx.lib, self._ctx
        c.check(lib.cbv_prepare_analysis(c.h, N.ptr(f), f.shape[1], f.shape[0], f.strides[0], N.ptr(gray), gray.strides[0],
                                         N.ptr(binary), binary.strides[0], None))
        return gray, binary

    def process_pipeline(self, frame):
        """apply_color_profile -> correct_lighting -> reduce_noise -> sharpen ->
        normalize_intensity (frame_enhancer.py:161-181) in one device round trip."""
        f = N.as_bgr(frame)
        out = np.empty(f.shape, np.uint8)
        p = self._params()
        lib, c = self._ctx.lib, self._ctx
        c.check(lib.cbv_process_pipeline(c.h, N.ptr(f), f.shape[1], f.shape[0], f.strides[0], p, N.ptr(out), out.strides[0]))
        return out


# the reference's final alias (frame_enhancer.py:184-190)
ImageEnhancer = ImageEnhancerHIP
