"""Python face of cbv_squares: device-resident per-square planes (current
gray, reference, mean, variance) for an ordered set of board squares."""
import ctypes as C

import numpy as np

from . import _native as N

GRAY, REF, MEAN, VAR = 0, 1, 2, 3


def _as_view_array(img):
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise ValueError("square images must be uint8 HxW or HxWx3, got %s %s" % (a.dtype, a.shape))
    ok = a.strides[-1] == 1 and (a.ndim == 2 or a.strides[1] == 3) and a.strides[0] >= a.shape[1] * (3 if a.ndim == 3 else 1)
    return a if ok else np.ascontiguousarray(a)


class SquareSet:
    def __init__(self, ctx=None):
        self.ctx = ctx or N.context()
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.cbv_squares_create(self.ctx.h, C.byref(h)))
        self.h = h
        self.keys = []     # position of each index
        self.index = {}
        self.shapes = []

    def close(self):
        if self.h:
            self.ctx.lib.cbv_squares_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, squares, blur_k, keys=None):
        """Preprocess `squares` ({pos: img}).  With `keys` the index order is
        fixed to that list and positions missing from `squares` keep their
        current gray.  Returns True when the geometry (keys/shapes) changed,
        i.e. all device state was reset."""
        order = list(squares.keys()) if keys is None else list(keys)
        arrs = [(_as_view_array(squares[k]) if k in squares else None) for k in order]
        shapes = [(a.shape[0], a.shape[1]) if a is not None else None for a in arrs]
        changed = order != self.keys or any(s is not None and s != t for s, t in zip(shapes, self.shapes))
        if changed and any(a is None for a in arrs):
            raise RuntimeError("squares missing while the geometry changes")
        views = (N.SquareView * len(order))()
        for i, a in enumerate(arrs):
            if a is None:
                views[i].data = None
                views[i].h, views[i].w = self.shapes[i]
                views[i].stride, views[i].cn = 0, 0
                continue
            views[i].data = a.ctypes.data
            views[i].h, views[i].w = a.shape[0], a.shape[1]
            views[i].stride = a.strides[0]
            views[i].cn = 3 if a.ndim == 3 else 1
        self.ctx.check(self.ctx.lib.cbv_squares_load(self.h, views, len(order), int(blur_k)))
        if changed:
            self.keys = order
            self.index = {k: i for i, k in enumerate(order)}
            self.shapes = [s for s in shapes]
        return changed

    def _select(self, positions):
        if positions is None:
            return None
        sel = np.zeros(len(self.keys), np.uint8)
        for p in positions:
            if p in self.index:
                sel[self.index[p]] = 1
        return sel

    def calibrate(self, initial_variance, positions=None):
        sel = self._select(positions)
        self.ctx.check(self.ctx.lib.cbv_squares_calibrate(self.h, float(initial_variance), N.ptr(sel) if sel is not None else None))

    def ema(self, alpha, positions=None):
        sel = self._select(positions)
        self.ctx.check(self.ctx.lib.cbv_squares_ema(self.h, float(alpha), N.ptr(sel) if sel is not None else None))

    def set_ref(self, positions=None):
        sel = self._select(positions)
        self.ctx.check(self.ctx.lib.cbv_squares_set_ref(self.h, N.ptr(sel) if sel is not None else None))

    def stats(self, use_ref=False, use_model=False, z_threshold=2.5):
        out = (N.SqStats * len(self.keys))()
        self.ctx.check(self.ctx.lib.cbv_squares_stats(self.h, 1 if use_ref else 0, 1 if use_model else 0, float(z_threshold), out))
        return out

    def hough(self, dp=1.2, param1=100, param2=25, min_radius_ratio=0.20, max_radius_ratio=0.55):
        """_detect_circle_unified (piece_detector.py:210-270) for every loaded square."""
        prm = N.HoughParams(float(dp), float(param1), float(param2), float(min_radius_ratio), float(max_radius_ratio))
        out = (N.HoughResult * len(self.keys))()
        self.ctx.check(self.ctx.lib.cbv_squares_hough(self.h, prm, out))
        return out

    def get(self, which, pos):
        i = self.index[pos]
        h, w = self.shapes[i]
        out = np.empty((h, w), np.uint8 if which in (GRAY, REF) else np.float32)
        self.ctx.check(self.ctx.lib.cbv_squares_get(self.h, which, i, N.ptr(out)))
        return out

    def set(self, which, pos, arr):
        i = self.index[pos]
        h, w = self.shapes[i]
        a = np.ascontiguousarray(arr, dtype=np.uint8 if which in (GRAY, REF) else np.float32)
        if a.shape != (h, w):
            raise ValueError("plane for %s must be %s, got %s" % (pos, (h, w), a.shape))
        self.ctx.check(self.ctx.lib.cbv_squares_set(self.h, which, i, N.ptr(a)))


class PlaneDict:
    """dict-like view of one device plane ({pos: ndarray}); arrays are fetched
    from / written to the GPU on access so the attribute keeps the reference's
    shape (`detector.means[pos]`) without shadow copies on the host."""

    def __init__(self, sqset, which):
        self._s = sqset
        self._which = which
        self._valid = set()

    def _mark(self, positions):
        self._valid.update(positions)

    def clear(self):
        self._valid.clear()

    def __contains__(self, pos):
        return pos in self._valid

    def __len__(self):
        return len(self._valid)

    def __iter__(self):
        return iter([k for k in self._s.keys if k in self._valid])

    def keys(self):
        return list(iter(self))

    def items(self):
        return [(k, self[k]) for k in self]

    def values(self):
        return [self[k] for k in self]

    def get(self, pos, default=None):
        return self[pos] if pos in self._valid else default

    def __getitem__(self, pos):
        if pos not in self._valid:
            raise KeyError(pos)
        return self._s.get(self._which, pos)

    def __setitem__(self, pos, arr):
        self._s.set(self._which, pos, arr)
        self._valid.add(pos)

    def __bool__(self):
        return bool(self._valid)
