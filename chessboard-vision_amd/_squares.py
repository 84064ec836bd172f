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


def _image_of(parent):
    """cbv_host_image of a uint8 HxW / HxWx3 array with contiguous pixels (any row stride), or None."""
    if not isinstance(parent, np.ndarray) or parent.dtype != np.uint8 or parent.ndim not in (2, 3):
        return None
    cn = 1 if parent.ndim == 2 else parent.shape[2]
    if cn not in (1, 3) or parent.shape[0] <= 0 or parent.shape[1] <= 0:
        return None
    st = parent.strides
    if st[-1] != 1 or (parent.ndim == 3 and st[1] != 3) or st[0] < parent.shape[1] * cn:
        return None
    img = N.HostImage()
    img.data = parent.ctypes.data
    img.w, img.h, img.stride, img.cn = parent.shape[1], parent.shape[0], st[0], cn
    img._keep = parent
    return img


def plan_of(squares):
    """(cbv_host_image, SquareLayout) when every value of `squares` is a view into one uint8 image (split_board's case,
    grid_extractor.py:46,153), else None.  A SquareDict from this package's split_board carries the answer.  Any other
    mapping is analysed through the views themselves: they must share their owner (`.base`; numpy collapses chains of
    views to the array that owns the memory), their strides, and lie on whole pixels of the rows those strides define
    over the owner's block, which is then the image (e.g. a frame that is itself a reshape of a capture buffer)."""
    from .grid_extractor import SquareDict, SquareLayout
    if type(squares) is SquareDict and squares._parent is not None and len(squares) == len(squares._layout.keys):
        img = _image_of(squares._parent)
        if img is not None:
            return img, squares._layout
    if not squares or len(squares) > N.MAX_SQUARES:
        return None
    vals = list(squares.values())
    a0 = vals[0]
    if not isinstance(a0, np.ndarray) or a0.dtype != np.uint8 or a0.ndim not in (2, 3) or (a0.ndim == 3 and a0.shape[2] != 3):
        return None
    owner = a0.base if a0.base is not None else a0
    if not isinstance(owner, np.ndarray) or not (owner.flags.c_contiguous or owner.flags.f_contiguous):
        return None
    cn = 1 if a0.ndim == 2 else 3
    st = a0.strides
    if st[-1] != 1 or (a0.ndim == 3 and st[1] != 3) or st[0] < a0.shape[1] * cn:
        return None
    b0, row = owner.ctypes.data, st[0]
    rows_full = owner.nbytes // row  # whole rows of `row` bytes inside the owner's block
    rects = []
    for a in vals:
        if not isinstance(a, np.ndarray) or (a.base if a.base is not None else a) is not owner or a.dtype != np.uint8 or a.strides != st:
            return None
        if a.shape[0] <= 0 or a.shape[1] <= 0:
            return None
        y0, xb = divmod(a.ctypes.data - b0, row)
        x0, rem = divmod(xb, cn)
        if rem or xb + a.shape[1] * cn > row or y0 < 0 or y0 + a.shape[0] > rows_full:
            return None
        rects.append((x0, y0, a.shape[1], a.shape[0]))
    img = N.HostImage()
    img.data = b0
    img.w, img.h, img.stride, img.cn = row // cn, rows_full, row, cn
    img._keep = owner
    return img, SquareLayout(list(squares.keys()), rects)


class SquareSet:
    def __init__(self, ctx=None):
        self.ctx = ctx or N.context()
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.cbv_squares_create(self.ctx.h, C.byref(h)))
        self.h = h
        self.keys = []     # position of each index
        self.index = {}
        self.shapes = []
        self._layout = None  # SquareLayout of the last image-based load (identity check on the per-frame path)
        self._piece_out = np.zeros(N.MAX_SQUARES, N.record_dtype(N.PieceResult))
        self._change_out = np.zeros(N.MAX_SQUARES, N.record_dtype(N.ChangeResult))

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
        plan = plan_of(squares) if (keys is None or len(squares) == len(keys)) else None
        if plan is not None and (keys is None or list(keys) == plan[1].keys):
            return self.load_image(plan[0], plan[1], blur_k)
        self._layout = None
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

    def adopt(self, lay):
        """Take over the keys / shapes of a layout; True when they differ from the current ones (the device state of
        the squares is then void, exactly as after load() of a changed geometry)."""
        if lay is self._layout:
            return False
        changed = lay.keys != self.keys or lay.shapes != self.shapes
        if changed:
            self.keys, self.index, self.shapes = lay.keys, lay.index, lay.shapes
        self._layout = lay
        return changed

    def load_image(self, img, lay, blur_k):
        """load() for squares that are all views of one image: one upload of the rows they cover, at call time."""
        changed = self.adopt(lay)
        self.ctx.check(self.ctx.lib.cbv_squares_load_image(self.h, img, lay.rois, len(lay.keys), int(blur_k)))
        return changed

    def set_ref_mask(self, mask):
        self.ctx.check(self.ctx.lib.cbv_squares_set_ref_mask(self.h, int(mask)))

    def detect_all(self, img, lay, prm):
        """cbv_squares_detect_all; returns one tuple per square (has_piece, method, changed, should_process, evaluated,
        cx, cy, radius, confidence, center_border_diff) as Python scalars."""
        n = len(lay.keys)
        self.ctx.check(self.ctx.lib.cbv_squares_detect_all(self.h, img, lay.rois, n, prm, self._piece_out.ctypes.data))
        return self._piece_out[:n].tolist()

    def detect_changes(self, img, lay, blur_k, prm):
        """cbv_squares_detect_changes; one tuple per square (in_result, intensity, is_circular, z_max, z_count, n)."""
        n = len(lay.keys)
        self.ctx.check(self.ctx.lib.cbv_squares_detect_changes(self.h, img, lay.rois, n, int(blur_k), prm, self._change_out.ctypes.data))
        return self._change_out[:n].tolist()

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
