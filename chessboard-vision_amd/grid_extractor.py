"""Board -> 64 square views, keyed (file, rank) with a1 = (0, 0) and image row 0
= rank 8 — same contract as grid_extractor.py:8-58,123-163.  Pure numpy views
(no pixel is touched); `roi_table` gives the same geometry as an index table
for the device-resident pipeline.

split_board returns a SquareDict: a dict in every respect (same keys, same view
objects, same order as the reference's), which also remembers the image its
views were cut from and their rectangles, so that the detectors can take all
64 squares with ONE upload of that image at call time instead of 64 packed
view copies.  Any mutation of the dict drops that knowledge (the detectors
then look at the views themselves)."""
import numpy as np


def _cells_linear(rows, cols):
    sh, sw = rows // 8, cols // 8
    for r in range(8):
        for c in range(8):
            yield r, c, c * sw, r * sh, sw, sh


def _cells_lines(gx, gy):
    for r in range(8):
        for c in range(8):
            x0, x1, y0, y1 = gx[c], gx[c + 1], gy[r], gy[r + 1]
            if x0 >= x1 or y0 >= y1:
                continue  # degenerate cell is skipped (grid_extractor.py:149-150)
            yield r, c, x0, y0, x1 - x0, y1 - y0


class SquareLayout:
    """Rectangles of an ordered set of squares inside one parent image, in the forms the detectors use."""

    def __init__(self, keys, rects):
        from . import _native as N
        self.keys = list(keys)
        self.rects = [tuple(int(v) for v in r) for r in rects]           # (x0, y0, w, h)
        self.shapes = [(r[3], r[2]) for r in self.rects]                  # (h, w) like ndarray.shape[:2]
        self.index = {k: i for i, k in enumerate(self.keys)}
        self.bit_of = {k: 1 << i for i, k in enumerate(self.keys)}
        self.bits = [1 << i for i in range(len(self.keys))]
        self.all_mask = (1 << len(self.keys)) - 1
        self.rois = (N.Roi * max(1, len(self.keys)))()
        for i, (x0, y0, w, h) in enumerate(self.rects):
            self.rois[i].x0, self.rois[i].y0, self.rois[i].w, self.rois[i].h = x0, y0, w, h
        self.max_x = max((r[0] + r[2] for r in self.rects), default=0)
        self.max_y = max((r[1] + r[3] for r in self.rects), default=0)
        self.in_bounds = all(r[0] >= 0 and r[1] >= 0 and r[2] > 0 and r[3] > 0 for r in self.rects)

    def mask(self, positions):
        """64-bit set of the positions that are squares of this layout."""
        m = 0
        get = self.bit_of.get
        for p in positions:
            m |= get(p, 0)
        return m


class SquareDict(dict):
    """{(file, rank): view} as split_board returns it, plus `_parent` (the image) and `_layout` (SquareLayout)."""
    __slots__ = ("_parent", "_layout")

    def __init__(self, *a, **k):
        dict.__init__(self, *a, **k)
        self._parent = None
        self._layout = None

    def __reduce__(self):
        # copy / deepcopy / pickle give a plain dict: a copied view is no longer a view of the remembered image
        return (dict, (dict(self),))

    def _detach(self):
        self._parent = None
        self._layout = None

    def __setitem__(self, k, v):
        self._detach()
        dict.__setitem__(self, k, v)

    def __delitem__(self, k):
        self._detach()
        dict.__delitem__(self, k)

    def __ior__(self, other):
        self._detach()
        return dict.__ior__(self, other)

    def clear(self):
        self._detach()
        dict.clear(self)

    def pop(self, *a):
        self._detach()
        return dict.pop(self, *a)

    def popitem(self):
        self._detach()
        return dict.popitem(self)

    def setdefault(self, *a):
        self._detach()
        return dict.setdefault(self, *a)

    def update(self, *a, **k):
        self._detach()
        dict.update(self, *a, **k)

    def copy(self):
        return dict(self)


_layout_cache = {}


def _split(img_warped, cells_key, cells):
    """{(c, 7 - r): img_warped[y:y+h, x:x+w]} in the reference's order, with the parent / layout annotation."""
    lay = _layout_cache.get(cells_key)
    if lay is None:
        table = list(cells())
        lay = SquareLayout([(c, 7 - r) for r, c, _, _, _, _ in table], [(x, y, w, h) for _, _, x, y, w, h in table])
        if len(_layout_cache) > 64:
            _layout_cache.clear()
        _layout_cache[cells_key] = lay
    out = SquareDict()
    put = dict.__setitem__
    for k, (x, y, w, h) in zip(lay.keys, lay.rects):
        put(out, k, img_warped[y:y + h, x:x + w])
    # (slices past the image's edges are clamped by numpy, negative bounds wrap: then the rectangles are not the views)
    ok = (isinstance(img_warped, np.ndarray) and img_warped.dtype == np.uint8 and 0 < len(lay.keys) <= 64 and lay.in_bounds
          and lay.max_x <= img_warped.shape[1] and lay.max_y <= img_warped.shape[0])
    out._parent = img_warped if ok else None
    out._layout = lay if ok else None
    return out


class GridExtractor:
    def split_board(self, img_warped):
        rows, cols = img_warped.shape[0], img_warped.shape[1]
        return _split(img_warped, ("linear", rows, cols), lambda: _cells_linear(rows, cols))

    def roi_table(self, rows, cols):
        """[(row, col, x0, y0, w, h)] in row-major order."""
        return list(_cells_linear(rows, cols))


def canny(img, threshold1, threshold2):
    """cv2.Canny(img, threshold1, threshold2) on the GPU; BGR input is converted to gray first."""
    from . import _native as N
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise ValueError("canny expects an HxW or HxWx3 uint8 image")
    if a.strides[-1] != 1 or (a.ndim == 3 and a.strides[1] != 3):
        a = np.ascontiguousarray(a)
    ctx = N.context()
    out = np.empty(a.shape[:2], np.uint8)
    ctx.check(ctx.lib.cbv_canny(ctx.h, a.ctypes.data, a.shape[1], a.shape[0], a.strides[0], 3 if a.ndim == 3 else 1,
                                float(threshold1), float(threshold2), out.ctypes.data, out.strides[0]))
    return out


def _lines_from_projection(proj, length):
    """9 line positions: the borders plus, per inner line, the first maximum of `proj` inside its search window."""
    step = length / 8.0
    radius = int(step * 0.3)
    lines = [0]
    for i in range(1, 8):
        nominal = int(i * step)
        lo, hi = max(0, nominal - radius), min(length, nominal + radius)
        lines.append(lo + int(np.argmax(proj[lo:hi])) if hi > lo else nominal)
    lines.append(length)
    return lines


class SmartGridExtractor:
    def __init__(self, debug=False):
        self.grid_lines_x = None
        self.grid_lines_y = None
        self.debug = debug

    def refine_grid(self, img_warped):
        """grid_extractor.py:66-121: Canny(gray, 50, 150) on the warped board, edge counts per column / row, and
        for each of the 7 inner lines the strongest count within +-30 % of a square around its nominal place.
        Canny runs on the GPU (cbv_canny; restated from the published algorithm, parity unpinned)."""
        h, w = img_warped.shape[:2]
        edges = canny(img_warped, 50, 150)
        self.grid_lines_x = _lines_from_projection(edges.sum(axis=0, dtype=np.uint64), w)
        self.grid_lines_y = _lines_from_projection(edges.sum(axis=1, dtype=np.uint64), h)
        if self.debug:
            print(f"Refined X: {self.grid_lines_x}")
            print(f"Refined Y: {self.grid_lines_y}")
        return self.grid_lines_x, self.grid_lines_y

    def split_board(self, img_warped):
        if self.grid_lines_x is None or self.grid_lines_y is None:
            return GridExtractor().split_board(img_warped)
        gx, gy = tuple(int(v) for v in self.grid_lines_x), tuple(int(v) for v in self.grid_lines_y)
        return _split(img_warped, ("lines", gx, gy), lambda: _cells_lines(gx, gy))

    def roi_table(self, rows, cols):
        if self.grid_lines_x is None or self.grid_lines_y is None:
            return GridExtractor().roi_table(rows, cols)
        return list(_cells_lines(self.grid_lines_x, self.grid_lines_y))
