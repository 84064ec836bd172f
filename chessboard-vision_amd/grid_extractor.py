"""Board -> 64 square views, keyed (file, rank) with a1 = (0, 0) and image row 0
= rank 8 — same contract as grid_extractor.py:8-58,123-163.  Pure numpy views
(no pixel is touched); `roi_table` gives the same geometry as an index table
for the device-resident pipeline."""
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


class GridExtractor:
    def split_board(self, img_warped):
        rows, cols = img_warped.shape[0], img_warped.shape[1]
        return {(c, 7 - r): img_warped[y:y + h, x:x + w] for r, c, x, y, w, h in _cells_linear(rows, cols)}

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
        return {(c, 7 - r): img_warped[y:y + h, x:x + w] for r, c, x, y, w, h in _cells_lines(self.grid_lines_x, self.grid_lines_y)}

    def roi_table(self, rows, cols):
        if self.grid_lines_x is None or self.grid_lines_y is None:
            return GridExtractor().roi_table(rows, cols)
        return list(_cells_lines(self.grid_lines_x, self.grid_lines_y))
