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


class SmartGridExtractor:
    def __init__(self, debug=False):
        self.grid_lines_x = None
        self.grid_lines_y = None
        self.debug = debug

    def refine_grid(self, img_warped):
        raise NotImplementedError("SmartGridExtractor.refine_grid is calibration-time code (Canny projections) and "
                                  "is outside the MI355X hot path; set grid_lines_x / grid_lines_y from calibration.json")

    def split_board(self, img_warped):
        if self.grid_lines_x is None or self.grid_lines_y is None:
            return GridExtractor().split_board(img_warped)
        return {(c, 7 - r): img_warped[y:y + h, x:x + w] for r, c, x, y, w, h in _cells_lines(self.grid_lines_x, self.grid_lines_y)}

    def roi_table(self, rows, cols):
        if self.grid_lines_x is None or self.grid_lines_y is None:
            return GridExtractor().roi_table(rows, cols)
        return list(_cells_lines(self.grid_lines_x, self.grid_lines_y))
