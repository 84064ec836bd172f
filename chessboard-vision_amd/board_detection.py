"""warp_image / reorder — drop-in for board_detection.py:49-71.

Corner detection and overlay drawing of the reference module are calibration/UI
code and are not part of this package."""
import ctypes as C

import numpy as np

from . import _native as N


def reorder(myPoints):
    """Order four corner points TL, TR, BL, BR by x+y and y-x extremes
    (board_detection.py:49-58).  Returns int32 (4,1,2)."""
    pts = np.asarray(myPoints).reshape((4, 2))
    out = np.zeros((4, 1, 2), np.int32)
    s = pts.sum(1)
    d = np.diff(pts, axis=1)
    out[0] = pts[np.argmin(s)]
    out[3] = pts[np.argmax(s)]
    out[1] = pts[np.argmin(d)]
    out[2] = pts[np.argmax(d)]
    return out


def get_perspective_transform(src_pts, dst_pts):
    """cv2.getPerspectiveTransform (8x8 LU in double); host-side, no GPU needed."""
    lib = N.load()
    s = np.ascontiguousarray(np.asarray(src_pts, dtype=np.float32).reshape(4, 2))
    d = np.ascontiguousarray(np.asarray(dst_pts, dtype=np.float32).reshape(4, 2))
    M = np.empty((3, 3), np.float64)
    rc = lib.cbv_get_perspective_transform(N.ptr(s), N.ptr(d), N.ptr(M))
    if rc != 0:
        raise RuntimeError(lib.cbv_last_error(None).decode())
    return M


def warp_perspective(img, M, dsize, rot180=False):
    c = N.context()
    f = N.as_bgr(img)
    M = np.ascontiguousarray(M, dtype=np.float64)
    dw, dh = int(dsize[0]), int(dsize[1])
    out = np.empty((dh, dw, 3), np.uint8)
    c.check(c.lib.cbv_warp_perspective(c.h, N.ptr(f), f.shape[1], f.shape[0], f.strides[0], N.ptr(M), dw, dh, 1 if rot180 else 0,
                                       N.ptr(out), out.strides[0]))
    return out


def warp_image(img, points, display_size=(1280, 720), margin=100):
    """Top-down view of the board (board_detection.py:61-71).
    Returns (warped, matrix, board_size)."""
    board_size = min(display_size) - margin
    pts1 = np.float32(points)
    pts2 = np.float32([[0, 0], [board_size, 0], [0, board_size], [board_size, board_size]])
    matrix = get_perspective_transform(pts1, pts2)
    warped = warp_perspective(img, matrix, (board_size, board_size))
    return warped, matrix, board_size
