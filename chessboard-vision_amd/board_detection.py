"""warp_image / reorder / find_chessboard_corners — drop-in for board_detection.py:4-71.

warp_image and reorder are the hot path's; find_chessboard_corners is the widening row f3 (pixel
stages on the GPU, contour following on the host inside the same library).  The overlay-drawing helpers of the
reference module (board_detection.py:84-146) are UI code and are not part of this package."""
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


def crop_inner_squares(img_warped, board_size, offset=0):
    """The warped board without an `offset`-pixel border, as a view, and the new board size (board_detection.py:74-82)."""
    cropped = img_warped[offset:board_size - offset, offset:board_size - offset]
    return cropped, board_size - 2 * offset


def corners_from_edges(edges):
    """Host half of find_chessboard_corners: (approx polygon int32 (4,1,2) or None, number of external contours)."""
    from . import _native as N
    import ctypes as C
    e = np.ascontiguousarray(np.asarray(edges, dtype=np.uint8))
    pts = (C.c_int32 * 8)()
    n = C.c_int(0)
    rc = N.load().cbv_board_corners_from_edges(e.ctypes.data, e.shape[1], e.shape[0], e.strides[0], pts, C.byref(n))
    if rc < 0:
        raise ValueError("corners_from_edges: bad arguments")
    return (np.array(pts, np.int32).reshape(4, 1, 2) if rc == 1 else None), n.value


def find_chessboard_corners(img, debug=False):
    """board_detection.py:4-28: the four corners (reordered TL, TR, BL, BR; int32 (4,1,2)) of the largest
    four-cornered contour of the dilated Canny edges, or an empty array.  Pixel stages on the GPU, contour
    following on the host (cbv_find_chessboard_corners); `debug=True` also returns the dilated edge image
    instead of showing a window."""
    from . import _native as N
    import ctypes as C
    a = N.as_bgr(img)
    ctx = N.context()
    pts = (C.c_int32 * 8)()
    dil = np.empty(a.shape[:2], np.uint8) if debug else None
    rc = ctx.lib.cbv_find_chessboard_corners(ctx.h, N.ptr(a), a.shape[1], a.shape[0], a.strides[0], pts,
                                             dil.ctypes.data if debug else None, dil.strides[0] if debug else 0)
    if rc < 0:
        ctx.check(rc)
    out = reorder(np.array(pts, np.int32).reshape(4, 1, 2)) if rc == 1 else np.array([])
    return (out, dil) if debug else out
