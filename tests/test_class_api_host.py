"""Host-side parts of the one-call class API (no GPU): detect_piece's decision in C (cbv_decide_piece) against the
reference's numpy arithmetic, and the SquareDict / SquareLayout bookkeeping that lets the detectors take all squares of
a split_board() dict with one upload."""
import ctypes as C

import numpy as np
import pytest

from chessboard_vision_amd import _native as N
from chessboard_vision_amd._squares import plan_of
from chessboard_vision_amd.grid_extractor import GridExtractor, SmartGridExtractor, SquareDict


def numpy_decide(st, w, h, hg, circle_threshold=0.6):
    """piece_detector.py:289-345 with numpy, as the reference evaluates it."""
    res = {"has_piece": False, "confidence": 0.0, "center": None, "radius": None, "method": None, "center_border_diff": 0}
    n, s, ss = int(st.n), int(st.sum), int(st.sumsq)
    if n * ss - s * s < 225 * n * n:  # np.std(gray) < 15
        return res
    if hg is not None and hg.found:
        kind = "tower_top" if hg.kind == 2 else "hough"
        res.update(has_piece=True, center=(int(hg.cx), int(hg.cy)), radius=int(hg.r), method=kind, confidence=0.9 if kind == "hough" else 0.75)
        return res
    with np.errstate(all="ignore"):
        cm = np.float64(st.center_sum) / np.float64(st.center_cnt) if st.center_cnt else np.float64("nan")
        bm = np.float64(st.border_sum) / np.float64(st.border_cnt) if st.border_cnt else np.float64("nan")
    diff = abs(cm - bm)
    res["center_border_diff"] = diff
    if diff > 40:
        res.update(has_piece=True, center=(w // 2, h // 2), radius=min(h, w) // 3, method="center_diff", confidence=min(1.0, diff / 80))
        return res
    rm = [np.float64(st.ring_sum[k]) / st.ring_cnt[k] for k in range(4) if st.ring_cnt[k] > 0]
    sym = 0.0 if len(rm) < 2 else min(1.0, np.var(rm) / 500)
    if sym > circle_threshold:
        res.update(has_piece=True, center=(w // 2, h // 2), radius=min(h, w) // 3, method="symmetry", confidence=sym)
    return res


def test_decide_piece_matches_numpy_arithmetic():
    lib = N.load()
    rng = np.random.default_rng(7)
    seen = set()
    for t in range(6000):
        w, h = int(rng.integers(8, 129)), int(rng.integers(8, 129))
        n = w * h
        st = N.SqStats()
        st.n = n
        mean = int(rng.integers(0, 256))
        spread = int(rng.integers(0, 90))
        st.sum = mean * n
        st.sumsq = min(2 ** 32 - 1, (mean * mean + spread * spread) * n)
        cc = int(rng.integers(0, n // 3 + 1)) if t % 17 else 0
        bc = int(rng.integers(0, n // 3 + 1)) if t % 19 else 0
        st.center_cnt, st.border_cnt = cc, bc
        st.center_sum, st.border_sum = int(rng.integers(0, 255 * cc + 1)), int(rng.integers(0, 255 * bc + 1))
        if t % 3 == 0 and cc and bc:  # near the diff > 40 edge and small differences
            st.border_sum = min(255 * bc, int(st.center_sum / cc * bc) + int(rng.integers(-50, 50)) * (bc // 8 + 1))
            st.border_sum = max(0, st.border_sum)
        base = int(rng.integers(0, 200))
        for k in range(4):
            st.ring_cnt[k] = int(rng.integers(0, n // 4 + 1)) if rng.integers(0, 8) else 0
            st.ring_sum[k] = int(min(255, max(0, base + rng.integers(-40, 40))) * st.ring_cnt[k] + rng.integers(0, st.ring_cnt[k] + 1))
        hg = None
        if t % 5 == 0:
            hg = N.HoughResult()
            hg.found = int(rng.integers(0, 2))
            hg.kind = int(rng.integers(1, 3))
            hg.cx, hg.cy, hg.r = float(rng.uniform(0, w)), float(rng.uniform(0, h)), float(rng.uniform(1, 60))
        out = N.PieceResult()
        assert lib.cbv_decide_piece(C.byref(st), C.byref(hg) if hg is not None else None, w, h, 0.6, C.byref(out)) == 0
        want = numpy_decide(st, w, h, hg)
        got_method = N.METHOD_NAMES[out.method]
        assert bool(out.has_piece) == want["has_piece"] and got_method == want["method"], t
        seen.add(got_method)
        if want["has_piece"]:
            assert (out.cx, out.cy) == want["center"] and out.radius == want["radius"], t
        assert out.confidence == want["confidence"], t  # bit for bit: same float64 operations in the same order
        d = want["center_border_diff"]
        assert out.center_border_diff == d or (np.isnan(d) and np.isnan(out.center_border_diff)), t
    assert seen == {None, "hough", "tower_top", "center_diff", "symmetry"}
    # a truncated HoughCircles record is refused, never decided on
    st = N.SqStats()
    st.n, st.sum, st.sumsq = 100, 100 * 100, 100 * (100 * 100 + 900)
    hg = N.HoughResult()
    hg.flags = N.HOUGH_OVERFLOW
    assert lib.cbv_decide_piece(C.byref(st), C.byref(hg), 10, 10, 0.6, C.byref(N.PieceResult())) == -5


def _is(img, arr, w=None, h=None):
    """the cbv_host_image describes `arr`'s memory"""
    return (img.data == arr.ctypes.data and img.stride == arr.strides[0] and img.w == (w or arr.shape[1]) and img.h == (h or arr.shape[0])
            and img.cn == (1 if arr.ndim == 2 else 3))


def test_split_board_dict_carries_its_parent_until_mutated():
    board = np.arange(620 * 620 * 3, dtype=np.uint32).astype(np.uint8).reshape(620, 620, 3).copy()
    for ge in (GridExtractor(), SmartGridExtractor()):
        sq = ge.split_board(board)
        assert isinstance(sq, dict) and type(sq) is SquareDict and len(sq) == 64
        assert list(sq.keys())[:3] == [(0, 7), (1, 7), (2, 7)] and sq[(0, 0)].base is board
        img, lay = plan_of(sq)
        assert _is(img, board) and lay.keys == list(sq.keys())
        for k, (x, y, w, h) in zip(lay.keys, lay.rects):
            assert sq[k].shape == (h, w, 3) and sq[k].ctypes.data == board[y:, x:].ctypes.data
        assert plan_of(ge.split_board(board))[1] is lay  # one layout per grid geometry, reused every frame
    smart = SmartGridExtractor()
    smart.grid_lines_x = [0, 79, 157, 234, 310, 386, 464, 541, 620]
    smart.grid_lines_y = [0, 80, 158, 235, 311, 388, 465, 542, 620]
    sq = smart.split_board(board)
    parent, lay = plan_of(sq)
    assert lay.rects[9] == (79, 80, 78, 78) and sq[(1, 6)].shape == (78, 78, 3)
    # mutation drops the annotation; the views are then analysed themselves and give the same rectangles
    sq2 = smart.split_board(board)
    sq2[(0, 0)] = sq2[(0, 0)]
    assert sq2._parent is None
    p2, lay2 = plan_of(sq2)
    assert _is(p2, board) and lay2.keys == lay.keys and lay2.rects == lay.rects
    plain = dict(sq)
    p3, lay3 = plan_of(plain)
    assert _is(p3, board) and lay3.rects == lay.rects
    # copies are plain dicts (a deep-copied view is not a view of the remembered image any more), a bare SquareDict is inert
    import copy
    import pickle
    for dup in (copy.copy(sq), copy.deepcopy(sq), pickle.loads(pickle.dumps(sq)), sq.copy()):
        assert type(dup) is dict and list(dup.keys()) == list(sq.keys())
    deep = copy.deepcopy(sq)
    deep[(3, 3)][:] = 7
    assert not (board == 7).all() and (plan_of(deep) is None or plan_of(deep)[0].data != board.ctypes.data)
    assert plan_of(SquareDict()) is None and SquareDict(a=1)._parent is None
    # a board that is itself a view (a reshaped capture buffer, a crop of a larger frame): rectangles in the owner's rows
    buf = np.zeros(700 * 640 * 3 + 5, np.uint8)
    frame = buf[:700 * 640 * 3].reshape(700, 640, 3)
    crop = frame[40:660, 10:630]
    sqc = SmartGridExtractor().split_board(crop)
    pc, layc = plan_of(sqc)
    assert _is(pc, crop) and layc.rects[0] == (0, 0, 77, 77)
    pd_, layd = plan_of(dict(sqc))
    assert pd_.data == buf.ctypes.data and pd_.stride == 640 * 3 and pd_.w == 640 and pd_.h == 700 and layd.rects[0] == (10, 40, 77, 77)
    assert layd.rects[63] == (10 + 7 * 77, 40 + 7 * 77, 77, 77)
    # grid lines past the image: numpy clamps the slices, the rectangles would lie -> analysed from the views
    smart.grid_lines_x = [0, 79, 157, 234, 310, 386, 464, 541, 700]
    sq4 = smart.split_board(board)
    assert sq4._parent is None and sq4[(7, 0)].shape == (78, 79, 3)
    assert plan_of(sq4)[1].rects[63] == (541, 542, 79, 78)


def test_plan_refuses_what_is_not_one_image():
    board = np.zeros((160, 160, 3), np.uint8)
    sq = dict(GridExtractor().split_board(board))
    assert plan_of(sq) is not None
    sq[(0, 0)] = sq[(0, 0)].copy()  # one square from elsewhere
    assert plan_of(sq) is None
    assert plan_of({}) is None
    assert plan_of({(0, 0): np.zeros((20, 20, 3), np.float32)}) is None
    flipped = board[::-1]
    assert plan_of(dict(GridExtractor().split_board(flipped))) is None  # negative row stride: packed copies instead
    own = {(0, 0): np.zeros((20, 20, 3), np.uint8)}  # a square that owns its pixels is its own parent
    p, lay = plan_of(own)
    assert _is(p, own[(0, 0)]) and lay.rects == [(0, 0, 20, 20)]
    gray = np.zeros((64, 64), np.uint8)
    p, lay = plan_of({(0, 0): gray[8:24, 8:24], (1, 0): gray[8:24, 24:40]})
    assert _is(p, gray) and lay.rects == [(8, 8, 16, 16), (24, 8, 16, 16)]
    # views whose rows would wrap around the owner's rows (a parent that starts mid-row in its buffer) are refused
    buf = np.zeros(64 * 64 + 7, np.uint8)
    odd = buf[7:].reshape(64, 64)
    assert plan_of({(0, 0): odd[0:16, 50:64]}) is None
