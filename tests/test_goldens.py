"""Golden vectors captured from the reference's own pure-Python / pure-numpy
functions (tests/golden/make_goldens.py) against (a) the oracle and (b) the
product's host-side logic.  No GPU."""
import json
import os

import numpy as np
import pytest

from oracle import cbv_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def jload(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_fen_generator_goldens():
    from chessboard_vision_amd import fen_generator as F
    g = jload("fen_generator.json")
    for c in g["get_chess_square"]:
        name, grid = F.get_chess_square(c["x"], c["y"], c["board_size"])
        assert name == c["name"] and list(grid) == c["grid"]
    for c in g["generate_fen"]:
        bm = {(gx, gy): {"fen": ch} for gx, gy, ch in c["board_map"]}
        assert F.generate_fen(bm, c["turn"]) == c["fen"]
    for c in g["map_detections"]:
        dets = [{"center": tuple(d["center"]), "class": d["class"], "conf": d["conf"]} for d in c["detections"]]
        bm = F.map_detections_to_board(dets, c["board_size"])
        assert sorted([gx, gy, v["fen"], v["conf"], v["class"]] for (gx, gy), v in bm.items()) == sorted(c["board_map"])
        assert F.generate_fen(bm) == c["fen"]


def test_synthetic_script_ends_in_known_position():
    from chessboard_vision_amd import synth as S
    g = jload("fen_generator.json")
    assert g["generate_fen"][0]["fen"].startswith("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w")
    assert len(S.position_after(16)) == 32


def _rois(sq, img):
    base = img.ctypes.data
    out = []
    for (f, r), v in sorted(sq.items()):
        off = v.ctypes.data - base
        y0, rem = divmod(off, img.strides[0])
        out.append([f, r, rem // 3, y0, v.shape[1], v.shape[0], list(v.strides)])
    return out


def test_grid_extractor_goldens():
    from chessboard_vision_amd import synth as S
    from chessboard_vision_amd.grid_extractor import GridExtractor, SmartGridExtractor
    g = jload("grid_and_reorder.json")
    img = np.zeros((620, 620, 3), np.uint8)
    assert _rois(GridExtractor().split_board(img), img) == g["linear_620"]
    sg = SmartGridExtractor()
    assert _rois(sg.split_board(img), img) == g["smart_unset_620"]
    sg.grid_lines_x, sg.grid_lines_y = list(S.CALIB_GRID_X), list(S.CALIB_GRID_Y)
    assert _rois(sg.split_board(img), img) == g["smart_calib_620"]
    # roi_table (device path) describes the same cells as the views
    tab = {(c, 7 - r): [x, y, w, h] for r, c, x, y, w, h in sg.roi_table(620, 620)}
    assert {(f, r): [x, y, w, h] for f, r, x, y, w, h, _ in g["smart_calib_620"]} == tab
    sg.grid_lines_x = [0, 79, 157, 157, 310, 386, 464, 541, 620]
    assert _rois(sg.split_board(img), img) == g["smart_degenerate_620"]
    img2 = np.zeros((400, 400, 3), np.uint8)
    assert [r[:6] for r in _rois(GridExtractor().split_board(img2), img2)] == g["linear_400"]
    # views, not copies (grid_extractor.py:46): writing through the parent shows in the square
    sq = GridExtractor().split_board(img)
    img[0, 539] = 7
    assert sq[(7, 7)][0, 0, 0] == 7


def test_reorder_goldens():
    from chessboard_vision_amd.board_detection import reorder
    for c in jload("grid_and_reorder.json")["reorder"]:
        out = reorder(np.array(c["in"], np.int32))
        assert out.shape == (4, 1, 2) and out.dtype == np.int32
        assert out.reshape(4, 2).tolist() == c["out"]


def test_hand_pattern_goldens():
    from chessboard_vision_amd.change_detector import ChangeDetectorHIP
    for c in jload("hand_pattern.json"):
        detailed = {(f, r): {"intensity": v} for f, r, v in c["in"]}
        res = ChangeDetectorHIP.classify_hand_pattern(None, detailed)
        assert res["is_hand"] == c["is_hand"] and res["is_move"] == c["is_move"]
        assert sorted(list(p) for p in res["move_candidates"]) == c["move_candidates"]


def test_history_goldens():
    from chessboard_vision_amd.piece_detector import PieceDetectorHIP
    pd = PieceDetectorHIP.__new__(PieceDetectorHIP)
    pd.history_size, pd.min_presence = 5, 0.6
    for c in jload("piece_history.json"):
        pd.detection_history = {}
        if "no_history" in c:
            assert pd._get_stable_detection((0, 0)) == c["no_history"]
            continue
        got = []
        for v in c["raw"]:
            pd._update_history((1, 1), v)
            got.append(bool(pd._get_stable_detection((1, 1))))
        assert got == c["stable"]


def test_piece_masks_and_sums_match_reference_numpy():
    """_detect_center_vs_border / _analyze_radial_symmetry (piece_detector.py:141-207)
    evaluated by the reference on 10 gray squares; oracle masks + sums and the
    product's decision arithmetic must reproduce the float64 results exactly."""
    from chessboard_vision_amd.piece_detector import PieceDetectorHIP
    z = np.load(os.path.join(G, "piece_numpy.npz"))
    cvb, sym = z["center_vs_border"], z["radial_symmetry"]
    pd = PieceDetectorHIP.__new__(PieceDetectorHIP)
    pd.circle_threshold = 0.6
    for i in range(len(sym)):
        g = z["gray_%d" % i]
        st = O.square_stats(np.ascontiguousarray(g))
        cm = np.float64(st.center_sum) / st.center_cnt
        bm = np.float64(st.border_sum) / st.border_cnt
        assert [abs(cm - bm), cm, bm] == cvb[i].tolist(), i
        rm = [np.float64(st.ring_sum[k]) / st.ring_cnt[k] for k in range(4) if st.ring_cnt[k] > 0]
        s = 0.0 if len(rm) < 2 else min(1.0, np.var(rm) / 500)
        assert s == sym[i], i
        # product decision chain on the same statistics
        res = pd._decide(st, g.shape)
        exp_has = bool(np.std(g) >= 15 and (cvb[i][0] > 40 or sym[i] > 0.6))
        assert res["has_piece"] == exp_has, i
        if np.std(g) >= 15:
            assert res["center_border_diff"] == cvb[i][0]


def _norm_noise(state, data):
    d = {}
    for k, v in data.items():
        d[k] = sorted(list(x) for x in v) if isinstance(v, set) else (list(v) if isinstance(v, tuple) else v)
    return state.name, d


def test_noise_handler_goldens():
    """Every (state, data) the reference's NoiseHandler returns over scripted and random change
    sequences (tests/golden/noise_handler.json) is reproduced by the drop-in class."""
    from chessboard_vision_amd.noise_handler import NoiseHandler, NoiseState
    seqs = jload("noise_handler.json")
    seen = set()
    for steps in seqs:
        h = NoiseHandler()
        assert h.state == NoiseState.IDLE and not h.is_blocked()
        for i, st in enumerate(steps):
            state, data = h.process({tuple(x) for x in st["changed"]})
            name, d = _norm_noise(state, data)
            assert name == st["state"] and d == st["data"], (i, name, d, st)
            assert h.is_blocked() == st["blocked"] and h.get_state_name() == st["name"]
            seen.add(d.get("message"))
    from chessboard_vision_amd.noise_handler import MESSAGES
    assert seen == set(MESSAGES)  # every transition of the machine is covered by the goldens
    h.reset()
    assert h.state == NoiseState.IDLE and h.stable_count == 0 and len(h.pending_squares) == 0


def _quad_mask(w, h, pts):
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.ones((h, w), bool)
    P = np.array(pts, float)
    for i in range(4):
        a, b = P[i], P[(i + 1) % 4]
        m &= ((b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0])) >= 0
    return m


def test_corners_from_edges_known_answers():
    """Host half of find_chessboard_corners (findContours EXTERNAL + contourArea + approxPolyDP, restated): a filled
    convex quadrilateral comes back as exactly its four vertices, counter-clockwise from the top-left one; nested
    components are not external contours; small or non-quadrilateral shapes do not qualify.  No cv2 to pin against."""
    from chessboard_vision_amd.board_detection import corners_from_edges, reorder
    w, h = 1280, 720
    pts = [(300, 100), (900, 140), (950, 650), (260, 600)]
    m = _quad_mask(w, h, pts).astype(np.uint8) * 255
    poly, n = corners_from_edges(m)
    assert n == 1 and poly.reshape(4, 2).tolist() == [[300, 100], [260, 600], [950, 650], [900, 140]]
    assert reorder(poly).reshape(4, 2).tolist() == [[300, 100], [900, 140], [260, 600], [950, 650]]
    ring = m.copy()
    ring[_quad_mask(w, h, [(320, 120), (880, 158), (930, 630), (280, 582)])] = 0
    ring[10:30, 10:60] = 255        # a small blob outside: a second external contour, too small to qualify
    ring[400:402, 600:640] = 255    # a blob inside the ring's hole: not an external contour
    poly2, n2 = corners_from_edges(ring)
    assert n2 == 2 and poly2.reshape(4, 2).tolist() == poly.reshape(4, 2).tolist()
    assert corners_from_edges(np.zeros((50, 60), np.uint8)) == (None, 0)
    small = np.zeros((300, 300), np.uint8)
    small[50:250, 50:250] = 255      # area 39601 < 100000
    assert corners_from_edges(small) == (None, 1)
    yy, xx = np.mgrid[0:700, 0:700]
    disc = (((xx - 350) ** 2 + (yy - 350) ** 2) <= 300 ** 2).astype(np.uint8) * 255   # large, but not four-cornered
    assert corners_from_edges(disc)[0] is None
    single = np.zeros((20, 20), np.uint8)
    single[5, 5] = 255
    single[10:12, 10:14] = 255
    assert corners_from_edges(single) == (None, 2)
    # an axis-aligned rectangle touching the image border
    edge = np.zeros((400, 500), np.uint8)
    edge[0:400, 0:300] = 255
    p3, _ = corners_from_edges(edge)
    assert sorted(p3.reshape(4, 2).tolist()) == [[0, 0], [0, 399], [299, 0], [299, 399]]


def test_crop_inner_squares_is_a_view_like_the_reference():
    """board_detection.py:74-82: a slice of the warped board (a view, not a copy) and the shrunken board size."""
    from chessboard_vision_amd.board_detection import crop_inner_squares
    w = np.arange(620 * 620 * 3, dtype=np.uint32).astype(np.uint8).reshape(620, 620, 3)
    c, n = crop_inner_squares(w, 620, 2)
    assert n == 616 and c.shape == (616, 616, 3) and c.base is not None and np.shares_memory(c, w)
    assert np.array_equal(c, w[2:618, 2:618])
    c0, n0 = crop_inner_squares(w, 620)
    assert n0 == 620 and c0.shape == w.shape
