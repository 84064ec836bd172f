"""The reference's own per-frame call pattern (game_session.py:124-161, calibrate_sensitivity.py:142-157) through the
drop-in classes: warp_image -> split_board -> detect_all_pieces / detect_changes_detailed, one library call per
detector and frame (cbv_squares_detect_all / cbv_squares_detect_changes) with the board uploaded at call time.
Checked against the restated reference logic on the oracle (tests/ref_logic.py, itself pinned by the reference-run
fixtures) and against the packed-view path of the same classes."""
import numpy as np
import pytest

from chessboard_vision_amd import synth as S
from helpers import oracle_frame

pytestmark = pytest.mark.gpu
W, H = 640, 480


@pytest.fixture()
def poison(gpu_ctx):
    """partial uploads land in buffers filled with 0xA5 first: a read outside the uploaded rows changes the result"""
    gpu_ctx.check(gpu_ctx.lib.cbv_debug_poison(gpu_ctx.h, 1))
    yield
    gpu_ctx.check(gpu_ctx.lib.cbv_debug_poison(gpu_ctx.h, 0))


def _grid(kind):
    from chessboard_vision_amd.grid_extractor import GridExtractor, SmartGridExtractor
    if kind == "linear":
        return GridExtractor()
    g = SmartGridExtractor()
    g.grid_lines_x, g.grid_lines_y = list(S.CALIB_GRID_X), list(S.CALIB_GRID_Y)
    return g


def _assert_same_detector_state(a, b, keys, tag):
    assert a.cached_results == b.cached_results, tag
    assert {k: list(v) for k, v in a.detection_history.items()} == {k: list(v) for k, v in b.detection_history.items()}, tag
    assert set(a.reference_squares.keys()) == set(b.reference_squares.keys()), tag
    for pos in keys:
        if pos in a.reference_squares:
            assert np.array_equal(a.reference_squares[pos], b.reference_squares[pos]), (tag, pos)


@pytest.mark.parametrize("grid", ["linear", "smart"])
def test_session_call_pattern_matches_reference_logic(gpu_ctx, oracle, poison, grid):
    """GameSession.on_frame's calls on a stream with moves: the one-call path == the packed-view path == the restated
    reference, for every result dict, visual_changes, cache, history and reference plane after every frame; with
    squares_to_check on most frames (the session's smart scan), a mid-stream update_references, and frames without
    smoothing / without delta."""
    from chessboard_vision_amd.board_detection import warp_image
    from chessboard_vision_amd.piece_detector import PieceDetector
    from ref_logic import RefPieceDetector
    ge = _grid(grid)
    pts = S.scaled_corners(W, H)
    fast, views, ref = PieceDetector(), PieceDetector(), RefPieceDetector(hough={})
    for t in range(22):
        f = oracle_frame(W, H, "dim" if grid == "smart" else "normal", frame_idx=t, frames_per_ply=3)
        warped = warp_image(f, pts)[0]
        assert np.array_equal(warped, oracle.warp_image(f, pts)[0])
        sq = ge.split_board(warped)
        copies = {k: v.copy() for k, v in sq.items()}  # owners of their pixels: not views of one image
        if t == 0:
            for d, s_ in ((fast, sq), (views, copies), (ref, sq)):
                d.update_references(s_)
        check = None if t % 5 == 0 else ({p for p in S.position_for_frame(t, 3)} | {(4, 3), (4, 4), (2, 5)})
        kw = dict(squares_to_check=check, use_smoothing=(t != 9), use_delta=(t not in (11, 12)))
        r1, v1 = fast.detect_all_pieces(sq, **kw)
        r2, v2 = views.detect_all_pieces(copies, **kw)
        r3, v3 = ref.detect_all_pieces(sq, **kw)
        assert v1 == v2 == v3, t
        assert list(r1.keys()) == list(r3.keys()) == list(sq.keys())
        for pos in r1:
            assert r1[pos] == r3[pos], (t, pos, r1[pos], r3[pos])
            assert r1[pos] == r2[pos], (t, pos, r1[pos], r2[pos])
        _assert_same_detector_state(fast, ref, sq.keys(), ("ref", t))
        _assert_same_detector_state(fast, views, sq.keys(), ("views", t))
        if t == 13:  # what the session does once it accepted a move (game_session.py:219-223)
            for d, s_ in ((fast, sq), (views, copies), (ref, sq)):
                d.update_references(s_)
            _assert_same_detector_state(fast, ref, sq.keys(), ("after update_references", t))


def test_pixels_drawn_on_the_board_after_warp_image_are_seen(gpu_ctx, oracle):
    """game_session.py:179 draws on `warped`; nothing of the board may be cached on the device between warp_image and
    detect_all_pieces: the detector must see what the host array holds WHEN IT IS CALLED."""
    from chessboard_vision_amd.board_detection import warp_image
    from chessboard_vision_amd.grid_extractor import GridExtractor
    from chessboard_vision_amd.piece_detector import PieceDetector
    from ref_logic import RefPieceDetector
    pts = S.scaled_corners(W, H)
    f = oracle_frame(W, H, "normal", frame_idx=0)
    det, ref = PieceDetector(), RefPieceDetector(hough={})
    ge = GridExtractor()
    warped = warp_image(f, pts)[0]
    sq = ge.split_board(warped)
    det.update_references(sq)
    ref.update_references(sq)
    r0, v0 = det.detect_all_pieces(sq)
    assert v0 == set() and not r0[(3, 3)]["has_piece"]
    ref.detect_all_pieces(sq)
    y, x = np.mgrid[:620, :620]
    for step, pos in enumerate([(3, 3), (4, 3), (2, 4)]):  # empty squares of the start position
        warped = warp_image(f, pts)[0]
        sq = ge.split_board(warped)
        # draw a disc on an empty square of the array warp_image returned, in place, then detect
        cy, cx = (7 - pos[1]) * 77 + 38, pos[0] * 77 + 38 + step
        warped[(y - cy) ** 2 + (x - cx) ** 2 <= 31 ** 2] = (250, 250, 245)
        r, v = det.detect_all_pieces(sq)
        r_ref, v_ref = ref.detect_all_pieces(sq)
        assert v == v_ref and pos in v, (step, v, v_ref)
        assert r == r_ref
        assert det.cached_results[pos]["has_piece"] and det.cached_results[pos]["method"] is not None
    # and the same for the ChangeDetector
    from chessboard_vision_amd.change_detector import ChangeDetector
    from ref_logic import RefChangeDetector
    cd, rcd = ChangeDetector(), RefChangeDetector(hough={})
    clean = ge.split_board(warp_image(f, pts)[0])
    cd.calibrate(clean)
    rcd.calibrate(clean)
    assert cd.detect_changes_detailed(clean) == {}
    warped = warp_image(f, pts)[0]
    sq = ge.split_board(warped)
    warped[(y - cy) ** 2 + (x - cx) ** 2 <= 31 ** 2] = (250, 250, 245)
    d = cd.detect_changes_detailed(sq)
    assert d == rcd.detect_changes_detailed(sq) and set(d) == {pos} and d[pos]["is_circular"]


@pytest.mark.parametrize("blur", [5, 8, 1])
def test_calibration_tool_call_pattern_matches_reference_logic(gpu_ctx, oracle, poison, blur):
    """calibrate_sensitivity.py:128-157: attributes written per frame (an even `blur_kernel` included), calibrate at one
    frame, detect_changes_detailed + detect_changes per frame; + the EMA update and focus squares the class offers.
    One-call path == packed-view path == restated reference."""
    from chessboard_vision_amd.board_detection import warp_image
    from chessboard_vision_amd.change_detector import ChangeDetector
    from ref_logic import RefChangeDetector
    ge = _grid("smart")
    pts = S.scaled_corners(W, H)
    fast, views, ref = ChangeDetector(), ChangeDetector(), RefChangeDetector(hough={})
    for d in (fast, views, ref):
        d.z_threshold, d.initial_variance, d.alpha, d.blur_kernel = 2.1, 180, 0.07, blur
        d._kernel = max(1, blur | 1)
    seen_circular = False
    for t in range(12):
        f = oracle_frame(W, H, "normal", frame_idx=t, frames_per_ply=2)
        sq = ge.split_board(warp_image(f, pts)[0])
        copies = {k: v.copy() for k, v in sq.items()}
        if t == 1:
            for d, s_ in ((fast, sq), (views, copies), (ref, sq)):
                d.calibrate(s_)
        d1, d2, d3 = fast.detect_changes_detailed(sq), views.detect_changes_detailed(copies), ref.detect_changes_detailed(sq)
        assert d1 == d3, (t, d1, d3)
        assert d1 == d2 and list(d1.keys()) == list(d3.keys()), t
        assert fast.detect_changes(sq) == ref.detect_changes(sq), t
        seen_circular |= any(v["is_circular"] for v in d1.values())
        if t == 6:
            for d in (fast, views):
                d.set_focus_squares([(4, 1), (4, 3), (6, 7)])
            ref.focus_squares = {(4, 1), (4, 3), (6, 7)}
        if t in (4, 7, 8):
            for d, s_ in ((fast, sq), (views, copies), (ref, sq)):
                d.update_all_references(s_)
            for pos in sq:
                assert np.array_equal(fast.means[pos], ref.means[pos]) and np.array_equal(fast.variances[pos], ref.variances[pos]), (t, pos)
    assert seen_circular


def test_boards_that_are_views_themselves(gpu_ctx, oracle, poison):
    """The squares' parent may be a crop of a wider frame (board_detection.crop_inner_squares), a reshaped capture
    buffer, or unknown (a plain dict of the views): rectangles are then taken in the owner's rows; a pitch more than
    twice the board's width goes through the 2-D copy.  Every variant == the restated reference."""
    from chessboard_vision_amd.board_detection import crop_inner_squares, warp_image
    from chessboard_vision_amd.grid_extractor import GridExtractor
    from chessboard_vision_amd.piece_detector import PieceDetector
    from ref_logic import RefPieceDetector
    pts = S.scaled_corners(W, H)
    ge = GridExtractor()
    for variant in ("crop", "plain_dict", "wide_pitch", "reshaped_buffer"):
        det, ref = PieceDetector(), RefPieceDetector(hough={})
        for t in range(5):
            f = oracle_frame(W, H, "normal", frame_idx=t, frames_per_ply=2)
            warped = warp_image(f, pts)[0]
            if variant == "crop":
                board = crop_inner_squares(warped, 620, 6)[0]
                sq = ge.split_board(board)
            elif variant == "plain_dict":
                sq = dict(ge.split_board(warped))
            elif variant == "wide_pitch":
                wide = np.full((700, 1500, 3), 77, np.uint8)
                wide[30:650, 400:1020] = warped
                sq = ge.split_board(wide[30:650, 400:1020])
            else:
                buf = np.zeros(620 * 620 * 3 + 11, np.uint8)
                buf[:620 * 620 * 3] = warped.reshape(-1)
                sq = dict(ge.split_board(buf[:620 * 620 * 3].reshape(620, 620, 3)))
            r, v = det.detect_all_pieces(sq)
            r_ref, v_ref = ref.detect_all_pieces(sq)
            assert v == v_ref and r == r_ref, (variant, t)
            _assert_same_detector_state(det, ref, sq.keys(), (variant, t))


def test_warp_image_uploads_only_what_it_samples(gpu_ctx, oracle, poison):
    """cbv_warp_perspective sends the rows of the quad's footprint only; with the staging buffer poisoned the result
    still equals the oracle's for quads inside, partly outside and far outside the frame, strided input included."""
    from chessboard_vision_amd.board_detection import get_perspective_transform, warp_image, warp_perspective
    rng = np.random.default_rng(5)
    f = rng.integers(0, 256, (H + 20, W + 40, 3), dtype=np.uint8)
    frames = [np.ascontiguousarray(f[:H, :W]), f[10:H + 10, 20:W + 20]]  # tight rows, and a view with a wider pitch
    quads = [S.scaled_corners(W, H), np.float32([[300, 200], [420, 210], [290, 330], [430, 320]]),
             np.float32([[-60, -40], [700, -10], [-30, 520], [720, 500]]), np.float32([[500, 400], [900, 380], [520, 700], [880, 720]]),
             np.float32([[5, 470], [630, 465], [2, 479], [639, 479]])]
    for fr in frames:
        for q in quads:
            got = warp_image(fr, q)[0]
            assert np.array_equal(got, oracle.warp_image(fr, q)[0])
        for k in range(8):
            q = np.float32([[rng.uniform(-50, 300), rng.uniform(-50, 200)], [rng.uniform(340, 700), rng.uniform(-50, 200)],
                            [rng.uniform(-50, 300), rng.uniform(260, 530)], [rng.uniform(340, 700), rng.uniform(260, 530)]])
            M = get_perspective_transform(q, np.float32([[0, 0], [200, 0], [0, 150], [200, 150]]))
            assert np.array_equal(warp_perspective(fr, M, (200, 150), rot180=bool(k & 1)),
                                  oracle.rotate180(oracle.warp_perspective(fr, M, (200, 150))) if k & 1 else oracle.warp_perspective(fr, M, (200, 150)))


def test_one_call_entry_points_fail_loudly(gpu_ctx):
    """Argument and state errors of the one-call entry points come back as CBV_ERR_* with a message; a rejected call
    leaves the set usable."""
    import ctypes as C
    from chessboard_vision_amd import _native as N
    from chessboard_vision_amd._squares import SquareSet, plan_of
    from chessboard_vision_amd.grid_extractor import GridExtractor
    board = np.random.default_rng(1).integers(0, 255, (160, 160, 3), dtype=np.uint8)
    sq = GridExtractor().split_board(board)
    img, lay = plan_of(sq)
    ss = SquareSet(gpu_ctx)
    prm = N.DetectParams()
    prm.change_threshold, prm.circle_threshold, prm.use_delta = 25.0, 0.6, 1
    prm.hough = N.HoughParams(1.2, 100.0, 25.0, 0.2, 0.55)
    out = np.zeros(64, N.record_dtype(N.PieceResult))
    lib = gpu_ctx.lib
    bad = (N.Roi * 64)()
    for i in range(64):
        bad[i].x0, bad[i].y0, bad[i].w, bad[i].h = lay.rects[i]
    bad[63].x0 = 150  # 150 + 20 > 160
    assert lib.cbv_squares_detect_all(ss.h, img, bad, 64, prm, out.ctypes.data) == -1
    assert b"outside" in lib.cbv_last_error(gpu_ctx.h)
    assert lib.cbv_squares_detect_all(ss.h, img, lay.rois, 0, prm, out.ctypes.data) == -1
    assert lib.cbv_squares_detect_all(ss.h, img, lay.rois, 64, None, out.ctypes.data) == -1
    prm.hough.max_radius_ratio = 9.0
    assert lib.cbv_squares_detect_all(ss.h, img, lay.rois, 64, prm, out.ctypes.data) == -1
    prm.hough.max_radius_ratio = 0.55
    cp = N.ChangeParams()
    cp.z_threshold, cp.select, cp.circle_threshold, cp.hough = 2.5, lay.all_mask, 0.6, prm.hough
    cout = np.zeros(64, N.record_dtype(N.ChangeResult))
    assert lib.cbv_squares_detect_changes(ss.h, img, lay.rois, 64, 5, cp, cout.ctypes.data) == -4  # not calibrated
    assert lib.cbv_squares_set_ref_mask(ss.h, 1) == -4                                              # nothing loaded yet
    # and after the refusals the set works
    rows = ss.detect_all(img, lay, prm)
    assert len(rows) == 64 and all(r[2] == 1 and r[4] == 1 for r in rows)  # no references yet: every square changed, evaluated
    gray_img = N.HostImage()
    gray_img.data, gray_img.w, gray_img.h, gray_img.stride, gray_img.cn = img.data, 160, 160, 480, 2
    assert lib.cbv_squares_detect_all(ss.h, gray_img, lay.rois, 64, prm, out.ctypes.data) == -1   # 2 channels


def test_grid_change_mid_stream_drops_the_references(gpu_ctx, oracle):
    """Switching the grid (other square shapes) under a live detector voids the device planes: every square reads as
    changed and is evaluated afresh, like a detector without references; going back does not revive old planes."""
    from chessboard_vision_amd.board_detection import warp_image
    from chessboard_vision_amd.piece_detector import PieceDetector
    from ref_logic import RefPieceDetector
    pts = S.scaled_corners(W, H)
    lin, smart = _grid("linear"), _grid("smart")
    det = PieceDetector()
    f = oracle_frame(W, H, "normal", frame_idx=0)
    warped = warp_image(f, pts)[0]
    det.update_references(lin.split_board(warped))
    r, v = det.detect_all_pieces(lin.split_board(warped))
    assert v == set()
    r, v = det.detect_all_pieces(smart.split_board(warped))  # 77..80-px squares instead of 77 x 77
    assert len(v) == 64 and len(det.reference_squares) == 64  # all changed (no references), all refreshed afterwards
    ref = RefPieceDetector(hough={})
    want, _ = ref.detect_all_pieces(smart.split_board(warped))
    assert {p: x["method"] for p, x in det.cached_results.items()} == {p: x["method"] for p, x in want.items()}
    r, v = det.detect_all_pieces(smart.split_board(warped))
    assert v == set()
    r, v = det.detect_all_pieces(lin.split_board(warped))
    assert len(v) == 64


def test_detectors_on_one_context_from_two_threads(gpu_ctx, oracle):
    """Two detectors driven from two threads on the one process-wide context (ctypes drops the GIL): the one-call entry
    points hold the context's lock for their whole call and share its pinned staging safely."""
    import threading
    from chessboard_vision_amd.board_detection import warp_image
    from chessboard_vision_amd.change_detector import ChangeDetector
    from chessboard_vision_amd.piece_detector import PieceDetector
    from ref_logic import RefChangeDetector, RefPieceDetector
    pts = S.scaled_corners(W, H)
    ge = _grid("smart")
    boards = [warp_image(oracle_frame(W, H, "normal", frame_idx=t, frames_per_ply=2), pts)[0] for t in range(8)]
    rp, rc = RefPieceDetector(hough={}), RefChangeDetector(hough={})
    rc.calibrate(ge.split_board(boards[0]))
    want_p = [rp.detect_all_pieces(ge.split_board(b)) for b in boards]
    want_c = [rc.detect_changes_detailed(ge.split_board(b)) for b in boards]
    errs = []

    def pieces():
        d = PieceDetector()
        for b, w_ in zip(boards, want_p):
            if d.detect_all_pieces(ge.split_board(b)) != w_:
                errs.append("pieces")

    def changes():
        d = ChangeDetector()
        d.calibrate(ge.split_board(boards[0]))
        for b, w_ in zip(boards, want_c):
            if d.detect_changes_detailed(ge.split_board(b)) != w_:
                errs.append("changes")

    ts = [threading.Thread(target=f) for f in (pieces, changes, pieces)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_session_loop_example_recognises_the_scripted_game(gpu_ctx):
    """examples/session_on_frame.py: GameSession.on_frame's calls on the drop-in classes (enhancer, warp_image, smart
    grid, smart-scan sets from the rules engine, detect_all_pieces, NoiseHandler, the stability rule) over the whole
    scripted game at 1080p: all 16 plies are recognised and the FEN is the closed Ruy Lopez."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "session_on_frame.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame")]
    assert [l.split()[2] for l in lines] == ["e2e4", "e7e5", "g1f3", "b8c6", "f1b5", "a7a6", "b5a4", "g8f6", "e1g1", "f8e7", "f1e1",
                                             "b7b5", "a4b3", "d7d6", "c2c3", "e8g8"]
    assert "final FEN r1bq1rk1/2p1bppp/p1np1n2/1p2p3/4P3/1BP2N2/PP1P1PPP/RNBQR1K1 w - - 1 9" in r.stdout


def test_planes_written_by_the_host_take_effect(gpu_ctx, oracle):
    """`detector.variances[pos] = array` / `means[pos] = array` (the dict-like views write the device planes): the
    statistics kernels read sqrt(variance) from a plane kept beside the variance, so a variance written by the host must
    refresh it (correctly rounded: np.sqrt on float32, bit for bit)."""
    from chessboard_vision_amd.board_detection import warp_image
    from chessboard_vision_amd.change_detector import ChangeDetector
    from ref_logic import RefChangeDetector
    pts = S.scaled_corners(W, H)
    ge = _grid("linear")
    a = ge.split_board(warp_image(oracle_frame(W, H, "normal", frame_idx=0), pts)[0])
    b = ge.split_board(warp_image(oracle_frame(W, H, "normal", frame_idx=5, frames_per_ply=1), pts)[0])
    cd, ref = ChangeDetector(), RefChangeDetector(hough={})
    cd.calibrate(a)
    ref.calibrate(a)
    rng = np.random.default_rng(3)
    for pos in list(a)[::3]:
        v = rng.uniform(10.0, 900.0, a[pos].shape[:2]).astype(np.float32)
        m = rng.uniform(0.0, 255.0, a[pos].shape[:2]).astype(np.float32)
        cd.variances[pos] = v
        cd.means[pos] = m
        ref.variances[pos] = v.copy()
        ref.means[pos] = m.copy()
    got, want = cd.detect_changes_detailed(b), ref.detect_changes_detailed(b)
    assert got == want and len(got) > 20
    cd.update_all_references(b)
    ref.update_all_references(b)
    assert cd.detect_changes_detailed(a) == ref.detect_changes_detailed(a)


def test_one_call_paths_take_the_second_hough_pass_when_a_square_needs_it(gpu_ctx, oracle):
    """White-noise squares hold more accumulator maxima than HoughCircles' first pass keeps; the one-call entry points
    read the second-pass list back with the results and launch the second pass only when it is not empty.  Both cases,
    for detect_all_pieces and detect_changes_detailed, against the restated reference logic."""
    from chessboard_vision_amd import _native as N
    from chessboard_vision_amd._squares import SquareSet
    from chessboard_vision_amd.change_detector import ChangeDetector
    from chessboard_vision_amd.grid_extractor import GridExtractor
    from chessboard_vision_amd.piece_detector import PieceDetector
    from ref_logic import RefChangeDetector, RefPieceDetector
    rng = np.random.default_rng(11)
    ge = GridExtractor()
    calm = np.full((620, 620, 3), 120, np.uint8)
    calm += rng.integers(0, 4, calm.shape, dtype=np.uint8)
    noisy = calm.copy()
    patterns = [rng.integers(0, 2, (77, 77), dtype=np.uint8) * 255, rng.integers(0, 256, (77, 77), dtype=np.uint8),
                rng.integers(0, 2, (77, 77), dtype=np.uint8) * 200 + 20]
    for (c, r), pat in zip(((1, 1), (5, 2), (6, 6)), patterns):  # three squares of noise, the rest of the board calm
        noisy[r * 77:(r + 1) * 77, c * 77:(c + 1) * 77] = pat[:, :, None]
    # with an accumulator threshold of 15 the noise squares really overflow the first pass (else this test exercises nothing)
    probe = SquareSet()
    probe.load(ge.split_board(noisy), 5)
    assert sum(h.n_centres > 512 for h in probe.hough(param2=15)) >= 2, [h.n_centres for h in probe.hough(param2=15)]
    det, ref = PieceDetector(), RefPieceDetector(hough={"param2": 15})
    cd, cref = ChangeDetector(), RefChangeDetector(hough={"param2": 15})
    det.hough_param2 = cd.piece_detector.hough_param2 = 15
    cd.calibrate(ge.split_board(calm))
    cref.calibrate(ge.split_board(calm))
    for board in (calm, noisy, calm, noisy):
        sq = ge.split_board(board)
        r, v = det.detect_all_pieces(sq)
        r_ref, v_ref = ref.detect_all_pieces(sq)
        assert v == v_ref and r == r_ref
        assert cd.detect_changes_detailed(sq) == cref.detect_changes_detailed(sq)
