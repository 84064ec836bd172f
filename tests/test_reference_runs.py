"""CPU side of the reference-run fixtures: the oracle (pixel functions) and tests/ref_logic.py (the detectors' host
logic restated on the oracle) against what the REFERENCE'S OWN CLASSES returned on the same inputs
(tests/golden/ref_*; recorded by tests/golden/make_reference_runs.py under the oracle-backed cv2 shim).

What passing means: the oracle's restatement of the reference's numpy arithmetic (the float32 HSV section of
apply_color_profile, EMA, z-scores, ring/corner masks, the detect_all_pieces control flow) is the reference's, bit for
bit.  The OpenCV-side numbers in the fixtures are the oracle's own and prove nothing about OpenCV (parity unpinned).
The -m gpu twin of this file (test_gpu_reference_runs.py) compares the HIP classes with the same fixtures directly."""
import numpy as np
import pytest

import refrun as R
from chessboard_vision_amd import synth as S
from chessboard_vision_amd.grid_extractor import GridExtractor, SmartGridExtractor
from helpers import oracle_frame

W, H = 640, 480


def test_fixture_inputs_are_reproducible(oracle):
    """The committed hashes of the generated inputs: guards the generator itself (synth) against drift."""
    meta = R.load_json("ref_enhancer.json")["c1_640x480"]
    assert R.sha(oracle_frame(W, H, "dim", stream_id=0, frame_idx=0)) == meta["input_sha256"]
    enh = R.load_npz("ref_enhancer.npz")
    assert np.array_equal(enh["in_scene_dim"], oracle_frame(160, 120, "dim", stream_id=1, frame_idx=3))


def test_apply_color_profile_numpy_section_is_the_references(oracle):
    """frame_enhancer.py:56-99 run for real (convertScaleAbs / BGR2HSV / HSV2BGR via the oracle, the float32 numpy
    section by numpy) == the oracle's one-function restatement, for six profiles on five frames."""
    meta, enh = R.load_json("ref_enhancer.json"), R.load_npz("ref_enhancer.npz")
    n = 0
    for pname, prof in meta["profiles"].items():
        for fname in ("smooth", "noise", "sweep", "scene_dim", "odd"):
            got = oracle.apply_color_profile(enh["in_" + fname], prof)
            assert np.array_equal(got, enh["profile_%s_%s" % (pname, fname)]), (pname, fname)
            n += 1
    assert n == 30


def test_reference_ctor_facts():
    meta = R.load_json("ref_enhancer.json")
    assert meta["ctor"]["profile_from_cwd"] == S.SHIPPED_PROFILE
    assert meta["ctor"]["sharpen_kernel"] == [[-1, -1, -1], [-1, 9, -1], [-1, -1, -1]]
    assert meta["ctor"]["clahe"] == [3.0, [8, 8]]
    assert any("Usando Python" in line or "PYTHON" in line for line in meta["import_log"])  # the selector's fallback print


def test_enhancement_stages_and_call_order(oracle):
    meta, enh = R.load_json("ref_enhancer.json"), R.load_npz("ref_enhancer.npz")
    for fname in ("smooth", "scene_dim", "odd"):
        f = enh["in_" + fname]
        assert np.array_equal(oracle.correct_lighting(f), enh["lighting_" + fname])
        assert np.array_equal(oracle.bilateral(f), enh["noise_" + fname])
        assert np.array_equal(oracle.filter3x3(f), enh["sharpen_" + fname])
        assert np.array_equal(oracle.normalize_minmax(f), enh["normalize_" + fname])
        g, b, _ = oracle.prepare_analysis(f)
        assert np.array_equal(g, enh["gray_" + fname]) and np.array_equal(b, enh["binary_" + fname])
        for pname in ("shipped", "radical", "none"):
            prof = meta["profiles"][pname] if pname != "none" else {}
            assert np.array_equal(oracle.process_pipeline(f, prof), enh["pipeline_%s_%s" % (pname, fname)]), (pname, fname)
    calls = [c[0] for c in meta["calls"]["pipeline_shipped_smooth"]]
    assert calls == ["convertScaleAbs", "cvtColor", "split", "merge", "cvtColor", "cvtColor", "split", "CLAHE.apply", "merge",
                     "cvtColor", "bilateralFilter", "filter2D", "normalize"]
    bil = [c for c in meta["calls"]["pipeline_shipped_smooth"] if c[0] == "bilateralFilter"][0][1]
    assert bil == {"d": 9, "sigmaColor": 75, "sigmaSpace": 75}
    k = [c for c in meta["calls"]["pipeline_shipped_smooth"] if c[0] == "filter2D"][0][1]
    assert k["ddepth"] == -1 and k["kernel_dtype"] == "int64"


def test_c1_single_640x480_frame_on_cpu(oracle):
    """BASELINE.json configs[0]: one 640x480 frame through process_pipeline (+ prepare_analysis) on the CPU."""
    meta = R.load_json("ref_enhancer.json")["c1_640x480"]
    out = oracle.process_pipeline(oracle_frame(W, H, "dim"), S.SHIPPED_PROFILE)
    assert R.sha(out) == meta["pipeline_sha256"] and int(out.sum(dtype=np.int64)) == meta["pipeline_sum"]
    g, b, _ = oracle.prepare_analysis(out)
    assert R.sha(g) == meta["gray_sha256"] and R.sha(b) == meta["binary_sha256"]


def test_warp_image(oracle):
    wz = R.load_npz("ref_warp.npz")
    img = oracle_frame(W, H, "normal", stream_id=2, frame_idx=5)
    warped, M, bs = oracle.warp_image(img, wz["pts_calib"])
    assert bs == 620 and np.array_equal(M, wz["M_calib"])
    assert R.sha(warped) == bytes(wz["sha_calib"]).hex() and np.array_equal(warped[::40], wz["rows_calib"])
    from chessboard_vision_amd.board_detection import reorder
    assert np.array_equal(reorder(wz["corners_small"]), wz["ordered_small"])
    w2, M2, bs2 = oracle.warp_image(wz["in_small"], wz["ordered_small"], display_size=(300, 196), margin=100)
    assert bs2 == 96 and np.array_equal(M2, wz["M_small"]) and np.array_equal(w2, wz["warp_small"])
    w3, M3, _ = oracle.warp_image(wz["in_small"], wz["pts_outside"], display_size=(228, 400), margin=100)
    assert np.array_equal(M3, wz["M_outside"]) and np.array_equal(w3, wz["warp_outside"])


# ------------------------------------------------------------------------------------------------------------------
def _grid(kind):
    if kind.startswith("smart"):
        ge = SmartGridExtractor()
        ge.grid_lines_x, ge.grid_lines_y = list(S.CALIB_GRID_X), list(S.CALIB_GRID_Y)
        return ge
    return GridExtractor()


def drive_piece_sequence(det, fx, warp_image, check_frame=None):
    """Replays ref_piece_sequence.json on a PieceDetector-like `det`; returns nothing, asserts per frame."""
    sc = fx["script"]
    pts = S.scaled_corners(W, H)
    ge = _grid(fx["grid"])
    for rec in fx["frames"]:
        i = rec["i"]
        img = oracle_frame(W, H, fx["scene"], stream_id=fx["stream_id"], frame_idx=i, frames_per_ply=sc["frames_per_ply"])
        warped = warp_image(img, pts)[0]
        assert R.sha(warped) == rec["warped_sha256"], i
        squares = ge.split_board(warped)
        to_check = None if rec["to_check_bits"] is None else R.unbits(rec["to_check_bits"])
        assert to_check == (None if i % sc["full_scan_every"] == 0 else R.check_set_for(i, sc["frames_per_ply"]))
        results, visual = det.detect_all_pieces(squares, squares_to_check=to_check, **rec["kwargs"])
        assert R.bits(visual) == rec["visual_bits"], i
        assert R.result_rows(results) == rec["results"], (i, [(a, b) for a, b in zip(R.result_rows(results), rec["results"]) if a != b][:3])
        assert R.detector_state(det) == rec["state"], i
        if "state_after_update_references" in rec:
            det.update_references(squares)
            assert R.detector_state(det) == rec["state_after_update_references"], i
        if check_frame:
            check_frame(i, results)


def test_piece_detector_sequence_logic_is_the_references(oracle):
    """tests/ref_logic.RefPieceDetector == piece_detector.PieceDetector over 26 frames with squares_to_check, a
    mid-stream update_references, use_smoothing=False and use_delta=False calls: results, visual changes, reference
    planes, cache and history after every call."""
    from ref_logic import RefPieceDetector
    fx = R.load_json("ref_piece_sequence.json")
    det = RefPieceDetector(hough=fx["settings"])
    drive_piece_sequence(det, fx, oracle.warp_image)
    # calibrate_reference / get_occupied_squares / detect_piece on single squares (BGR and gray input)
    det2 = RefPieceDetector(hough=fx["settings"])
    img = oracle_frame(W, H, "normal", stream_id=5, frame_idx=0, frames_per_ply=3)
    squares = _grid(fx["grid"]).split_board(oracle.warp_image(img, S.scaled_corners(W, H))[0])
    det2.calibrate_reference(squares)
    assert R.detector_state(det2) == fx["calibrate_reference"]["state"]
    assert R.result_rows(det2.cached_results) == fx["calibrate_reference"]["cached"]
    assert R.bits(det2.get_occupied_squares(squares)) == fx["get_occupied_squares_bits"]
    from ref_logic import detect_piece
    for s in fx["detect_piece"]:
        pos = tuple(s["pos"])
        assert R.result_rows({pos: detect_piece(squares[pos], hough=fx["settings"])[0]})[0] == s["bgr"]
        assert R.result_rows({pos: detect_piece(oracle.bgr2gray(np.ascontiguousarray(squares[pos])), hough=fx["settings"])[0]})[0] == s["gray_input"]


def test_detect_piece_every_branch(oracle):
    from ref_logic import detect_piece
    rows, arrs = R.load_json("ref_piece_shapes.json"), R.load_npz("ref_piece_shapes.npz")
    methods = set()
    for row in rows:
        hough = dict(min_radius_ratio=row["ratios"][0], max_radius_ratio=row["ratios"][1])
        if "hough_param2" in row:
            hough.update(param1=row["hough_param1"], param2=row["hough_param2"])
        res = detect_piece(arrs[row["name"]], hough=hough)[0]
        assert R.result_rows({(0, 0): res})[0][2:] == row["result"], row
        methods.add(row["result"][1])
    assert methods == {None, "hough", "tower_top", "center_diff", "symmetry"}


def drive_change_sequence(make_detector, run, warp_image, planes=None):
    cd = make_detector()
    pts = S.scaled_corners(W, H)
    ge = GridExtractor()
    for rec in run["frames"]:
        i = rec["i"]
        for k, v in run["attrs"].items():
            setattr(cd, k, v)
        if "blur_kernel" in run["attrs"]:
            cd._kernel = max(1, run["attrs"]["blur_kernel"] | 1)
        img = oracle_frame(W, H, run["scene"], stream_id=run["stream_id"], frame_idx=i, frames_per_ply=run["frames_per_ply"])
        squares = ge.split_board(warp_image(img, pts)[0])
        if rec.get("calibrated"):
            cd.calibrate(squares)
        if rec.get("focus_set"):
            cd.set_focus_squares([tuple(p) for p in run["focus"]])
        if rec.get("focus_cleared"):
            cd.clear_focus()
        assert cd.get_focus_count() == rec["focus_count"]
        detailed = cd.detect_changes_detailed(squares) if cd.is_calibrated else {}
        changes = cd.detect_changes(squares) if cd.is_calibrated else {}
        got = [[p[0], p[1], v["z_score"], v["pct_changed"], v["intensity"], bool(v["is_circular"]), v["center_ratio"]] for p, v in detailed.items()]
        assert got == rec["detailed"], (run["name"], i, got, rec["detailed"])
        assert [[p[0], p[1], v] for p, v in changes.items()] == rec["changes"], (run["name"], i)
        if detailed:
            pat = cd.classify_hand_pattern(detailed)
            assert {"is_hand": pat["is_hand"], "is_move": pat["is_move"],
                    "move_candidates": sorted(list(p) for p in pat["move_candidates"])} == rec["pattern"]
        if rec.get("ema"):
            cd.update_all_references(squares)
        if cd.is_calibrated:
            assert R.planes_sha(cd.means) == rec["means_sha256"], (run["name"], i, "means")
            assert R.planes_sha(cd.variances) == rec["vars_sha256"], (run["name"], i, "variances")
    if planes is not None:
        for pos in [(4, 1), (4, 3), (0, 7)]:
            assert np.array_equal(cd.means[pos], planes["%s_mean_%d_%d" % (run["name"], pos[0], pos[1])])
            assert np.array_equal(cd.variances[pos], planes["%s_var_%d_%d" % (run["name"], pos[0], pos[1])])
    return cd


@pytest.mark.parametrize("run_idx", [0, 1, 2])
def test_change_detector_sequence_logic_is_the_references(oracle, run_idx):
    """tests/ref_logic.RefChangeDetector == change_detector.ChangeDetectorPython: tool-written attributes (z_threshold,
    initial_variance, alpha, blur_kernel incl. an even one), EMA in float32 with weak Python scalars, focus squares."""
    from ref_logic import RefChangeDetector
    fx = R.load_json("ref_change_sequence.json")
    drive_change_sequence(lambda: RefChangeDetector(hough={}), fx["runs"][run_idx], oracle.warp_image, R.load_npz("ref_change_planes.npz"))


def test_change_detector_reference_regression_case_recorded(oracle):
    """The reference's own test (test_change_detector_regression.py:31-54), run for real when the fixtures were
    recorded: the recorded outcome, and the oracle-side logic giving the same."""
    from ref_logic import RefChangeDetector
    fx = R.load_json("ref_change_sequence.json")
    assert fx["regression_case"]["changes"] == [[3, 3, 100.0]]
    assert fx["regression_case"]["detailed"] == [[3, 3, 25.5, 100.0, "TOTAL", False]]
    cd = RefChangeDetector(hough={})
    sq = {(c, r): np.zeros((50, 50), np.uint8) for r in range(8) for c in range(8)}
    cd.calibrate(sq)
    sq[(3, 3)] = np.full((50, 50), 255, np.uint8)
    det = cd.detect_changes_detailed(sq)
    assert [[p[0], p[1], v["z_score"], v["pct_changed"], v["intensity"], v["is_circular"]] for p, v in det.items()] == fx["regression_case"]["detailed"]
    # update_all_references before calibrate == calibrate
    cd2 = RefChangeDetector(hough={})
    img = oracle_frame(W, H, "normal", stream_id=7, frame_idx=0, frames_per_ply=2)
    cd2.update_all_references(GridExtractor().split_board(oracle.warp_image(img, S.scaled_corners(W, H))[0]))
    assert cd2.is_calibrated and R.planes_sha(cd2.means) == fx["update_before_calibrate"]["means_sha256"]
    assert R.planes_sha(cd2.variances) == fx["update_before_calibrate"]["vars_sha256"]


def chain_reference(oracle, run):
    """The composed chain on the oracle + ref_logic for one run of ref_chain_sequence.json; yields per-frame dicts."""
    from ref_logic import RefPieceDetector
    det = RefPieceDetector(hough={})
    pts = S.scaled_corners(W, H)
    ge = _grid(run["grid"])
    for rec in run["frames"]:
        i = rec["i"]
        img = oracle_frame(W, H, run["scene"], stream_id=run["stream_id"], frame_idx=i, frames_per_ply=run["frames_per_ply"])
        e = oracle.process_pipeline(img, run["profile"])
        warped = oracle.warp_image(e, pts)[0]
        if run["rot180"]:
            warped = oracle.rotate180(warped)
        squares = ge.split_board(warped)
        to_check = None if rec["to_check_bits"] is None else R.unbits(rec["to_check_bits"])
        results, visual = det.detect_all_pieces(squares, use_delta=True, squares_to_check=to_check)
        yield rec, e, warped, results, visual, det
        if "state_after_update_references" in rec:
            det.update_references(squares)
            assert R.detector_state(det) == rec["state_after_update_references"]


@pytest.mark.parametrize("run_idx", [0, 1, 2])
def test_composed_chain_is_the_references(oracle, run_idx):
    fx = R.load_json("ref_chain_sequence.json")
    for rec, e, warped, results, visual, det in chain_reference(oracle, fx["runs"][run_idx]):
        assert R.sha(e) == rec["enhanced_sha256"] and R.sha(warped) == rec["warped_sha256"], rec["i"]
        assert R.result_rows(results) == rec["results"], rec["i"]
        assert R.bits(visual) == rec["visual_bits"] and R.detector_state(det) == rec["state"], rec["i"]
        assert R.bits(p for p, r in results.items() if r["has_piece"]) == rec["occupied_bits"]


def test_cython_twins_agree_with_the_python_classes():
    """The reference's selector prefers its Cython twins when they are built (frame_enhancer.py:12-21,
    change_detector.py:11-19).  They were compiled from the reference's own sources (oracle/build_ref_cython.sh ->
    oracle/_ref/) and driven on the same inputs when the fixtures were recorded: every output identical, so the
    fixtures above are the twins' outputs as well."""
    rep = R.load_json("ref_cython_twins.json")
    assert rep["built"]
    fe = rep["frame_enhancer"]
    assert fe["outputs"] == fe["identical"] == 57 and fe["differences"] == [] and fe["sharpen_kernel_dtype"] == "float32"
    for run in rep["change_detector"]:
        assert run["frames"] == run["identical"] and run["first_difference"] is None, run
