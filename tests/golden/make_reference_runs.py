"""Records what the REFERENCE'S OWN CLASSES return on committed synthetic inputs.

Runs only in the build container (needs /root/reference, which never travels):
the reference modules are imported from there with `cv2` bound to
tests/golden/cv2_oracle_shim.py (opencv-python is not installed in this image),
the real classes are driven exactly as the reference's callers drive them, and
their outputs are written as data files under tests/golden/:

  ref_enhancer.npz / .json     ImageEnhancerPython: apply_color_profile (shipped, radical, partial, clipping
                               profiles), every stage, process_pipeline, prepare_analysis  (frame_enhancer.py:56-181)
  ref_warp.npz                 board_detection.warp_image                                   (board_detection.py:61-71)
  ref_piece_sequence.json      PieceDetector.detect_all_pieces over a stream, driven like GameSession.on_frame
                               (warp -> split_board -> detect_all_pieces with squares_to_check, a mid-stream
                               update_references, use_smoothing / use_delta variants)        (piece_detector.py:348-453)
  ref_piece_shapes.npz / .json hand-made squares through detect_piece: every branch (std prefilter, hough, tower_top,
                               center_diff, symmetry, nothing), three radius settings, hough_param1/2 via getattr
  ref_chain_sequence.json      process_pipeline -> warp_image -> split_board -> detect_all_pieces (the composed chain)
  ref_change_sequence.json     ChangeDetectorPython with the attributes calibrate_sensitivity.py:135-139 writes:
                               calibrate / update_all_references / detect_changes_detailed / detect_changes /
                               classify_hand_pattern, focus squares                          (change_detector.py:36-201)
  ref_change_planes.npz        a few mean / variance planes of that run (all planes are covered by sha256)
  ref_refine_grid.json         SmartGridExtractor.refine_grid                                (grid_extractor.py:66-121)
  ref_cython_twins.json        ImageEnhancerCython / ChangeDetectorCython (src/cython/*.pyx, built by
                               oracle/build_ref_cython.sh into a scratch directory) on the same inputs: identical or not

What these pin: see cv2_oracle_shim.py — the reference's numpy arithmetic and control flow exactly; the OpenCV-side
pixel numbers are the oracle's own (circular) and stay "parity unpinned".

Run:  python tests/golden/make_reference_runs.py
"""
import contextlib
import hashlib
import io
import json
import os
import shutil
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)

from tests.golden import cv2_oracle_shim as shim  # noqa: E402

try:  # a machine with opencv-python records REAL OpenCV numbers; this image has none (no network, nothing installed)
    import cv2 as _cv2_real
    CV2_KIND = "opencv-python " + _cv2_real.__version__
except ImportError:
    sys.modules["cv2"] = shim.as_module()
    CV2_KIND = "oracle shim (tests/golden/cv2_oracle_shim.py): OpenCV-side numbers are the oracle's own"

# the reference reads color_profile.json / piece_detector_settings.json from the cwd at construction
# (frame_enhancer.py:48, piece_detector.py:54): run inside a scratch directory holding copies of the shipped files
_SCRATCH = tempfile.mkdtemp(prefix="cbv_ref_")
for _f in ("color_profile.json", "piece_detector_settings.json"):
    shutil.copy(os.path.join(REF, _f), _SCRATCH)
os.chdir(_SCRATCH)
with contextlib.redirect_stdout(io.StringIO()) as _import_log:
    import board_detection
    import change_detector
    import frame_enhancer
    import grid_extractor
    import piece_detector
IMPORT_LOG = _import_log.getvalue()

from chessboard_vision_amd import synth as S  # noqa: E402
from helpers import oracle_frame, random_frame  # noqa: E402

W, H = 640, 480


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def plain(v):
    """JSON-able copy that keeps every float exactly (repr round-trips doubles; float32 widen exactly)."""
    if isinstance(v, dict):
        return {(("%d,%d" % k) if isinstance(k, tuple) else k): plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [plain(x) for x in v]
    if isinstance(v, (set, frozenset)):
        return sorted(plain(x) for x in v)
    if isinstance(v, (np.bool_, bool)):
        return bool(v)
    if isinstance(v, np.integer):
        return int(v)
    if isinstance(v, np.floating):
        return float(v)
    return v


def bits(positions):
    """{(file, rank)} -> 64-bit int, bit = rank * 8 + file (a1 = bit 0)."""
    m = 0
    for (f, r) in positions:
        m |= 1 << (r * 8 + f)
    return m


# ---------------------------------------------------------------------------------------------------------------
PROFILES = {
    "shipped": json.load(open(os.path.join(REF, "color_profile.json"))),
    "radical": {"hue_shift": 17, "sat_scale": 1.3, "val_scale": 0.8, "contrast": 0.9, "brightness": 12,
                "radical_mode": 1, "target_hue": 30, "hue_window": 26},
    "radical_wrap": {"radical_mode": 1, "target_hue": 175, "hue_window": 12, "hue_shift": 95},
    "partial_negative": {"hue_shift": -200},
    "clipping": {"hue_shift": 179.5, "sat_scale": 3.7, "val_scale": 0.31, "contrast": 2.2, "brightness": -140},
    "fractional": {"hue_shift": 0.5, "sat_scale": 0.999, "val_scale": 1.001, "contrast": 1.0, "brightness": 0.49},
}


def small_frames():
    hsv_sweep = np.zeros((72, 96, 3), np.uint8)  # every hue sector and saturation ramp, dark to bright
    yy, xx = np.mgrid[:72, :96]
    hsv_sweep[..., 0] = (xx * 255 // 95)
    hsv_sweep[..., 1] = (yy * 255 // 71)
    hsv_sweep[..., 2] = ((xx + yy) * 255 // 166)
    return {
        "smooth": random_frame(96, 72, 11),
        "noise": random_frame(96, 72, 12, smooth=False),
        "sweep": hsv_sweep,
        "scene_dim": oracle_frame(160, 120, "dim", stream_id=1, frame_idx=3),
        "odd": random_frame(37, 29, 13),
    }


def gold_enhancer():
    arrs, meta = {}, {"cv2": CV2_KIND, "import_log": IMPORT_LOG.strip().splitlines(), "profiles": PROFILES, "calls": {}}
    enh = quiet(frame_enhancer.ImageEnhancer)
    assert type(enh).__name__ == "ImageEnhancerPython"       # the selector fell back as frame_enhancer.py:19-21 says
    assert enh.profile == PROFILES["shipped"]                 # read from cwd
    meta["ctor"] = {"profile_from_cwd": enh.profile, "sharpen_kernel_dtype": str(enh.sharpen_kernel.dtype),
                    "sharpen_kernel": enh.sharpen_kernel.tolist(), "clahe": [enh.clahe.getClipLimit(), list(enh.clahe.getTilesGridSize())]}
    frames = small_frames()
    for fname, f in frames.items():
        arrs["in_" + fname] = f
        for pname, prof in PROFILES.items():
            enh.profile = prof
            arrs["profile_%s_%s" % (pname, fname)] = enh.apply_color_profile(f)
        enh.profile = {}
        assert enh.apply_color_profile(f) is f                # frame_enhancer.py:57-58
    for fname in ("smooth", "scene_dim", "odd"):
        f = frames[fname]
        arrs["lighting_" + fname] = enh.correct_lighting(f)
        arrs["noise_" + fname] = enh.reduce_noise(f)
        arrs["sharpen_" + fname] = enh.sharpen(f)
        arrs["normalize_" + fname] = enh.normalize_intensity(f)
        g, b = enh.prepare_analysis(f)
        arrs["gray_" + fname], arrs["binary_" + fname] = g, b
        for pname in ("shipped", "radical"):
            enh.profile = PROFILES[pname]
            del shim.CALLS[:]
            arrs["pipeline_%s_%s" % (pname, fname)] = enh.process_pipeline(f)
            meta["calls"]["pipeline_%s_%s" % (pname, fname)] = plain(shim.CALLS)
        enh.profile = {}
        arrs["pipeline_none_" + fname] = enh.process_pipeline(f)
    # configs[0]: one 640x480 frame through process_pipeline (hash only: the array is 0.9 MB of noise)
    big = oracle_frame(W, H, "dim", stream_id=0, frame_idx=0)
    enh.profile = PROFILES["shipped"]
    out = enh.process_pipeline(big)
    g, b = enh.prepare_analysis(out)
    meta["c1_640x480"] = {"input": {"scene": "dim", "stream_id": 0, "frame_idx": 0}, "input_sha256": sha(big),
                          "pipeline_sha256": sha(out), "gray_sha256": sha(g), "binary_sha256": sha(b),
                          "pipeline_sum": int(out.sum(dtype=np.int64))}
    np.savez_compressed(os.path.join(OUT, "ref_enhancer.npz"), **arrs)
    with open(os.path.join(OUT, "ref_enhancer.json"), "w") as f:
        json.dump(plain(meta), f, indent=1)


def gold_warp():
    arrs = {}
    img = oracle_frame(W, H, "normal", stream_id=2, frame_idx=5)
    pts = S.scaled_corners(W, H)
    warped, M, bs = board_detection.warp_image(img, pts)
    assert bs == 620 and warped.shape == (620, 620, 3)
    arrs["pts_calib"], arrs["M_calib"] = pts, M
    arrs["sha_calib"] = np.frombuffer(bytes.fromhex(sha(warped)), np.uint8)
    arrs["rows_calib"] = warped[::40].copy()  # 16 rows kept in full for diagnosis
    # a small board (display_size/margin arguments) kept in full, from reordered integer corners
    small = random_frame(200, 150, 21)
    corners = np.array([[150, 130], [30, 20], [170, 25], [22, 120]], np.int32)
    ordered = board_detection.reorder(corners)
    w2, M2, bs2 = board_detection.warp_image(small, ordered, display_size=(300, 196), margin=100)
    assert bs2 == 96
    arrs["in_small"], arrs["corners_small"], arrs["ordered_small"], arrs["M_small"], arrs["warp_small"] = small, corners, ordered, M2, w2
    # partly outside the frame: BORDER_CONSTANT 0
    out_pts = np.float32([[-20, -10], [210, 5], [-5, 160], [190, 140]])
    w3, M3, _ = board_detection.warp_image(small, out_pts, display_size=(228, 400), margin=100)
    arrs["pts_outside"], arrs["M_outside"], arrs["warp_outside"] = out_pts, M3, w3
    np.savez_compressed(os.path.join(OUT, "ref_warp.npz"), **arrs)


# ---------------------------------------------------------------------------------------------------------------
def detector_state(pd_):
    keys = sorted(pd_.reference_squares.keys())
    return {"ref_keys_bits": bits(keys), "ref_sha256": sha(np.concatenate([pd_.reference_squares[k].ravel() for k in keys])) if keys else "",
            "cached_bits": bits(pd_.cached_results.keys()),
            "cached_has_bits": bits(k for k, v in pd_.cached_results.items() if v["has_piece"]),
            "history": {k: [bool(x) for x in v] for k, v in sorted(pd_.detection_history.items())}}


def result_rows(results):
    """per square, in dict order: [file, rank, has_piece, method, center, radius, confidence, center_border_diff]"""
    return [[p[0], p[1], bool(r["has_piece"]), r["method"], plain(r["center"]), plain(r["radius"]), plain(r["confidence"]),
             plain(r["center_border_diff"])] for p, r in results.items()]


PIECE_SCRIPT = {  # frame -> how detect_all_pieces is called (game_session.py:130-161 and the other signatures in use)
    "frames_per_ply": 3,
    "n": 26,
    "full_scan_every": 6,          # squares_to_check=None on those frames (the session's every-30th-frame scan)
    "update_references_after": 13,  # PieceDetector.update_references(squares) after that frame (game_session.py:219-223)
    "no_smoothing_frames": [17],
    "no_delta_frames": [19, 20],   # use_delta=False with a squares_to_check set: only forced squares are processed
}


def check_set_for(frame_idx, frames_per_ply):
    """A stand-in for the session's smart-scan set: the squares the scripted position occupies before this frame
    plus the destination squares of the next two plies (what python-chess's legal moves would include)."""
    ply = frame_idx // frames_per_ply
    s = set(S.position_after(ply).keys())
    for nxt in S.SCRIPT[ply % (len(S.SCRIPT) + 1):][:2]:
        for _, to in nxt:
            s.add(("abcdefgh".index(to[0]), int(to[1]) - 1))
    return s


def gold_piece_sequence():
    sc = PIECE_SCRIPT
    pd_ = quiet(piece_detector.PieceDetector)
    assert (pd_.min_radius_ratio, pd_.max_radius_ratio) == (0.25, 0.55)  # piece_detector_settings.json from the cwd
    ge = grid_extractor.SmartGridExtractor()
    ge.grid_lines_x, ge.grid_lines_y = list(S.CALIB_GRID_X), list(S.CALIB_GRID_Y)
    pts = S.scaled_corners(W, H)
    out = {"script": sc, "scene": "normal", "stream_id": 5, "size": [W, H], "settings": {"min_radius_ratio": 0.25, "max_radius_ratio": 0.55},
           "grid": "smart_calib", "frames": []}
    for i in range(sc["n"]):
        img = oracle_frame(W, H, "normal", stream_id=5, frame_idx=i, frames_per_ply=sc["frames_per_ply"])
        warped, _, _ = board_detection.warp_image(img, pts)        # the session warps the RAW frame (game_session.py:124)
        squares = ge.split_board(warped)
        to_check = None if i % sc["full_scan_every"] == 0 else check_set_for(i, sc["frames_per_ply"])
        kw = {"use_delta": True, "squares_to_check": to_check}
        if i in sc["no_smoothing_frames"]:
            kw["use_smoothing"] = False
        if i in sc["no_delta_frames"]:
            kw["use_delta"] = False
        results, visual = pd_.detect_all_pieces(squares, **kw)
        rec = {"i": i, "warped_sha256": sha(warped), "to_check_bits": None if to_check is None else bits(to_check),
               "kwargs": {k: v for k, v in kw.items() if k != "squares_to_check"},
               "results": result_rows(results), "visual_bits": bits(visual),
               "occupied_bits": bits(p for p, r in results.items() if r["has_piece"]), "state": detector_state(pd_)}
        if i == sc["update_references_after"]:
            pd_.update_references(squares)
            rec["state_after_update_references"] = detector_state(pd_)
        out["frames"].append(rec)
    # calibrate_reference (piece_detector.py:70-80) and get_occupied_squares (:442-445) on a fresh detector
    pd2 = quiet(piece_detector.PieceDetector)
    img = oracle_frame(W, H, "normal", stream_id=5, frame_idx=0, frames_per_ply=3)
    squares = ge.split_board(board_detection.warp_image(img, pts)[0])
    pd2.calibrate_reference(squares)
    out["calibrate_reference"] = {"state": detector_state(pd2), "cached": result_rows(pd2.cached_results)}
    out["get_occupied_squares_bits"] = bits(pd2.get_occupied_squares(squares))
    # detect_piece on single squares, gray input as well (piece_detector.py:126-130)
    singles = []
    for pos in [(0, 0), (4, 1), (3, 3), (7, 7), (2, 6)]:
        r = pd2.detect_piece(squares[pos], pos)
        g = shim.cvtColor(squares[pos], shim.COLOR_BGR2GRAY)
        rg = pd2.detect_piece(g)
        singles.append({"pos": list(pos), "bgr": result_rows({pos: r})[0], "gray_input": result_rows({pos: rg})[0]})
    out["detect_piece"] = singles
    with open(os.path.join(OUT, "ref_piece_sequence.json"), "w") as f:
        json.dump(plain(out), f)


def shape_squares():
    """Hand-made squares that reach every branch of detect_piece (piece_detector.py:272-345): uniform (std prefilter),
    discs (hough / tower_top), blocks and blobs (center_diff), rings and ramps (symmetry or nothing), noise."""
    rng = np.random.default_rng(99)
    out = {}
    def canvas(h, w, v):
        return np.full((h, w), v, np.float64)
    yy, xx = np.mgrid[:77, :77]
    d = np.sqrt((xx - 38) ** 2 + (yy - 38) ** 2)
    out["uniform"] = canvas(77, 77, 120)
    out["low_texture"] = canvas(77, 77, 120) + rng.integers(-10, 11, (77, 77))
    a = canvas(77, 77, 200); a[d <= 28] = 40; out["disc_dark"] = a
    a = canvas(77, 77, 60); a[d <= 24] = 230; out["disc_light"] = a
    a = canvas(77, 77, 200); a[d <= 11] = 30; out["disc_small"] = a
    a = canvas(77, 77, 200); a[20:58, 20:58] = 40; out["block"] = a
    a = canvas(77, 77, 90) + 120 * np.exp(-(d / 16.0) ** 2); out["blob"] = a
    a = canvas(77, 77, 90) + 70 * np.exp(-(d / 30.0) ** 2); out["soft_blob"] = a
    a = canvas(77, 77, 128) + 100 * np.cos(d / 6.0); out["rings"] = a
    a = canvas(77, 77, 0) + xx * 3.0; out["ramp"] = a
    a = canvas(77, 77, 200); a[:, 40:] = 30; out["half"] = a
    a = canvas(77, 77, 180); a[d <= 30] = 100; a[d <= 15] = 180; out["annulus"] = a
    out["noise"] = rng.integers(0, 256, (77, 77)).astype(np.float64)
    a = canvas(80, 76, 210); y2, x2 = np.mgrid[:80, :76]; a[np.sqrt((x2 - 30) ** 2 + (y2 - 45) ** 2) <= 20] = 50; out["disc_off_centre_80x76"] = a
    a = canvas(50, 50, 200); y3, x3 = np.mgrid[:50, :50]; a[np.sqrt((x3 - 25) ** 2 + (y3 - 25) ** 2) <= 15] = 40; out["disc_50x50"] = a
    res = {}
    for k, v in out.items():
        v = v + rng.integers(-3, 4, v.shape)
        g = np.clip(v, 0, 255).astype(np.uint8)
        res[k] = g
        bgr = np.stack([np.clip(g.astype(np.int16) + 10, 0, 255), g, np.clip(g.astype(np.int16) - 15, 0, 255)], axis=-1).astype(np.uint8)
        res[k + "_bgr"] = bgr
    return res


def gold_detect_piece_shapes():
    arrs = shape_squares()
    rows = []
    for ratios in ((0.20, 0.55), (0.12, 0.55), (0.25, 0.40)):
        pd_ = quiet(piece_detector.PieceDetector)
        pd_.min_radius_ratio, pd_.max_radius_ratio = ratios
        for name, img in arrs.items():
            r = pd_.detect_piece(img)
            rows.append({"name": name, "ratios": list(ratios), "result": result_rows({(0, 0): r})[0][2:]})
    pd_ = quiet(piece_detector.PieceDetector)
    pd_.hough_param2 = 60   # read through getattr (piece_detector.py:229): fewer circles, later branches decide
    pd_.hough_param1 = 180
    for name, img in arrs.items():
        r = pd_.detect_piece(img)
        rows.append({"name": name, "ratios": [pd_.min_radius_ratio, pd_.max_radius_ratio], "hough_param1": 180, "hough_param2": 60,
                     "result": result_rows({(0, 0): r})[0][2:]})
    np.savez_compressed(os.path.join(OUT, "ref_piece_shapes.npz"), **arrs)
    with open(os.path.join(OUT, "ref_piece_shapes.json"), "w") as f:
        json.dump(plain(rows), f)


def gold_chain_sequence():
    """process_pipeline -> warp_image -> split_board -> detect_all_pieces on the dim scene with the shipped profile
    (the composed north-star chain; the configuration of BoardPipeline in the -m gpu tests), linear grid + rotate 180
    in a second run."""
    out = {"size": [W, H], "runs": []}
    pts = S.scaled_corners(W, H)
    for name, scene, profile, grid, rot, n, fpp, check, upd in (
            ("dim_shipped_smart", "dim", PROFILES["shipped"], "smart", False, 12, 2, False, -1),
            ("dim_shipped_smart_checked", "dim", PROFILES["shipped"], "smart", False, 14, 2, True, 7),
            ("normal_none_linear_rot180", "normal", {}, "linear", True, 8, 2, False, -1)):
        enh = quiet(frame_enhancer.ImageEnhancer)
        enh.profile = profile
        pd_ = quiet(piece_detector.PieceDetector)
        pd_.min_radius_ratio, pd_.max_radius_ratio = 0.20, 0.55  # class defaults, as BoardPipeline.configure's
        if grid == "smart":
            ge = grid_extractor.SmartGridExtractor()
            ge.grid_lines_x, ge.grid_lines_y = list(S.CALIB_GRID_X), list(S.CALIB_GRID_Y)
        else:
            ge = grid_extractor.GridExtractor()
        frames = []
        for i in range(n):
            img = oracle_frame(W, H, scene, stream_id=3, frame_idx=i, frames_per_ply=fpp)
            e = enh.process_pipeline(img)
            warped, M, _ = board_detection.warp_image(e, pts)
            if rot:
                warped = shim.rotate(warped, shim.ROTATE_180)   # game_session.py:125-126
            squares = ge.split_board(warped)
            to_check = check_set_for(i, fpp) if (check and i % 6 != 0) else None
            results, visual = pd_.detect_all_pieces(squares, use_delta=True, squares_to_check=to_check)
            frames.append({"i": i, "enhanced_sha256": sha(e), "warped_sha256": sha(warped), "results": result_rows(results),
                           "to_check_bits": None if to_check is None else bits(to_check),
                           "visual_bits": bits(visual), "occupied_bits": bits(p for p, r in results.items() if r["has_piece"]),
                           "raw_bits": bits(p for p, r in pd_.cached_results.items() if r["has_piece"]), "state": detector_state(pd_)})
            if i == upd:
                pd_.update_references(squares)   # what the session does once it accepted a move (game_session.py:219-223)
                frames[-1]["state_after_update_references"] = detector_state(pd_)
        out["runs"].append({"name": name, "scene": scene, "profile": profile, "grid": grid, "rot180": rot, "stream_id": 3,
                            "frames_per_ply": fpp, "update_references_after": upd, "frames": frames})
    with open(os.path.join(OUT, "ref_chain_sequence.json"), "w") as f:
        json.dump(plain(out), f)


def gold_change_sequence():
    """ChangeDetectorPython driven like calibrate_sensitivity.main (:110-162): attributes written every frame,
    calibrate at a given frame, detect_changes_detailed + detect_changes + classify_hand_pattern per frame; plus
    update_all_references (EMA) and the focus API, which the tool does not call but the class offers."""
    runs = []
    planes = {}
    pts = S.scaled_corners(W, H)
    settings_file = json.load(open(os.path.join(REF, "sensitivity_settings.json")))
    configs = [
        dict(name="defaults", attrs={}, calibrate_at=1, n=10, ema_frames=[3, 4, 7], focus=None),
        dict(name="tool_settings", attrs=dict(z_threshold=settings_file["z_threshold"], initial_variance=settings_file["initial_variance"],
                                              alpha=settings_file["alpha"], blur_kernel=settings_file["blur_kernel"]),
             calibrate_at=2, n=10, ema_frames=[4, 5, 6, 8], focus=None),
        dict(name="even_kernel_and_focus", attrs=dict(z_threshold=1.45, initial_variance=50, alpha=0.37, blur_kernel=4),
             calibrate_at=0, n=9, ema_frames=[2, 3, 5], focus=[(4, 1), (4, 3), (6, 0), (5, 2), (0, 0)]),
    ]
    for cfg in configs:
        with contextlib.redirect_stdout(io.StringIO()):
            cd = change_detector.ChangeDetector()
        assert type(cd).__name__ == "ChangeDetectorPython"
        ge = grid_extractor.GridExtractor()
        frames = []
        for i in range(cfg["n"]):
            for k, v in cfg["attrs"].items():          # calibrate_sensitivity.py:135-139
                setattr(cd, k, v)
            if "blur_kernel" in cfg["attrs"]:
                cd._kernel = max(1, cfg["attrs"]["blur_kernel"] | 1)
            img = oracle_frame(W, H, "normal", stream_id=7, frame_idx=i, frames_per_ply=2)
            warped, _, _ = board_detection.warp_image(img, pts)
            squares = ge.split_board(warped)
            rec = {"i": i}
            if i == cfg["calibrate_at"]:
                cd.calibrate(squares)
                rec["calibrated"] = True
            if cfg["focus"] is not None and i == 4:
                cd.set_focus_squares(cfg["focus"])
                rec["focus_set"] = True
            if cfg["focus"] is not None and i == 7:
                cd.clear_focus()
                rec["focus_cleared"] = True
            rec["focus_count"] = cd.get_focus_count()
            detailed = cd.detect_changes_detailed(squares) if cd.is_calibrated else {}
            changes = cd.detect_changes(squares) if cd.is_calibrated else {}
            rec["detailed"] = [[p[0], p[1], v["z_score"], v["pct_changed"], v["intensity"], bool(v["is_circular"]), v["center_ratio"]]
                               for p, v in detailed.items()]
            rec["changes"] = [[p[0], p[1], v] for p, v in changes.items()]
            pat = cd.classify_hand_pattern(detailed) if detailed else {}
            rec["pattern"] = {"is_hand": pat["is_hand"], "is_move": pat["is_move"], "move_candidates": sorted(pat["move_candidates"])} if pat else {}
            if i in cfg["ema_frames"]:
                cd.update_all_references(squares)
                rec["ema"] = True
            if cd.is_calibrated:
                keys = sorted(cd.means.keys())
                rec["means_sha256"] = sha(np.concatenate([cd.means[k].ravel() for k in keys]))
                rec["vars_sha256"] = sha(np.concatenate([cd.variances[k].ravel() for k in keys]))
                assert all(cd.means[k].dtype == np.float32 and cd.variances[k].dtype == np.float32 for k in keys)
            frames.append(rec)
        for pos in [(4, 1), (4, 3), (0, 7)]:
            planes["%s_mean_%d_%d" % (cfg["name"], pos[0], pos[1])] = cd.means[pos]
            planes["%s_var_%d_%d" % (cfg["name"], pos[0], pos[1])] = cd.variances[pos]
        runs.append({"name": cfg["name"], "attrs": cfg["attrs"], "calibrate_at": cfg["calibrate_at"], "ema_frames": cfg["ema_frames"],
                     "focus": cfg["focus"], "scene": "normal", "stream_id": 7, "frames_per_ply": 2, "frames": frames})
    # update_all_references before any calibrate == calibrate (change_detector.py:69-71)
    with contextlib.redirect_stdout(io.StringIO()):
        cd = change_detector.ChangeDetector()
    img = oracle_frame(W, H, "normal", stream_id=7, frame_idx=0, frames_per_ply=2)
    squares = grid_extractor.GridExtractor().split_board(board_detection.warp_image(img, pts)[0])
    cd.update_all_references(squares)
    keys = sorted(cd.means.keys())
    first = {"is_calibrated": cd.is_calibrated, "means_sha256": sha(np.concatenate([cd.means[k].ravel() for k in keys])),
             "vars_sha256": sha(np.concatenate([cd.variances[k].ravel() for k in keys]))}
    # the reference's own regression test, under the shim (test_change_detector_regression.py:31-54)
    with contextlib.redirect_stdout(io.StringIO()):
        cd = change_detector.ChangeDetector()
    sq = {(c, r): np.zeros((50, 50), np.uint8) for r in range(8) for c in range(8)}
    cd.calibrate(sq)
    sq[(3, 3)] = np.full((50, 50), 255, np.uint8)
    ch = cd.detect_changes(sq)
    det = cd.detect_changes_detailed(sq)
    assert (3, 3) in ch and ch[(3, 3)] > 50.0 and det[(3, 3)]["intensity"] == "TOTAL"
    regression = {"changes": [[p[0], p[1], v] for p, v in ch.items()],
                  "detailed": [[p[0], p[1], v["z_score"], v["pct_changed"], v["intensity"], bool(v["is_circular"])] for p, v in det.items()]}
    with open(os.path.join(OUT, "ref_change_sequence.json"), "w") as f:
        json.dump(plain({"size": [W, H], "runs": runs, "update_before_calibrate": first, "regression_case": regression}), f)
    np.savez_compressed(os.path.join(OUT, "ref_change_planes.npz"), **planes)


def gold_cython_twins():
    """The reference's Cython twins (src/cython/*.pyx: the slot its selector fills first, frame_enhancer.py:12-21),
    compiled by oracle/build_ref_cython.sh into a scratch directory, driven on the same inputs as the Python classes above.
    Records whether every output is identical; the HIP classes are compared with the Python classes' fixtures, so
    this is what makes those fixtures speak for the Cython twins too."""
    # built into the generator's scratch directory (outside the repository, removed with it): compiled reference code
    # never sits in the tree that travels to the GPU box
    import subprocess
    ref_dir = os.path.join(_SCRATCH, "cython_twins")
    out = {"built": False}
    try:
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref_cython.sh"), REF, ref_dir], stdout=subprocess.DEVNULL)
    except (subprocess.CalledProcessError, OSError):
        with open(os.path.join(OUT, "ref_cython_twins.json"), "w") as f:
            json.dump(out, f)
        return
    sys.path.insert(0, ref_dir)
    with contextlib.redirect_stdout(io.StringIO()):
        from src.cython.frame_enhancer_cython import ImageEnhancerCython
        from src.cython.change_detector_cython import ChangeDetectorCython
    out["built"] = True
    # ImageEnhancerCython vs the recorded ImageEnhancerPython outputs
    fx = np.load(os.path.join(OUT, "ref_enhancer.npz"))
    enh = quiet(ImageEnhancerCython)
    assert enh.profile == PROFILES["shipped"]
    n = same = 0
    diffs = []
    def chk(name, got):
        nonlocal n, same
        n += 1
        if np.array_equal(got, fx[name]):
            same += 1
        else:
            diffs.append([name, int(np.abs(got.astype(np.int16) - fx[name].astype(np.int16)).max())])
    for fname in ("smooth", "noise", "sweep", "scene_dim", "odd"):
        for pname, prof in PROFILES.items():
            enh.profile = prof
            chk("profile_%s_%s" % (pname, fname), enh.apply_color_profile(fx["in_" + fname]))
    for fname in ("smooth", "scene_dim", "odd"):
        f = fx["in_" + fname]
        chk("lighting_" + fname, enh.correct_lighting(f))
        chk("noise_" + fname, enh.reduce_noise(f))
        chk("sharpen_" + fname, enh.sharpen(f))
        chk("normalize_" + fname, enh.normalize_intensity(f))
        g, b = enh.prepare_analysis(f)
        chk("gray_" + fname, g)
        chk("binary_" + fname, b)
        for pname in ("shipped", "radical", "none"):
            enh.profile = PROFILES[pname] if pname != "none" else {}
            chk("pipeline_%s_%s" % (pname, fname), enh.process_pipeline(f))
    out["frame_enhancer"] = {"outputs": n, "identical": same, "differences": diffs,
                             "sharpen_kernel_dtype": str(np.asarray(enh.sharpen_kernel).dtype)}
    # ChangeDetectorCython vs the recorded ChangeDetectorPython runs
    rec = json.load(open(os.path.join(OUT, "ref_change_sequence.json")))
    pts = S.scaled_corners(W, H)
    cd_report = []
    for run in rec["runs"]:
        cd = quiet(ChangeDetectorCython)
        ge = grid_extractor.GridExtractor()
        frames_same = 0
        first_diff = None
        for fr in run["frames"]:
            i = fr["i"]
            for k, v in run["attrs"].items():
                setattr(cd, k, v)
            img = oracle_frame(W, H, run["scene"], stream_id=run["stream_id"], frame_idx=i, frames_per_ply=run["frames_per_ply"])
            squares = ge.split_board(board_detection.warp_image(img, pts)[0])
            if fr.get("calibrated"):
                cd.calibrate(squares)
            if fr.get("focus_set"):
                cd.set_focus_squares([tuple(p) for p in run["focus"]])
            if fr.get("focus_cleared"):
                cd.clear_focus()
            detailed = cd.detect_changes_detailed(squares) if cd.is_calibrated else {}
            got = plain([[p[0], p[1], v["z_score"], v["pct_changed"], v["intensity"], bool(v["is_circular"]), v["center_ratio"]]
                         for p, v in detailed.items()])
            if fr.get("ema"):
                cd.update_all_references(squares)
            ok = got == fr["detailed"]
            if cd.is_calibrated:
                keys = sorted(cd.means.keys())
                ok = ok and sha(np.concatenate([cd.means[k].ravel() for k in keys])) == fr["means_sha256"] \
                    and sha(np.concatenate([cd.variances[k].ravel() for k in keys])) == fr["vars_sha256"]
            frames_same += ok
            if not ok and first_diff is None:
                first_diff = i
        cd_report.append({"run": run["name"], "frames": len(run["frames"]), "identical": frames_same, "first_difference": first_diff})
    out["change_detector"] = cd_report
    with open(os.path.join(OUT, "ref_cython_twins.json"), "w") as f:
        json.dump(out, f, indent=1)


def gold_refine_grid():
    out = []
    pts = S.scaled_corners(W, H)
    for scene, idx in (("normal", 0), ("normal", 9), ("dim", 4)):
        img = oracle_frame(W, H, scene, stream_id=9, frame_idx=idx, frames_per_ply=2)
        warped, _, _ = board_detection.warp_image(img, pts)
        ge = grid_extractor.SmartGridExtractor()
        gx, gy = ge.refine_grid(warped)
        rois = ge.split_board(warped)
        out.append({"scene": scene, "stream_id": 9, "frame_idx": idx, "frames_per_ply": 2, "warped_sha256": sha(warped),
                    "grid_x": plain(gx), "grid_y": plain(gy), "n_squares": len(rois)})
    with open(os.path.join(OUT, "ref_refine_grid.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    gold_enhancer()
    gold_warp()
    gold_piece_sequence()
    gold_detect_piece_shapes()
    gold_chain_sequence()
    gold_change_sequence()
    gold_refine_grid()
    gold_cython_twins()
    shutil.rmtree(_SCRATCH, ignore_errors=True)
    print("reference-run fixtures written to", OUT)
