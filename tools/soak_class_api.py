"""One-off soak of the one-call class API (not part of the test suite): random boards (size, grid lines, content), random
squares_to_check / use_smoothing / use_delta, occasional update_references / calibrate_reference, random ChangeDetector
attributes, boards that are crops or padded views; PieceDetector / ChangeDetector on the GPU against the restated
reference logic on the oracle (tests/ref_logic.py), every result dict, visual_changes, cache, history, reference and model
planes after every call.

    python tools/soak_class_api.py [n_streams]      (GPU box)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "16")
from chessboard_vision_amd.change_detector import ChangeDetector  # noqa: E402
from chessboard_vision_amd.grid_extractor import GridExtractor, SmartGridExtractor  # noqa: E402
from chessboard_vision_amd.piece_detector import PieceDetector  # noqa: E402
from ref_logic import RefChangeDetector, RefPieceDetector  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fails = 0
t0 = time.time()


def board_image(rng, S, t, pieces):
    """A board-like image: checker squares + discs that come and go + noise."""
    img = np.empty((S, S, 3), np.uint8)
    q = S // 8
    yy, xx = np.mgrid[:S, :S]
    chk = ((yy // max(q, 1)) + (xx // max(q, 1))) & 1
    img[:] = np.where(chk[..., None] == 1, np.uint8(rng.integers(60, 120)), np.uint8(rng.integers(140, 200)))
    for (c, r), (col, rad) in pieces.items():
        cy, cx = r * q + q // 2 + int(rng.integers(-1, 2)), c * q + q // 2 + int(rng.integers(-1, 2))
        img[(yy - cy) ** 2 + (xx - cx) ** 2 <= rad * rad] = col
    noise = rng.integers(-6, 7, img.shape)
    return np.clip(img.astype(np.int16) + noise, 0, 255).astype(np.uint8)


for it in range(n):
    rng = np.random.default_rng(7000 + it)
    S = int(rng.integers(160, 700))
    q = S // 8
    if it % 2:
        ge = SmartGridExtractor()
        lines = [0] + sorted(int(i * S / 8 + rng.integers(-q // 6, q // 6 + 1)) for i in range(1, 8)) + [S]
        ge.grid_lines_x, ge.grid_lines_y = lines, list(lines)
    else:
        ge = GridExtractor()
    gpu, ref = PieceDetector(), RefPieceDetector(hough={})
    cg, cr = ChangeDetector(), RefChangeDetector(hough={})
    zt, iv, al, bk = float(rng.uniform(1.5, 4)), float(rng.uniform(20, 400)), float(rng.uniform(0.02, 0.3)), int(rng.integers(1, 12))
    for d in (cg, cr):
        d.z_threshold, d.initial_variance, d.alpha, d.blur_kernel = zt, iv, al, bk
    pieces = {(int(rng.integers(0, 8)), int(rng.integers(0, 8))): (tuple(int(v) for v in rng.integers(0, 256, 3)), int(rng.integers(q // 5, q // 2 + 1))) for _ in range(14)}
    variant = it % 4
    for t in range(10):
        if rng.random() < 0.5:  # something moves
            k = list(pieces.keys())[int(rng.integers(0, len(pieces)))]
            v = pieces.pop(k)
            pieces[(int(rng.integers(0, 8)), int(rng.integers(0, 8)))] = v
        board = board_image(rng, S, t, pieces)
        if variant == 1:  # a crop of a padded frame
            big = np.full((S + 20, S + 37, 3), 9, np.uint8)
            big[10:10 + S, 11:11 + S] = board
            board = big[10:10 + S, 11:11 + S]
        elif variant == 2:  # a wide pitch
            big = np.full((S + 4, 3 * S, 3), 9, np.uint8)
            big[2:2 + S, S:2 * S] = board
            board = big[2:2 + S, S:2 * S]
        sq = ge.split_board(board)
        if variant == 3:
            sq = dict(sq)
        tag = "it=%d S=%d t=%d variant=%d" % (it, S, t, variant)
        try:
            if t == 0 and it % 3 == 0:
                gpu.update_references(sq); ref.update_references(sq)
            if t == 5 and it % 5 == 0:
                gpu.calibrate_reference(sq); ref.calibrate_reference(sq)
            check = None if rng.random() < 0.3 else {(int(rng.integers(0, 8)), int(rng.integers(0, 8))) for _ in range(int(rng.integers(0, 40)))}
            kw = dict(squares_to_check=check, use_smoothing=bool(rng.random() < 0.8), use_delta=bool(rng.random() < 0.8))
            r1, v1 = gpu.detect_all_pieces(sq, **kw)
            r2, v2 = ref.detect_all_pieces(sq, **kw)
            ok = v1 == v2 and r1 == r2 and list(r1) == list(r2) and gpu.cached_results == ref.cached_results
            ok = ok and {k: list(v) for k, v in gpu.detection_history.items()} == {k: list(v) for k, v in ref.detection_history.items()}
            ok = ok and set(gpu.reference_squares.keys()) == set(ref.reference_squares.keys())
            ok = ok and all(np.array_equal(gpu.reference_squares[p], ref.reference_squares[p]) for p in ref.reference_squares)
            if not ok:
                fails += 1
                print("FAIL pieces", tag, kw, flush=True)
            if t == 1:
                cg.calibrate(sq); cr.calibrate(sq)
            if t >= 1:
                if t == 6:
                    fs = [(int(rng.integers(0, 8)), int(rng.integers(0, 8))) for _ in range(5)]
                    cg.set_focus_squares(fs); cr.focus_squares = set(fs)
                d1, d2 = cg.detect_changes_detailed(sq), cr.detect_changes_detailed(sq)
                if d1 != d2 or list(d1) != list(d2):
                    fails += 1
                    bad = [(p, d1.get(p), d2.get(p)) for p in (list(d2) + [k for k in d1 if k not in d2]) if d1.get(p) != d2.get(p)]
                    print("FAIL changes", tag, "blur", cg.blur_kernel, "keys", len(d1), len(d2), "order", list(d1) == list(d2), "first", bad[:1], flush=True)
                if t in (3, 7):
                    cg.update_all_references(sq); cr.update_all_references(sq)
                    if not all(np.array_equal(cg.means[p], cr.means[p]) and np.array_equal(cg.variances[p], cr.variances[p]) for p in sq):
                        fails += 1
                        print("FAIL model planes", tag, flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print("EXC", tag, type(e).__name__, e, flush=True)
            break
    if it % 10 == 0:
        print("stream %d / %d, %d failures, %.0f s" % (it, n, fails, time.time() - t0), flush=True)
print("soak_class_api: %d streams x 10 frames, %d failures, %.0f s" % (n, fails, time.time() - t0))
sys.exit(1 if fails else 0)
