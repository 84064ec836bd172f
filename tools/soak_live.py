"""One-off soak of the frame-at-a-time path (the launches a short run leaves out: no reset of the histogram / min-max words
between passes, byte map worked out by the consumer, second-pass counter zeroed by the warp, result records mirrored to
pinned memory): random pipelines, every stream processed once as ONE batched run and once in random pieces of 1-7
frames with `results` read after every piece (and other pipelines of the same context run in between).  Everything a
caller can read must agree: result records, NoiseHandler outputs, HoughCircles records, warped boards.

    python tools/soak_live.py [n_streams]      (GPU box)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chessboard_vision_amd import synth as S  # noqa: E402
from chessboard_vision_amd.stream import BoardPipeline  # noqa: E402

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fails, t0 = 0, time.time()
other = None
seen_occ, seen_vis = set(), 0
for it in range(n_streams):
    rng = np.random.default_rng(700000 + it)
    w, h = [(640, 480), (320, 240), (800, 600), (1280, 720)][int(rng.integers(0, 4))]
    n = int(rng.integers(9, 33))
    scene = ["dim", "normal", "white_noise"][int(rng.choice(3, p=[0.5, 0.4, 0.1]))]
    kw = dict(profile=S.SHIPPED_PROFILE if rng.random() < 0.6 else {}, chunk=int(rng.integers(1, 9)), lanes=int(rng.integers(1, 4)),
              keep_enhanced=bool(rng.random() < 0.3), rot180=bool(rng.random() < 0.3), use_hough=bool(rng.random() < 0.85),
              grid_lines=(S.CALIB_GRID_X, S.CALIB_GRID_Y) if rng.random() < 0.5 else None)
    kw["enhance_region"] = bool(rng.random() < 0.3) and not kw["keep_enhanced"]
    pts = S.scaled_corners(w, h)
    p = BoardPipeline(w, h, n)
    p.configure(pts, **kw, **({k: v for k, v in S.SHIPPED_DETECTOR.items() if k not in kw}))
    p.synth(0, n, stream_id=it, scene=scene, frames_per_ply=int(rng.integers(1, 5)))
    calibrated = rng.random() < 0.5

    def start():
        p.reset_state()
        if calibrated:
            p.run(0, 1)
            p.calibrate_changes(0)
            p.reset_state()

    def snapshot():
        res = [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed) for r in p.results(0, n)]
        hough = [[(x.flags, x.found, x.cx, x.cy, x.r) for x in p.hough(i)] for i in range(n)] if kw["use_hough"] else None
        return res, p.noise_results(0, n), [p.download(2, i) for i in range(n)], hough

    try:
        start()
        p.run(0, n)
        whole = snapshot()
        start()
        s0, live = 0, []
        while s0 < n:
            c = int(min(n - s0, rng.integers(1, 8)))
            p.run(s0, c)
            live += [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed) for r in p.results(s0, c)]
            if other is not None and rng.random() < 0.3:  # another camera on the same context in between
                other.run(0, int(rng.integers(1, 4)))
                other.results(0, 1)
            s0 += c
        got = snapshot()
        assert live == whole[0], "records read piece by piece differ from the batched run"
        assert got[0] == whole[0] and got[1] == whole[1] and got[3] == whole[3], "snapshot differs from the batched run"
        for a, b in zip(got[2], whole[2]):
            assert np.array_equal(a, b), "warped board differs"
        seen_occ.update(r[0] for r in whole[0])
        seen_vis += sum(1 for r in whole[0] if r[2])
    except (AssertionError, RuntimeError) as e:
        fails += 1
        print("stream", it, (w, h, n, scene), kw, "FAILED", str(e)[:300], flush=True)
    if other is not None:
        other.close()
    other = p  # stays alive as the next stream's neighbour
    if it % 20 == 19:
        print("stream %d / %d, %d failures, %.0f s" % (it + 1, n_streams, fails, time.time() - t0), flush=True)
print("soak_live: %d streams, %d failures, %.0f s (%d distinct raw occupancies, %d frames with visual changes)" % (n_streams, fails, time.time() - t0, len(seen_occ), seen_vis))
