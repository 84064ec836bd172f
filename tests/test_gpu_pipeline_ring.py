"""Ring discipline of BoardPipeline: slot ranges that are re-used while older runs are still in flight (three and
four partitions, as bench.py drives them), the ingest ring with more than two partitions, failed reconfiguration,
and BASELINE.json configs[2] at its full size (1080p, 512 frames in flight, the bench's chunk / lane / split)."""
import numpy as np
import pytest

from chessboard_vision_amd import synth as S
from helpers import oracle_frame

pytestmark = pytest.mark.gpu


def _tuples(res):
    return [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed, r.changed, r.parcial, r.total, r.circular) for r in res]


@pytest.mark.parametrize("parts", [3, 4])
def test_partition_ring_without_collecting_between_runs(gpu_ctx, parts):
    """run A, B, C(, D), A, B, ... back to back with nothing collected in between: a run re-uses slots whose scan is
    several runs back and may still be queued.  Equal to the same sequence with a full synchronisation after every run."""
    from chessboard_vision_amd.stream import BoardPipeline
    w, h, per = 640, 480, 6
    n = parts * per
    pts = S.scaled_corners(w, h)
    outs = []
    for sync_each in (True, False):
        p = BoardPipeline(w, h, n)
        p.configure(pts, profile=S.SHIPPED_PROFILE, chunk=3, lanes=2)
        p.synth(0, n, scene="dim", frames_per_ply=2)
        p.run(0, 1)
        p.calibrate_changes(0)
        p.reset_state()
        for rnd in range(4):
            for k in range(parts):
                p.run(k * per, per)
                if sync_each:
                    p.results(k * per, per)  # joins every stream and waits
        outs.append(_tuples(p.results(0, n)))
        p.close()
    assert outs[0] == outs[1]


def test_ingest_ring_three_partitions(gpu_ctx):
    """submit() into a partition whose last reader is THREE runs back (not the last run) waits for that reader."""
    from chessboard_vision_amd.stream import BoardPipeline
    w, h, per = 322, 241, 4
    n = 3 * per
    pts = S.scaled_corners(w, h)
    frames = [oracle_frame(w, h, "normal", frame_idx=i, frames_per_ply=2) for i in range(5 * per)]
    batches = [np.stack(frames[b * per:(b + 1) * per]) for b in range(5)]
    ref = BoardPipeline(w, h, n)
    ref.configure(pts, profile={}, chunk=2)
    want = []
    for b in range(5):
        s0 = (b % 3) * per
        for i in range(per):
            ref.upload(s0 + i, batches[b][i])
        ref.run(s0, per)
        want.append(_tuples(ref.results(s0, per)))
    a = BoardPipeline(w, h, n)
    a.configure(pts, profile={}, chunk=2)
    ring = a.host_ring()
    for b in range(3):
        ring[b * per:(b + 1) * per] = batches[b]
    a.submit(0, n)
    a.wait_submitted()                # the host ring is rewritten below: its copies must have left (runs are not waited for)
    for b in range(3):
        a.run(b * per, per)           # three runs in flight, nothing collected
    got = {}
    ring[0:per] = batches[3]
    a.submit(0, per)                  # reader of partition 0 is three runs back
    ring[per:2 * per] = batches[4]
    a.submit(per, per)
    got[2] = _tuples(a.results(2 * per, per))
    a.run(0, per)
    a.run(per, per)
    got[3] = _tuples(a.results(0, per))
    got[4] = _tuples(a.results(per, per))
    assert got[2] == want[2] and got[3] == want[3] and got[4] == want[4]
    assert np.array_equal(a.download(0, 1), batches[3][1]) and np.array_equal(a.download(0, per + 2), batches[4][2])


def test_failed_reconfigure_leaves_pipeline_unconfigured(gpu_ctx):
    """A configure that is rejected after an earlier good one must not leave a half-new configuration behind:
    run() then reports a state error instead of launching with stale descriptors."""
    from chessboard_vision_amd import _native as N
    from chessboard_vision_amd.stream import BoardPipeline
    w, h = 320, 240
    p = BoardPipeline(w, h, 2)
    pts = S.scaled_corners(w, h)
    p.configure(pts, profile={})
    p.synth(0, 2)
    p.run(0, 2)
    good = _tuples(p.results(0, 2))
    bad = N.PipelineConfig.from_buffer_copy(p._cfg)
    bad.rois[5].w = 100000
    assert p.ctx.lib.cbv_pipeline_configure(p.h_, bad) != 0   # rejected before any state was touched
    p.reset_state()
    p.run(0, 2)
    assert _tuples(p.results(0, 2)) == good                     # the earlier configuration is intact
    bad2 = N.PipelineConfig.from_buffer_copy(p._cfg)
    bad2.hough.dp = -1.0                                        # found only after buffers were re-made
    assert p.ctx.lib.cbv_pipeline_configure(p.h_, bad2) != 0
    with pytest.raises(RuntimeError):
        p.run(0, 2)
    p.configure(pts, profile={})
    p.run(0, 2)
    assert _tuples(p.results(0, 2)) == good


def test_hough_second_pass_handles_more_maxima_than_the_first_keeps(gpu_ctx, oracle):
    """The first HoughCircles pass keeps 512 accumulator maxima per square; white noise has more (never a board
    square).  Such squares are redone by the second pass, sized for every possible maximum: the result equals the
    oracle's (which has no limit), no overflow flag, and the class raises nothing."""
    from chessboard_vision_amd import _native as N
    from chessboard_vision_amd._squares import GRAY, SquareSet
    from ref_logic import detect_circle_unified
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[:77, :77]
    raw = {0: rng.integers(0, 256, (77, 77), dtype=np.uint8), 1: (((xx // 2 + yy // 2) & 1) * 255).astype(np.uint8),
           2: (((xx + yy) & 1) * 255).astype(np.uint8), 3: rng.integers(0, 2, (77, 77), dtype=np.uint8) * 255,
           4: rng.integers(0, 256, (80, 76), dtype=np.uint8)}
    s = SquareSet()
    s.load(raw, 5)
    grays = {k: s.get(GRAY, k) for k in raw}
    many = 0
    for prm2 in (25, 15):
        hg = s.hough(param2=prm2)
        for k, g in grays.items():
            found, center, radius, kind, circles = detect_circle_unified(g, param2=prm2)
            r = hg[k]
            assert not (r.flags & N.HOUGH_OVERFLOW), (k, prm2, r.n_centres)
            many += r.n_centres > 512
            assert r.n_circles == len(circles) and bool(r.found) == found, (k, prm2, r.n_circles, len(circles))
            for j in range(min(len(circles), 6)):
                assert tuple(np.float32(r.circles[j][c]) for c in range(4)) == tuple(np.float32(v) for v in circles[j]), (k, prm2, j)
            if found:
                assert (int(r.cx), int(r.cy)) == center and int(r.r) == radius
    assert many >= 2  # the second pass really ran


def test_configs2_full_size_512_frames_in_flight(gpu_ctx, oracle):
    """BASELINE.json configs[2]: 1080p, 512 frames resident, driven exactly like bench.py (chunk 64, 2 lanes, a step =
    4 consecutive runs of 128 frames, ChangeDetector calibrated from frame 0, shipped profile / grid / detector
    settings).  Size-independent properties over all 512 frames + exact comparison of sampled frames:
      * raw occupancy == the scripted position of every frame (a ply every 32 frames);
      * identical results for (chunk, lanes, splits) = (64, 2, 4) [what bench.py runs], (64, 2, 2) and (32, 1, 1);
      * a second step (slots re-used, temporal state carried) is deterministic;
      * warped boards of sampled frames == the oracle chain on the same frames."""
    from chessboard_vision_amd.stream import BoardPipeline
    w, h, F = 1920, 1080, 512
    pts = S.scaled_corners(w, h)
    grid = (S.CALIB_GRID_X, S.CALIB_GRID_Y)
    outs, warped, second = [], {}, []
    for chunk, lanes, splits in ((64, 2, 4), (64, 2, 2), (32, 1, 1)):  # bench.py's own split first
        p = BoardPipeline(w, h, F)
        p.configure(pts, profile=S.SHIPPED_PROFILE, grid_lines=grid, chunk=chunk, lanes=lanes, **S.SHIPPED_DETECTOR)
        p.synth(0, F, stream_id=0, scene="dim")
        p.run(0, 1)
        p.calibrate_changes(0)
        p.reset_state()
        bounds = [(k * F) // splits for k in range(splits + 1)]
        for k in range(splits):
            p.run(bounds[k], bounds[k + 1] - bounds[k])
        res = p.results(0, F)
        outs.append(_tuples(res))
        if not warped:
            for i in (0, 63, 64, 300, 511):
                warped[i] = p.download(2, i)
            for i in range(F):
                assert p.occupied(res[i], stable=False) == set(S.position_for_frame(i).keys()), i
            # stable occupancy lags a move by at most history_size frames and equals the script elsewhere
            for i in range(F):
                if i % 32 >= 5:
                    assert p.occupied(res[i], stable=True) == set(S.position_for_frame(i).keys()), i
            for rep in range(2):  # two more steps over the same slots, as the timed loop of bench.py does
                for k in range(splits):
                    p.run(bounds[k], bounds[k + 1] - bounds[k])
                second.append(_tuples(p.results(0, F)))
        else:
            for i, wimg in warped.items():
                assert np.array_equal(p.download(2, i), wimg), i
        p.close()
    assert outs[0] == outs[1] == outs[2]
    assert second[0] == second[1]  # steady state: every later step sees the same stream from the same carried state
    for i in (63, 300):
        f = oracle_frame(w, h, "dim", stream_id=0, frame_idx=i)
        enh = oracle.process_pipeline(f, S.SHIPPED_PROFILE)
        assert np.array_equal(warped[i], oracle.warp_image(enh, pts)[0]), i


def test_pipeline_hough_second_pass_on_noise_frames(gpu_ctx, oracle):
    """White-noise frames through the whole pipeline: many squares have more accumulator maxima than the first
    HoughCircles pass keeps, the per-run second pass redoes them before the scan, and occupancy / circles still equal
    the oracle chain's."""
    from chessboard_vision_amd.grid_extractor import GridExtractor
    from chessboard_vision_amd.stream import BoardPipeline
    from helpers import random_frame
    from ref_logic import RefPieceDetector, detect_circle_unified
    w, h, n = 640, 480, 5
    pts = S.scaled_corners(w, h)
    frames = [random_frame(w, h, 100 + i, smooth=False) for i in range(n)]
    p = BoardPipeline(w, h, n)
    p.configure(pts, profile={}, chunk=2, lanes=2, use_hough=2, hough_param2=6)   # 2 = HoughCircles on every non-uniform square; a low
    # accumulator threshold makes nearly every local maximum a candidate centre
    for i, f in enumerate(frames):
        p.upload(i, f)
    p.run(0, 3)
    p.run(3, 2)
    res = p.results(0, n)
    det = RefPieceDetector(hough=dict(param2=6))
    big = 0
    for i, f in enumerate(frames):
        warped = oracle.warp_image(oracle.process_pipeline(f, {}), pts)[0]
        assert np.array_equal(p.download(2, i), warped)
        sq = GridExtractor().split_board(warped)
        ref, vis = det.detect_all_pieces(sq)
        assert p.occupied(res[i], stable=True) == {k for k, r in ref.items() if r["has_piece"]}, i
        assert p.occupied(res[i], stable=False) == {k for k, r in det.cached_results.items() if r["has_piece"]}, i
        hg = p.hough(i)
        for roi, (r, c) in enumerate(p.rois_rc):
            g = oracle.square_preprocess(sq[(c, 7 - r)], 5)
            if hg[roi].flags & 2:  # skipped: uniform square
                continue
            found, center, radius, kind, circles = detect_circle_unified(g, param2=6)
            assert hg[roi].flags == 0 and hg[roi].n_circles == len(circles) and bool(hg[roi].found) == found, (i, roi)
            big += hg[roi].n_centres > 512
    assert big > 0, "no square needed the second pass: the test does not exercise it"
