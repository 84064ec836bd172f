"""GPU parity of the composed, device-resident chain (stream.BoardPipeline):
synthetic frames -> enhance -> warp -> 64-square detect with temporal logic,
against the oracle chain on the same frames, plus the size-independent
properties used at BASELINE.json's full sizes."""
import numpy as np
import pytest

from chessboard_vision_amd import synth as S
from helpers import oracle_frame

pytestmark = pytest.mark.gpu

W, H = 640, 480


def _oracle_chain(oracle, frames, profile, pts, grid_lines=None):
    from chessboard_vision_amd.grid_extractor import GridExtractor, SmartGridExtractor
    from ref_logic import RefPieceDetector
    ge = GridExtractor()
    if grid_lines is not None:
        ge = SmartGridExtractor()
        ge.grid_lines_x, ge.grid_lines_y = list(grid_lines[0]), list(grid_lines[1])
    det = RefPieceDetector(hough={})
    out = []
    for f in frames:
        enh = oracle.process_pipeline(f, profile)
        warped, _, _ = oracle.warp_image(enh, pts)
        res, vis = det.detect_all_pieces(ge.split_board(warped))
        out.append(dict(enh=enh, warped=warped, stable={p for p, r in res.items() if r["has_piece"]},
                        raw={p for p, r in det.cached_results.items() if r["has_piece"]}, visual=set(vis)))
    return out


@pytest.mark.parametrize("scene,profile,grid", [("dim", S.SHIPPED_PROFILE, "smart"), ("normal", {}, "linear")])
def test_pipeline_matches_oracle_chain(gpu_ctx, oracle, scene, profile, grid):
    from chessboard_vision_amd.stream import BoardPipeline, bits_to_positions
    n = 12
    pts = S.scaled_corners(W, H)
    gl = (S.CALIB_GRID_X, S.CALIB_GRID_Y) if grid == "smart" else None
    p = BoardPipeline(W, H, n)
    p.configure(pts, profile=profile, grid_lines=gl, keep_enhanced=True, chunk=5, lanes=2)
    p.synth(0, n, stream_id=3, scene=scene, frames_per_ply=2)
    frames = [oracle_frame(W, H, scene, stream_id=3, frame_idx=i, frames_per_ply=2) for i in range(n)]
    for i in (0, 5, n - 1):
        assert np.array_equal(p.download(0, i), frames[i]), "synthetic frame %d differs" % i
    p.run(0, n)
    res = p.results(0, n)
    ref = _oracle_chain(oracle, frames, profile, pts, gl)
    for i in range(n):
        assert np.array_equal(p.download(1, i), ref[i]["enh"]), "enhanced frame %d" % i
        assert np.array_equal(p.download(2, i), ref[i]["warped"]), "warped board %d" % i
        assert bits_to_positions(res[i].stable_occupied, p.rois_rc) == ref[i]["stable"], i
        assert bits_to_positions(res[i].raw_occupied, p.rois_rc) == ref[i]["raw"], i
        assert bits_to_positions(res[i].visual_changes, p.rois_rc) == ref[i]["visual"], i
        # raw occupancy covers the scripted position of that frame; exactly so in the clean scene (in the dim,
        # noise-amplified scene HoughCircles also fires on a few empty squares, in the oracle and here alike)
        want = set(S.position_for_frame(i, 2).keys())
        assert ref[i]["raw"] >= want and (scene == "dim" or ref[i]["raw"] == want)


def test_fused_and_unfused_paths_and_lane_counts_agree(gpu_ctx):
    """normalize folded into the warp gather (bench path) == materialised enhanced frame; any chunk/lane split."""
    from chessboard_vision_amd.stream import BoardPipeline
    n = 9
    pts = S.scaled_corners(W, H)
    outs = []
    for keep, chunk, lanes in ((True, 4, 1), (False, 4, 1), (False, 2, 3), (False, 9, 2)):
        p = BoardPipeline(W, H, n)
        p.configure(pts, profile=S.SHIPPED_PROFILE, keep_enhanced=keep, chunk=chunk, lanes=lanes)
        p.synth(0, n, scene="dim", frames_per_ply=2)
        p.run(0, n)
        res = p.results(0, n)
        outs.append(([p.download(2, i) for i in range(n)], [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed) for r in res]))
        p.close()
    for o in outs[1:]:
        for a, b in zip(outs[0][0], o[0]):
            assert np.array_equal(a, b)
        assert outs[0][1] == o[1]


def test_state_carries_across_runs_and_resets(gpu_ctx):
    from chessboard_vision_amd.stream import BoardPipeline
    n = 10
    pts = S.scaled_corners(W, H)
    p = BoardPipeline(W, H, n)
    p.configure(pts, profile={}, chunk=3)
    p.synth(0, n, scene="normal", frames_per_ply=2)
    p.run(0, n)
    whole = [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed) for r in p.results(0, n)]
    p.reset_state()
    p.run(0, 4)
    p.run(4, 6)  # temporal state (references, cache, history) continues from the previous call
    split = [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed) for r in p.results(0, n)]
    assert whole == split
    assert whole[0][3] == 0xFFFFFFFFFFFFFFFF and whole[0][2] == 0xFFFFFFFFFFFFFFFF  # first frame: no reference yet


def test_frame_at_a_time_equals_one_batched_run(gpu_ctx):
    """The live-camera way of calling (one or two frames per run: scan on the caller's stream, small-batch launch
    geometries of CLAHE apply / bilateral / squares, pack + NoiseHandler in one launch) gives what ONE run over all the
    frames gives: warped boards, per-frame results, NoiseHandler outputs, HoughCircles records."""
    from chessboard_vision_amd.stream import BoardPipeline
    n = 24
    pts = S.scaled_corners(W, H)
    p = BoardPipeline(W, H, n)
    p.configure(pts, profile=S.SHIPPED_PROFILE, chunk=8, lanes=2, **S.SHIPPED_DETECTOR)
    p.synth(0, n, scene="dim", frames_per_ply=3)

    def snapshot():
        res = [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed) for r in p.results(0, n)]
        hough = [[(h.flags, h.found, h.cx, h.cy, h.r) for h in p.hough(i)] for i in range(n)]
        return res, p.noise_results(0, n), [p.download(2, i) for i in range(n)], hough

    p.run(0, n)
    whole = snapshot()
    for sizes in ((1,) * n, (2, 1, 4, 1, 5, 3, 2, 6)):
        assert sum(sizes) == n
        p.reset_state()
        s0 = 0
        for c in sizes:
            p.run(s0, c)
            if c == 1:
                p.results(s0, 1)  # a live caller reads every frame's result before the next one arrives
            s0 += c
        got = snapshot()
        assert got[0] == whole[0]
        assert got[1] == whole[1]
        for a, b in zip(got[2], whole[2]):
            assert np.array_equal(a, b)
        assert got[3] == whole[3]
    p.close()


def test_two_frame_run_with_one_frame_chunks_joins_its_second_lane(gpu_ctx):
    """chunk = 1 with two lanes puts frame s + 1 of run(s, 2) on the second lane while the scan stays on the caller's
    stream (runs of <= 2 frames): the scan, a following submit / configure and results() must all be ordered after
    that lane.  Compared with chunk = 2 on one lane, repeatedly (an unjoined lane shows as a differing or changing
    result), then run(0, 2) followed at once by a reconfigure that frees the lanes' scratch."""
    from chessboard_vision_amd.stream import BoardPipeline
    n = 12
    pts = S.scaled_corners(W, H)

    def drive(chunk, lanes):
        p = BoardPipeline(W, H, n)
        p.configure(pts, profile=S.SHIPPED_PROFILE, chunk=chunk, lanes=lanes, **S.SHIPPED_DETECTOR)
        p.synth(0, n, scene="dim", frames_per_ply=2)
        outs = []
        for rep in range(3):
            p.reset_state()
            for s0 in range(0, n, 2):
                p.run(s0, 2)
            res = [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed) for r in p.results(0, n)]
            hough = [[(h.flags, h.found, h.cx, h.cy, h.r) for h in p.hough(i)] for i in range(n)]
            outs.append((res, hough, [p.download(2, i) for i in range(n)]))
        return p, outs

    p1, want = drive(2, 1)
    p2, got = drive(1, 2)
    for o in want[1:] + got:
        assert o[0] == want[0][0] and o[1] == want[0][1]
        for a, b in zip(o[2], want[0][2]):
            assert np.array_equal(a, b)
    # run + immediate reconfigure (frees A / B of both lanes) + run again: no fault, same answer
    p2.reset_state()
    p2.run(0, 2)
    p2.configure(pts, profile=S.SHIPPED_PROFILE, chunk=1, lanes=2, **S.SHIPPED_DETECTOR)
    p2.run(0, 2)
    res = [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed) for r in p2.results(0, 2)]
    assert res == want[0][0][:2]
    p1.close()
    p2.close()


def test_upload_path_equals_synth_path(gpu_ctx):
    from chessboard_vision_amd.stream import BoardPipeline
    pts = S.scaled_corners(W, H)
    a, b = BoardPipeline(W, H, 2), BoardPipeline(W, H, 2)
    for p in (a, b):
        p.configure(pts, profile=S.SHIPPED_PROFILE)
    a.synth(0, 2, scene="dim")
    for i in range(2):
        b.upload(i, oracle_frame(W, H, "dim", frame_idx=i))
    a.run(0, 2)
    b.run(0, 2)
    assert [r.stable_occupied for r in a.results(0, 2)] == [r.stable_occupied for r in b.results(0, 2)]
    assert np.array_equal(a.download(2, 1), b.download(2, 1))


def test_full_size_properties_1080p(gpu_ctx, oracle):
    """BASELINE.json full size (1080p stream): one frame exact against the oracle, then
    size-independent properties over the stream: determinism, occupancy = scripted position."""
    from chessboard_vision_amd.stream import BoardPipeline
    w, h, n = 1920, 1080, 40
    pts = S.scaled_corners(w, h)
    p = BoardPipeline(w, h, n)
    p.configure(pts, profile=S.SHIPPED_PROFILE, grid_lines=(S.CALIB_GRID_X, S.CALIB_GRID_Y), keep_enhanced=True, chunk=8)
    p.synth(0, n, scene="dim", frames_per_ply=4)
    p.run(0, n)
    res = p.results(0, n)
    f7 = oracle_frame(w, h, "dim", frame_idx=7, frames_per_ply=4)
    assert np.array_equal(p.download(0, 7), f7)
    enh = oracle.process_pipeline(f7, S.SHIPPED_PROFILE)
    assert np.array_equal(p.download(1, 7), enh)
    assert np.array_equal(p.download(2, 7), oracle.warp_image(enh, pts)[0])
    for i in range(n):
        assert p.occupied(res[i], stable=False) == set(S.position_for_frame(i, 4).keys()), i
    first = [(r.raw_occupied, r.stable_occupied) for r in res]
    p.reset_state()
    p.run(0, n)
    assert first == [(r.raw_occupied, r.stable_occupied) for r in p.results(0, n)]


def test_4k_frame_config4(gpu_ctx, oracle):
    """BASELINE.json configs[3]: 3840x2160, bilateral d = 9 (tile/halo stress)."""
    from chessboard_vision_amd.frame_enhancer import ImageEnhancer
    e = ImageEnhancer()
    e.profile = S.SHIPPED_PROFILE
    f = oracle_frame(3840, 2160, "dim")
    out = e.process_pipeline(f)
    assert np.array_equal(out, oracle.process_pipeline(f, S.SHIPPED_PROFILE))


@pytest.mark.parametrize("size", [(9, 7), (16, 16), (33, 5), (5, 40), (131, 67)])
def test_tiny_and_ragged_frames(gpu_ctx, oracle, size):
    """Images smaller than a tile / a CLAHE grid cell / the bilateral radius."""
    from chessboard_vision_amd.frame_enhancer import ImageEnhancer
    w, h = size
    rng = np.random.default_rng(w * 100 + h)
    f = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    e = ImageEnhancer()
    e.profile = S.SHIPPED_PROFILE
    assert np.array_equal(e.apply_color_profile(f), oracle.apply_color_profile(f, S.SHIPPED_PROFILE))
    assert np.array_equal(e.correct_lighting(f), oracle.correct_lighting(f))
    assert np.array_equal(e.reduce_noise(f), oracle.bilateral(f))
    assert np.array_equal(e.sharpen(f), oracle.filter3x3(f))
    assert np.array_equal(e.normalize_intensity(f), oracle.normalize_minmax(f))
    g, b = e.prepare_analysis(f)
    og, ob, _ = oracle.prepare_analysis(f)
    assert np.array_equal(g, og) and np.array_equal(b, ob)
    assert np.array_equal(e.process_pipeline(f), oracle.process_pipeline(f, S.SHIPPED_PROFILE))


def test_pipeline_change_detector_stage(gpu_ctx, oracle):
    """enhance -> warp -> change_detect + piece_detect in one device pass: the ChangeDetector stage
    (calibrate on an early frame, then detect_changes_detailed per frame) against the oracle-side
    restatement of change_detector.py on the same warped boards."""
    from chessboard_vision_amd.grid_extractor import GridExtractor
    from chessboard_vision_amd.stream import BoardPipeline
    from ref_logic import RefChangeDetector
    n = 10
    pts = S.scaled_corners(W, H)
    p = BoardPipeline(W, H, n)
    p.configure(pts, profile={}, keep_enhanced=True, chunk=4, z_threshold=2.55, initial_variance=600)
    p.synth(0, n, scene="normal", frames_per_ply=2)
    p.run(0, 1)
    p.calibrate_changes(0)
    p.reset_state()
    p.run(0, n)
    res = p.results(0, n)
    ref = RefChangeDetector(hough={})
    ref.z_threshold, ref.initial_variance = 2.55, 600
    seen = set()
    for i in range(n):
        f = oracle_frame(W, H, "normal", frame_idx=i, frames_per_ply=2)
        warped, _, _ = oracle.warp_image(oracle.process_pipeline(f, {}), pts)
        sq = GridExtractor().split_board(warped)
        if i == 0:
            ref.calibrate(sq)
        exp = ref.detect_changes_detailed(sq)
        got = p.changes_detailed(res[i], i)
        assert got == exp, (i, got, exp)
        seen |= {v["intensity"] for v in exp.values()}
    assert "TOTAL" in seen or "PARCIAL" in seen  # the scripted moves do change squares


def test_noise_handler_device_matches_reference_goldens(gpu_ctx):
    """cbv_noise_run (the k_noise kernel) against every step of the reference NoiseHandler's
    recorded sequences (tests/golden/noise_handler.json)."""
    import json
    import os
    from chessboard_vision_amd.noise_handler import run_on_device
    seqs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "noise_handler.json")))
    pos_to_index = {(f, r): (7 - r) * 8 + f for f in range(8) for r in range(8)}
    for steps in seqs:
        sets = [{tuple(x) for x in st["changed"]} for st in steps]
        got, _ = run_on_device(sets, pos_to_index)
        for i, ((state, data), st) in enumerate(zip(got, steps)):
            d = {k: (sorted(list(x) for x in v) if isinstance(v, set) else (list(v) if isinstance(v, tuple) else v)) for k, v in data.items()}
            assert state.name == st["state"] and d == st["data"], (i, state, d, st)


def test_pipeline_noise_stage_matches_host_class(gpu_ctx):
    """The pipeline's per-frame NoiseHandler outputs equal the host class fed with the same visual_changes."""
    from chessboard_vision_amd.noise_handler import NoiseHandler
    from chessboard_vision_amd.stream import BoardPipeline, bits_to_positions
    n = 48
    pts = S.scaled_corners(W, H)
    p = BoardPipeline(W, H, n)
    p.configure(pts, profile={}, chunk=16)
    p.synth(0, n, scene="normal", frames_per_ply=14)
    p.run(0, n)
    res, noise = p.results(0, n), p.noise_results(0, n)
    h = NoiseHandler()
    states = set()
    for i in range(n):
        exp = h.process(bits_to_positions(res[i].visual_changes, p.rois_rc))
        assert noise[i] == exp, (i, noise[i], exp)
        states.add(exp[0].name)
    assert {"NOISE_ACTIVE", "IDLE", "MOVE_PENDING"} <= states  # first frame = 64 changes (hand), then moves settle


def test_pipeline_hough_short_circuit_is_exact(gpu_ctx, oracle):
    """use_hough=1 runs HoughCircles only where centre-vs-border and radial symmetry left has_piece open;
    use_hough=2 runs it on every non-uniform square; use_hough=0 not at all.  1 and 2 must agree bit for bit
    (has_piece is an OR), and wherever the transform ran its outcome equals the oracle's."""
    from chessboard_vision_amd import _native as N
    from chessboard_vision_amd.stream import BoardPipeline
    from ref_logic import detect_circle_unified
    n = 6
    pts = S.scaled_corners(W, H)
    outs, ps = {}, {}
    for mode in (0, 1, 2):
        p = BoardPipeline(W, H, n)
        p.configure(pts, profile=S.SHIPPED_PROFILE, grid_lines=(S.CALIB_GRID_X, S.CALIB_GRID_Y), use_hough=mode, chunk=4)
        p.synth(0, n, scene="dim", frames_per_ply=2)
        p.run(0, n)
        outs[mode] = [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed, r.circular) for r in p.results(0, n)]
        ps[mode] = p
    assert outs[1] == outs[2]
    assert outs[0] != outs[1], "the dim scene is chosen so that HoughCircles changes some decisions"
    with pytest.raises(RuntimeError):
        ps[0].hough(0)
    warped = ps[2].download(2, 3)
    ran = {1: 0, 2: 0}
    for mode in (1, 2):
        hg = ps[mode].hough(3)
        for i, (r, c) in enumerate(ps[mode].rois_rc):
            if hg[i].flags & N.HOUGH_SKIPPED:
                continue
            ran[mode] += 1
            x0, y0, w, h = (ps[mode]._cfg.rois[i].x0, ps[mode]._cfg.rois[i].y0, ps[mode]._cfg.rois[i].w, ps[mode]._cfg.rois[i].h)
            gray = oracle.square_preprocess(warped[y0:y0 + h, x0:x0 + w], 5)
            found, center, radius, kind, circles = detect_circle_unified(gray)
            assert bool(hg[i].found) == found and hg[i].n_circles == len(circles), (mode, i)
            if found:
                assert (int(hg[i].cx), int(hg[i].cy), int(hg[i].r)) == (center[0], center[1], radius), (mode, i)
    assert 0 < ran[1] < ran[2] <= 64


def test_pipeline_occupancy_drives_game_state(gpu_ctx):
    """SURVEY f1 on top of the path: per-frame occupancy words -> GameState (native rules engine) -> moves and a
    FEN with piece identity, for the scripted game."""
    from chessboard_vision_amd import chess_rules as chess
    from chessboard_vision_amd.game_state import GameState
    from chessboard_vision_amd.stream import BoardPipeline
    fpp, n = 3, 3 * 16 + 2
    p = BoardPipeline(W, H, n)
    p.configure(S.scaled_corners(W, H), profile={}, chunk=16)
    p.synth(0, n, scene="normal", frames_per_ply=fpp)
    p.run(0, n)
    gs, played = GameState(), []
    for r in p.results(0, n):
        move, status = gs.process_occupancy_bits(chess.roi_bits_to_squares(r.raw_occupied))
        if move is not None:
            played.append((move.uci(), status))
    assert [m for m, _ in played] == ["e2e4", "e7e5", "g1f3", "b8c6", "f1b5", "a7a6", "b5a4", "g8f6", "e1g1", "f8e7",
                                      "f1e1", "b7b5", "a4b3", "d7d6", "c2c3", "e8g8"]
    assert [s for _, s in played].count("castling_confirmed") == 2
    assert gs.get_fen() == "r1bq1rk1/2p1bppp/p1np1n2/1p2p3/4P3/1BP2N2/PP1P1PPP/RNBQR1K1 w - - 1 9"


def test_ingest_ring_submit_run_overlap(gpu_ctx, oracle):
    """Ingest front end: frames written into the pinned host ring, copied asynchronously (submit) while the other
    half of the ring is processed (run); results equal those of synchronous uploads, in any interleaving."""
    from chessboard_vision_amd.stream import BoardPipeline
    w, h, half = 322, 241, 4  # odd size: frame stride is padded
    n = 2 * half
    pts = S.scaled_corners(w, h)
    frames = [oracle_frame(w, h, "normal", frame_idx=i, frames_per_ply=2) for i in range(3 * half)]
    ref = BoardPipeline(w, h, n)
    ref.configure(pts, profile={}, chunk=4)
    a = BoardPipeline(w, h, n)
    a.configure(pts, profile={}, chunk=4)
    ring = a.host_ring()
    assert ring.shape == (n, h, w, 3)
    want, got = [], []
    # batches of `half` frames alternate between the two halves of the ring
    batches = [frames[0:half], frames[half:2 * half], frames[2 * half:3 * half]]
    for b, fr in enumerate(batches):
        s0 = (b % 2) * half
        for i, f in enumerate(fr):
            ref.upload(s0 + i, f)
        ref.run(s0, half)
        want += [(r.raw_occupied, r.stable_occupied, r.visual_changes) for r in ref.results(s0, half)]
    ring[0:half] = np.stack(batches[0])
    a.submit(0, half)
    for b in range(3):
        s0 = (b % 2) * half
        if b + 1 < 3:  # next batch goes to the other half while this one is processed
            o0 = ((b + 1) % 2) * half
            ring[o0:o0 + half] = np.stack(batches[b + 1])   # safe: that half's previous run was collected below
            a.submit(o0, half)
        a.run(s0, half)
        got += [(r.raw_occupied, r.stable_occupied, r.visual_changes) for r in a.results(s0, half)]
        assert np.array_equal(a.download(0, s0 + 1), batches[b][1])
    assert got == want
    with pytest.raises(RuntimeError):
        a.submit(n - 1, 2)
    with pytest.raises(RuntimeError):
        ref.submit(0, 1)  # host ring never requested


def test_error_paths_are_loud(gpu_ctx):
    """Argument and state errors come back as CBV_ERR_* (RuntimeError in the wrappers) with a message; nothing is
    silently clamped or routed to a fallback."""
    from chessboard_vision_amd._squares import SquareSet
    from chessboard_vision_amd.stream import BoardPipeline
    p = BoardPipeline(W, H, 4)
    with pytest.raises(RuntimeError):
        p.run(0, 1)                                   # not configured
    pts = S.scaled_corners(W, H)
    with pytest.raises(RuntimeError, match="radius ratios"):
        p.configure(pts, max_radius_ratio=5.0)
    with pytest.raises(RuntimeError, match="history_size"):
        p.configure(pts, history_size=9)
    p.configure(pts)
    for bad in ((-1, 1), (0, 0), (3, 2)):
        with pytest.raises(RuntimeError, match="slot range"):
            p.run(*bad)
    with pytest.raises(RuntimeError):
        p.calibrate_changes(9)
    ss = SquareSet(gpu_ctx)
    with pytest.raises(RuntimeError, match="no squares"):
        ss.hough()
    with pytest.raises(RuntimeError):
        ss.load({0: np.zeros((300, 300), np.uint8)}, 5)   # beyond CBV_MAX_SQUARE_DIM
    ss.load({0: np.zeros((40, 40), np.uint8)}, 5)
    with pytest.raises(RuntimeError, match="radius ratios"):
        ss.hough(max_radius_ratio=4.5)
    with pytest.raises(RuntimeError, match="model requested"):
        ss.stats(use_model=True)


def test_stream_to_moves_end_to_end(gpu_ctx):
    """Whole loop on the synthetic stream: pipeline -> per-frame stable occupancy + NoiseHandler state (device) ->
    StableMoveTracker (20 stable frames, cooldown) -> GameState; the first six plies of the script come out."""
    from chessboard_vision_amd.game_state import GameState, StableMoveTracker
    from chessboard_vision_amd.stream import BoardPipeline
    fpp, plies = 26, 6
    n = fpp * plies + 24
    p = BoardPipeline(W, H, n)
    p.configure(S.scaled_corners(W, H), profile={}, chunk=25)
    p.synth(0, n, scene="normal", frames_per_ply=fpp)
    p.run(0, n)
    res, noise = p.results(0, n), p.noise_results(0, n)
    gs = GameState()
    t = [0.0]
    tracker = StableMoveTracker(gs, clock=lambda: t[0])
    moves = []
    for i in range(n):
        t[0] += 1.0 / 8.0   # an 8 fps camera: 26 frames a ply outlast the 2 s cooldown
        state, _ = noise[i]
        m = tracker.process(p.occupied(res[i]), noise_active=(state.name == "NOISE_ACTIVE"))
        if m is not None:
            moves.append((i, m.uci()))
    assert [u for _, u in moves] == ["e2e4", "e7e5", "g1f3", "b8c6", "f1b5", "a7a6"]
    # a move is accepted 20+ frames after its ply starts (stability) and before the next ply starts
    for k, (i, _) in enumerate(moves):
        assert (k + 1) * fpp + 19 <= i < (k + 2) * fpp + 24, (k, i)
    assert gs.get_fen().startswith("r1bqkbnr/1ppp1ppp/p1n5/1B2p3/4P3/5N2/PPPP1PPP/RNBQK2R w KQkq -")


def test_temporal_logic_on_random_board_sequences(gpu_ctx, oracle):
    """The scan kernel and the device NoiseHandler against the restated host logic on boards that do NOT follow a game:
    pieces appear and vanish at random, a 'hand' flips a dozen squares for a few frames, quiet stretches let the
    5-frame majority and the conditional reference refresh settle.  Every per-frame set must match."""
    from chessboard_vision_amd.grid_extractor import GridExtractor
    from chessboard_vision_amd.noise_handler import NoiseHandler
    from chessboard_vision_amd.stream import BoardPipeline, bits_to_positions
    from helpers import oracle_scene
    from ref_logic import RefPieceDetector
    rng = np.random.default_rng(2024)
    n = 36
    pts = S.scaled_corners(W, H)
    Hinv = oracle.get_perspective_transform(pts, S.BOARD_UNIT_QUAD)
    board = S.board_array(S.position_after(0)).copy()
    frames, boards = [], []
    for i in range(n):
        if i % 3 == 2:                                   # a few random squares change
            for sq in rng.integers(0, 64, int(rng.integers(1, 4))):
                board[sq] = 0 if board[sq] else int(rng.integers(1, 3))
        b = board.copy()
        if 14 <= i < 18:                                 # a hand: many squares look different for four frames
            for sq in rng.choice(64, 14, replace=False):
                b[sq] = 0 if b[sq] else 2
        boards.append(b)
        frames.append(oracle.synth_frame(S.frame_seed(5, i), W, H, Hinv, b, oracle_scene("normal")))
    p = BoardPipeline(W, H, n)
    p.configure(pts, profile={}, chunk=7, lanes=2, keep_enhanced=True)
    for i, f in enumerate(frames):
        p.upload(i, f)
    p.run(0, 20)                                         # two runs: the temporal state must carry over
    p.run(20, n - 20)
    res, noise = p.results(0, n), p.noise_results(0, n)
    det, nh, ge = RefPieceDetector(hough={}), NoiseHandler(), GridExtractor()
    for i in range(n):
        warped, _, _ = oracle.warp_image(oracle.process_pipeline(frames[i], {}), pts)
        r, vis = det.detect_all_pieces(ge.split_board(warped))
        want_stable = {pos for pos, info in r.items() if info["has_piece"]}
        want_raw = {pos for pos, info in det.cached_results.items() if info["has_piece"]}
        assert bits_to_positions(res[i].visual_changes, p.rois_rc) == set(vis), i
        assert bits_to_positions(res[i].raw_occupied, p.rois_rc) == want_raw, i
        assert bits_to_positions(res[i].stable_occupied, p.rois_rc) == want_stable, i
        state, data = nh.process(set(vis))
        assert noise[i][0] == state and noise[i][1] == data, (i, noise[i], (state, data))
    assert any(len(bits_to_positions(r.visual_changes, p.rois_rc)) > 8 for r in res[14:18]), "the hand must be visible"


def test_pipeline_squares_to_check(gpu_ctx, oracle):
    """detect_all_pieces(squares_to_check=...) on the device (game_session.py:130-160 passes such a set on 29 of 30
    frames): forced squares are processed although unchanged and cached, which also refreshes their reference."""
    from chessboard_vision_amd.grid_extractor import GridExtractor
    from chessboard_vision_amd.stream import BoardPipeline, bits_to_positions
    from ref_logic import RefPieceDetector
    rng = np.random.default_rng(77)
    n = 14
    pts = S.scaled_corners(W, H)
    frames = [oracle_frame(W, H, "normal", stream_id=2, frame_idx=i, frames_per_ply=3) for i in range(n)]
    all_pos = [(f, r) for f in range(8) for r in range(8)]
    sets = []
    for i in range(n):
        if i % 5 == 0:
            sets.append(None)                                          # the full scan frame: squares_to_check=None
        else:
            sets.append({all_pos[k] for k in rng.choice(64, int(rng.integers(0, 30)), replace=False)})
    p = BoardPipeline(W, H, n)
    p.configure(pts, profile={}, chunk=5)
    for i, f in enumerate(frames):
        p.upload(i, f)
    p.set_check_squares(0, [s if s is not None else set() for s in sets])
    p.run(0, n)
    res = p.results(0, n)
    det, ge = RefPieceDetector(hough={}), GridExtractor()
    forced_seen = 0
    for i in range(n):
        warped, _, _ = oracle.warp_image(oracle.process_pipeline(frames[i], {}), pts)
        r, vis = det.detect_all_pieces(ge.split_board(warped), squares_to_check=sets[i])
        assert bits_to_positions(res[i].visual_changes, p.rois_rc) == set(vis), i
        assert bits_to_positions(res[i].processed, p.rois_rc) == det.last_processed, i
        assert bits_to_positions(res[i].stable_occupied, p.rois_rc) == {pos for pos, info in r.items() if info["has_piece"]}, i
        forced_seen += len(det.last_processed - set(vis)) if i else 0
    assert forced_seen > 20, "forced squares must show up as processed without a visual change"
    # clearing the masks restores squares_to_check=None for every slot
    p.set_check_squares(0, None)
    p.reset_state()
    p.run(0, n)
    det2 = RefPieceDetector(hough={})
    for i in range(n):
        warped, _, _ = oracle.warp_image(oracle.process_pipeline(frames[i], {}), pts)
        det2.detect_all_pieces(ge.split_board(warped))
        assert bits_to_positions(p.results(i, 1)[0].processed, p.rois_rc) == det2.last_processed, i


def test_pipeline_update_references_mid_stream(gpu_ctx, oracle):
    """game_session.py:219-223 on the device: after frame k the references are replaced by frame k's squares and the
    cache is cleared (history kept), the noise handler is reset; the following frames must match the host logic doing
    the same."""
    from chessboard_vision_amd.grid_extractor import GridExtractor
    from chessboard_vision_amd.noise_handler import NoiseHandler
    from chessboard_vision_amd.stream import BoardPipeline, bits_to_positions
    from ref_logic import RefPieceDetector
    n, k = 16, 8
    pts = S.scaled_corners(W, H)
    frames = [oracle_frame(W, H, "normal", stream_id=4, frame_idx=i, frames_per_ply=3) for i in range(n)]
    p = BoardPipeline(W, H, n)
    p.configure(pts, profile={}, chunk=4)
    for i, f in enumerate(frames):
        p.upload(i, f)
    p.run(0, k + 1)
    p.update_references(k, reset_noise=True)
    p.run(k + 1, n - k - 1)
    res, noise = p.results(0, n), p.noise_results(0, n)
    det, nh, ge = RefPieceDetector(hough={}), NoiseHandler(), GridExtractor()
    for i in range(n):
        warped, _, _ = oracle.warp_image(oracle.process_pipeline(frames[i], {}), pts)
        sq = ge.split_board(warped)
        r, vis = det.detect_all_pieces(sq)
        state, data = nh.process(set(vis))
        assert bits_to_positions(res[i].visual_changes, p.rois_rc) == set(vis), i
        assert bits_to_positions(res[i].processed, p.rois_rc) == det.last_processed, i
        assert bits_to_positions(res[i].stable_occupied, p.rois_rc) == {pos for pos, info in r.items() if info["has_piece"]}, i
        assert noise[i] == (state, data), i
        if i == k:
            det.update_references(sq)
            nh.reset()
    assert len(bits_to_positions(res[k + 1].processed, p.rois_rc)) == 64, "cleared cache: everything is processed again"
