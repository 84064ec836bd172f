"""Region-limited enhancement (cbv_pipeline_config.enhance_region): CLAHE apply, bilateral and sharpen run first on the
part of each frame the warp samples; the rest of the frame is processed only for frames whose region does not already
hold a 0 and a 255 after sharpen.  Every output must equal whole-frame enhancement, in both cases."""
import numpy as np
import pytest

from chessboard_vision_amd import synth as S

pytestmark = pytest.mark.gpu


def _snapshot(p, n):
    res = [(r.raw_occupied, r.stable_occupied, r.visual_changes, r.processed, r.changed, r.parcial, r.total) for r in p.results(0, n)]
    hough = [[(h.flags, h.found, h.cx, h.cy, h.r) for h in p.hough(i)] for i in range(n)]
    return res, [p.download(2, i) for i in range(n)], hough


def _run_both(w, h, n, fill, pts=None, chunk=0, **cfg):
    from chessboard_vision_amd.stream import BoardPipeline
    pts = pts if pts is not None else S.scaled_corners(w, h)
    out = []
    for region in (False, True):
        p = BoardPipeline(w, h, n)
        p.configure(pts, profile=S.SHIPPED_PROFILE, chunk=chunk, enhance_region=region, **cfg)
        fill(p)
        p.run(0, n)
        out.append(_snapshot(p, n))
        p.close()
    return out


def _assert_same(a, b):
    assert a[0] == b[0]
    for x, y in zip(a[1], b[1]):
        assert np.array_equal(x, y)
    assert a[2] == b[2]


@pytest.mark.parametrize("size,n,chunk", [((640, 480), 12, 5), ((1920, 1080), 6, 4), ((1920, 1080), 1, 0), ((317, 203), 3, 0)])
@pytest.mark.parametrize("scene", ["normal", "dim"])
def test_region_limited_equals_whole_frame_on_synthetic_streams(gpu_ctx, size, n, chunk, scene):
    w, h = size
    full, reg = _run_both(w, h, n, lambda p: p.synth(0, n, scene=scene, frames_per_ply=2), chunk=chunk, **S.SHIPPED_DETECTOR)
    _assert_same(full, reg)


def _flat_frame(w, h, value, rng, blocks):
    f = np.full((h, w, 3), value, np.uint8)
    for (x0, y0, x1, y1) in blocks:
        f[y0:y1, x0:x1] = rng.integers(0, 256, (y1 - y0, x1 - x0, 3), dtype=np.uint8)
    return f


@pytest.mark.parametrize("size", [(640, 480), (1920, 1080)])
def test_complement_pass_runs_when_the_region_does_not_saturate(gpu_ctx, size):
    """Frames whose board region is flat (min == max there) while the extremes of the sharpened frame lie OUTSIDE the
    region: normalize's LUT then depends on pixels only the complement pass sees.  Mixed with frames that do saturate
    inside the region in the same launch (the gate is per frame)."""
    w, h = size
    rng = np.random.default_rng(77)
    pts = S.scaled_corners(w, h)
    xs, ys = [p[0] for p in pts], [p[1] for p in pts]
    inside = (int(min(xs)) + 40, int(min(ys)) + 40, int(min(xs)) + 90, int(min(ys)) + 90)
    frames = [
        _flat_frame(w, h, 100, rng, [(0, 0, 30, 20)]),                      # extremes top-left, outside
        _flat_frame(w, h, 100, rng, [(w - 30, h - 18, w, h)]),              # bottom-right
        _flat_frame(w, h, 100, rng, [inside]),                              # extremes inside the region: gate closed
        _flat_frame(w, h, 37, rng, []),                                     # flat everywhere: max == min
        _flat_frame(w, h, 100, rng, [(0, h // 2, 12, h // 2 + 9)]),         # left edge, beside the region's rows
        rng.integers(0, 256, (h, w, 3), dtype=np.uint8),                    # noise everywhere
    ]
    n = len(frames)

    def fill(p):
        for i, f in enumerate(frames):
            p.upload(i, f)

    full, reg = _run_both(w, h, n, fill, chunk=4)
    _assert_same(full, reg)
    # the test is only worth something if the outside blocks really move the LUT: frame 0's board is not black
    assert full[1][0].max() > 0 and len(np.unique(full[1][3])) == 1


def test_region_edge_cases(gpu_ctx):
    """A quad that covers the whole frame (empty complement), one in a corner (region clipped by two borders), a tiny one."""
    w, h = 640, 480
    n = 3
    for pts in ([(0, 0), (w - 1, 0), (0, h - 1), (w - 1, h - 1)], [(2, 3), (150, 5), (1, 140), (160, 150)],
                [(300, 200), (330, 202), (298, 231), (333, 236)]):
        full, reg = _run_both(w, h, n, lambda p: p.synth(0, n, scene="normal", frames_per_ply=1), pts=pts)
        _assert_same(full, reg)


def test_region_flag_is_ignored_when_the_enhanced_frame_is_kept(gpu_ctx):
    from chessboard_vision_amd.stream import BoardPipeline
    w, h, n = 640, 480, 2
    outs = []
    for region in (False, True):
        p = BoardPipeline(w, h, n)
        p.configure(S.scaled_corners(w, h), profile=S.SHIPPED_PROFILE, keep_enhanced=True, enhance_region=region)
        p.synth(0, n, scene="dim")
        p.run(0, n)
        outs.append([p.download(1, i) for i in range(n)] + [p.download(2, i) for i in range(n)])
        p.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def _random_case(seed):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(40, 900)), int(rng.integers(40, 600))
    # a convex-ish quad: a rectangle with jittered corners, possibly reaching outside the frame
    cx, cy = rng.uniform(0.1, 0.9) * w, rng.uniform(0.1, 0.9) * h
    hw, hh = rng.uniform(8, 0.6 * w), rng.uniform(8, 0.6 * h)
    j = lambda s: rng.uniform(-0.15, 0.15) * s
    pts = [(cx - hw + j(hw), cy - hh + j(hh)), (cx + hw + j(hw), cy - hh + j(hh)),
           (cx - hw + j(hw), cy + hh + j(hh)), (cx + hw + j(hw), cy + hh + j(hh))]
    n = int(rng.integers(1, 6))
    kinds = rng.integers(0, 4, n)
    frames = []
    for k in kinds:
        if k == 0:
            f = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        elif k == 1:
            f = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)
        elif k == 2:  # flat with a few noise blocks anywhere
            f = np.full((h, w, 3), int(rng.integers(20, 230)), np.uint8)
            for _ in range(int(rng.integers(1, 4))):
                x0, y0 = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
                x1, y1 = min(w, x0 + int(rng.integers(2, 60))), min(h, y0 + int(rng.integers(2, 40)))
                f[y0:y1, x0:x1] = rng.integers(0, 256, (y1 - y0, x1 - x0, 3), dtype=np.uint8)
        else:  # smooth gradient + mild noise: often no 0 / 255 anywhere
            gx = np.linspace(60, 160, w)[None, :, None] + np.linspace(0, 40, h)[:, None, None]
            f = np.clip(gx + rng.normal(0, 1.5, (h, w, 3)), 0, 255).astype(np.uint8)
        frames.append(f)
    return w, h, pts, frames, dict(chunk=int(rng.integers(0, 4)), lanes=int(rng.integers(1, 3)),
                                   tile_grid_size=(int(rng.integers(1, 9)), int(rng.integers(1, 9))),
                                   profile=S.SHIPPED_PROFILE if seed % 2 else {})


def run_random_region_case(seed):
    from chessboard_vision_amd.stream import BoardPipeline
    w, h, pts, frames, cfg = _random_case(seed)
    n = len(frames)
    out = []
    for region in (False, True):
        p = BoardPipeline(w, h, n)
        p.configure(pts, enhance_region=region, **cfg)
        for i, f in enumerate(frames):
            p.upload(i, f)
        p.run(0, n)
        out.append(_snapshot(p, n))
        p.close()
    _assert_same(out[0], out[1])


@pytest.mark.parametrize("seed", range(24))
def test_region_limited_random_shapes_quads_and_frames(gpu_ctx, seed):
    run_random_region_case(4000 + seed)
