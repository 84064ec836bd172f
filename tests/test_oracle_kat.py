"""Known-answer tests that pin the CPU oracle (no GPU): closed-form cases and
the reference's own regression test, restated on the oracle."""
import numpy as np
import pytest

from oracle import cbv_oracle as O


def test_gray_coefficients():
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 0, 0]]], np.uint8)
    assert O.bgr2gray(px).tolist() == [[255, 29, 150, 76, 0]]


def test_hsv_primaries_and_roundtrip():
    px = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [255, 255, 255], [0, 0, 0], [128, 128, 128]]], np.uint8)
    hsv = O.bgr2hsv(px)
    assert hsv[0].tolist() == [[0, 255, 255], [60, 255, 255], [120, 255, 255], [0, 0, 255], [0, 0, 0], [0, 0, 128]]
    assert np.array_equal(O.hsv2bgr(hsv), px)


def test_lab_white_black_and_roundtrip():
    px = np.array([[[255, 255, 255], [0, 0, 0]]], np.uint8)
    assert O.bgr2lab(px)[0].tolist() == [[255, 128, 128], [0, 128, 128]]
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
    back = O.lab2bgr(O.bgr2lab(img))
    # 8-bit Lab quantisation: round trip within a level on average; dark saturated colours are the
    # known worst case of the 8-bit encoding
    err = np.abs(back.astype(int) - img.astype(int))
    assert err.mean() < 1.0 and np.percentile(err, 99) <= 8
    gray = np.repeat(np.arange(256, dtype=np.uint8).reshape(1, 256, 1), 3, axis=2)
    assert np.abs(O.lab2bgr(O.bgr2lab(gray)).astype(int) - gray.astype(int)).max() <= 1


def test_lab_values_opencv_is_publicly_known_to_return():
    """The few OpenCV-side numbers that can be anchored without an OpenCV build: what `cv2.cvtColor(.., COLOR_BGR2LAB)`
    returns for the primaries on 8-bit images is quoted all over OpenCV's forum / Q&A literature (red (136, 208, 195),
    green (224, 42, 211), blue (82, 207, 20)), as are L = 137 for mid-gray 128 and the gray / HSV values of the
    primaries above.  Written down from that public knowledge, not from a fixture: a weak pin, but a real one — the
    integer Lab path with its gamma, cube-root and D65 tables has to be right to hit all nine values."""
    px = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [128, 128, 128]]], np.uint8)
    assert O.bgr2lab(px)[0].tolist() == [[136, 208, 195], [224, 42, 211], [82, 207, 20], [137, 128, 128]]


def test_convert_scale_abs_folds_negative():
    x = np.arange(256, dtype=np.uint8).reshape(1, 256)
    y = O.convert_scale_abs(x, 1.48, -30)
    a, b = np.float32(1.48), np.float32(-30)
    # fused multiply-add (v_fma in cvtabs_32f): exact product + addend in float64, ONE rounding to float32
    fma = (x.astype(np.float64) * np.float64(a) + np.float64(b)).astype(np.float32)
    exp = np.clip(np.rint(np.abs(fma)), 0, 255).astype(np.uint8)
    assert np.array_equal(y, exp)
    assert y[0, 0] == 30 and y[0, 255] == 255


def test_gaussian_kernels():
    import ctypes as C
    for k, exp in ((1, [256]), (3, [64, 128, 64]), (5, [16, 64, 96, 64, 16]), (7, [8, 28, 56, 72, 56, 28, 8])):
        coef = (C.c_int * 64)()
        O.lib().orc_gaussian_kernel_q8(k, coef)
        assert list(coef[:k]) == exp
    for k in (9, 11, 13, 15):
        coef = (C.c_int * 64)()
        O.lib().orc_gaussian_kernel_q8(k, coef)
        c = list(coef[:k])
        assert sum(c) == 256 and c == c[::-1] and max(c) == c[k // 2]


def test_constant_image_is_a_fixed_point_of_the_stencils():
    img = np.full((40, 52, 3), 93, np.uint8)
    assert np.array_equal(O.filter3x3(img), img)          # 9c - 8c = c
    assert np.array_equal(O.bilateral(img), img)
    assert np.array_equal(O.gaussian_blur(img[..., 0].copy(), 5), img[..., 0])
    assert O.normalize_minmax(img).max() == 0              # flat image -> scale 0


def test_blur_impulse_response():
    g = np.zeros((9, 9), np.uint8)
    g[4, 4] = 255
    out = O.gaussian_blur(g, 5)
    k = np.array([1, 4, 6, 4, 1])
    exp = (np.outer(k, k) * 255 + 128) >> 8
    assert np.array_equal(out[2:7, 2:7], exp)


def test_sharpen_saturates():
    img = np.zeros((5, 5, 3), np.uint8)
    img[2, 2] = 100
    out = O.filter3x3(img)
    assert out[2, 2].tolist() == [255, 255, 255] and out[1, 1].tolist() == [0, 0, 0]


def test_normalize_stretches_to_full_range():
    rng = np.random.default_rng(1)
    img = rng.integers(50, 180, size=(30, 30, 3), dtype=np.uint8)
    out = O.normalize_minmax(img)
    assert out.min() == 0 and out.max() == 255
    mn, mx = int(img.min()), int(img.max())
    scale = 255.0 * (1.0 / (mx - mn))
    a, b = np.float32(scale), np.float32(0.0 - mn * scale)
    fma = (img.astype(np.float64) * np.float64(a) + np.float64(b)).astype(np.float32)  # v_fma in cvt_32f
    exp = np.clip(np.rint(fma), 0, 255).astype(np.uint8)
    assert np.array_equal(out, exp)


def test_otsu_two_level():
    hist = np.zeros(256, np.int32)
    hist[40] = 500
    hist[200] = 300
    assert O.otsu_from_hist(hist) == 40   # first maximum: every t in [40, 199] separates the classes
    img = np.full((20, 40, 3), 40, np.uint8)
    img[:, 25:] = 200
    gray, binary, t = O.prepare_analysis(img)
    assert gray[0, 0] == 40 and gray[0, 39] == 200
    assert binary[0, 0] == 0 and binary[0, 39] == 255 and 40 <= t < 200


def test_clahe_constant_tile_arithmetic():
    # 64x64, 8x8 tiles of 8x8 = 64 px; clip = max(int(3*64/256), 1) = 1
    v = 77
    img = np.full((64, 64), v, np.uint8)
    out, lut = O.clahe(img, 3.0, (8, 8), return_lut=True)
    # hist[v] = 64 -> clipped 63 -> batch 0, residual 63 -> step 4 -> +1 on bins 0,4,...,248
    hist = np.zeros(256, int)
    hist[v] = 1
    hist[np.arange(0, 249, 4)[:63]] += 1
    exp = np.clip(np.rint(np.cumsum(hist).astype(np.float32) * np.float32(255.0 / 64)), 0, 255).astype(np.uint8)
    assert np.array_equal(lut[0], exp) and np.array_equal(lut[63], exp)
    assert (out == exp[v]).all()


def test_clahe_non_divisible_pads_both_axes():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(50, 64), dtype=np.uint8)  # width divisible by 8, height not
    out = O.clahe(img, 2.0, (8, 8))
    assert out.shape == img.shape


def test_axis_aligned_warp_is_a_copy():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(700, 800, 3), dtype=np.uint8)
    pts = np.float32([[0, 0], [620, 0], [0, 620], [620, 620]])
    w, M, bs = O.warp_image(img, pts)
    assert bs == 620 and np.allclose(M, np.eye(3), atol=1e-12)
    assert np.array_equal(w, img[:620, :620])
    assert np.array_equal(O.rotate180(w), w[::-1, ::-1])


def test_perspective_transform_maps_corners():
    src = np.float32([[556, 112], [1560, 108], [550, 1005], [1562, 1024]])
    dst = np.float32([[0, 0], [620, 0], [0, 620], [620, 620]])
    M = O.get_perspective_transform(src, dst)
    for (x, y), (u, v) in zip(src, dst):
        p = M @ np.array([x, y, 1.0])
        assert abs(p[0] / p[2] - u) < 1e-8 and abs(p[1] / p[2] - v) < 1e-8
    assert np.allclose(O.invert3x3(M) @ M, np.eye(3), atol=1e-9)


def test_reference_regression_case_on_oracle():
    """test_change_detector_regression.py:31-54 of the reference."""
    from ref_logic import RefChangeDetector
    det = RefChangeDetector()
    squares = {(c, r): np.zeros((50, 50), np.uint8) for r in range(8) for c in range(8)}
    det.calibrate(squares)
    assert det.is_calibrated
    squares[(3, 3)] = np.full((50, 50), 255, np.uint8)
    detailed = det.detect_changes_detailed(squares)
    assert set(detailed) == {(3, 3)}
    assert detailed[(3, 3)]["pct_changed"] == 100.0 and detailed[(3, 3)]["pct_changed"] > 50.0
    assert detailed[(3, 3)]["intensity"] == "TOTAL"
    assert detailed[(3, 3)]["z_score"] == 25.5
    assert detailed[(3, 3)]["is_circular"] is False


def test_ema_matches_numpy_float32():
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, size=(20, 20), dtype=np.uint8)
    mean = rng.uniform(0, 255, size=(20, 20)).astype(np.float32)
    var = rng.uniform(5, 900, size=(20, 20)).astype(np.float32)
    alpha = 0.13
    gf = g.astype(np.float32)
    nm = (1 - alpha) * mean + alpha * gf
    d = gf - nm
    nv = np.maximum((1 - alpha) * var + alpha * (d ** 2), 10.0)
    m2, v2 = mean.copy(), var.copy()
    O.ema_update(g, alpha, m2, v2)
    assert np.array_equal(m2, nm) and np.array_equal(v2, nv)


def test_zscore_matches_numpy_float32():
    rng = np.random.default_rng(4)
    g = rng.integers(0, 256, size=(31, 29), dtype=np.uint8)
    mean = rng.uniform(0, 255, size=g.shape).astype(np.float32)
    var = rng.uniform(10, 900, size=g.shape).astype(np.float32)
    z = np.abs(g.astype(np.float32) - mean) / np.sqrt(var)
    st = O.square_stats(g, mean=mean, var=var, z_thresh=2.55)
    assert st.z_count == np.count_nonzero(z > 2.55) and st.z_max == float(np.max(z))


def test_synthetic_frame_is_deterministic_and_has_a_board():
    from helpers import oracle_frame
    a = oracle_frame(320, 240, "normal")
    b = oracle_frame(320, 240, "normal")
    c = oracle_frame(320, 240, "normal", frame_idx=1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a[120, 180].max() > 100  # inside the board quad


def test_hough_canny_step_edge(oracle):
    """Canny inside HoughCircles: a vertical step of height 200 gives |dx| = 800 on the two columns at the step;
    non-maximum suppression (m > left, m >= right) keeps the left one, replicate borders keep it straight."""
    g = np.zeros((20, 30), np.uint8)
    g[:, 15:] = 200
    _, edges = oracle.hough_circles(g, 1.2, 6, 100, 25, 4, 11, return_edges=True)
    want = np.zeros_like(g)
    want[:, 14] = 255
    assert np.array_equal(edges, want)
    # a ramp whose gradient never exceeds param1 / 2 has no edges at all
    ramp = np.tile((np.arange(30) * 4).astype(np.uint8), (20, 1))
    _, e2 = oracle.hough_circles(ramp, 1.2, 6, 100, 25, 4, 11, return_edges=True)
    assert not e2.any()


@pytest.mark.parametrize("cx,cy,rad", [(38, 38, 30), (30, 44, 22), (45, 33, 26)])
def test_hough_recovers_a_drawn_disc(oracle, cx, cy, rad):
    """Known-answer property of the restated HoughCircles: a clean blurred disc comes back as the first circle,
    centre within one accumulator cell (dp = 1.2) and radius within 2 px."""
    yy, xx = np.mgrid[0:77, 0:77]
    img = np.where((xx - cx) ** 2 + (yy - cy) ** 2 <= rad * rad, 220, 50).astype(np.uint8)
    circles = oracle.hough_circles(oracle.gaussian_blur(img, 5), 1.2, 25, 100, 25, 15, 42)
    assert circles, "no circle found"
    x, y, r, votes = circles[0]
    assert abs(x - cx) <= 1.8 and abs(y - cy) <= 1.8 and abs(r - rad) <= 2.0 and votes > 25
    # circles come back sorted by support, centres at least minDist apart
    for a in range(len(circles)):
        for b in range(a):
            assert (circles[a][0] - circles[b][0]) ** 2 + (circles[a][1] - circles[b][1]) ** 2 >= 25 * 25
            assert circles[b][3] >= circles[a][3]


def test_hough_flat_and_tiny(oracle):
    assert oracle.hough_circles(np.full((77, 77), 90, np.uint8)) == []
    assert oracle.hough_circles(np.array([[0, 255], [255, 0]], np.uint8), 1.2, 0, 100, 25, 0, 1) == []


def test_detect_piece_hough_branch(oracle):
    """piece_detector.py:308-317: a found circle short-circuits with 'hough' (r >= 20 % of the square) or
    'tower_top' (smaller); the uniformity pre-filter still comes first."""
    from ref_logic import detect_piece
    yy, xx = np.mgrid[0:77, 0:77]
    big = np.dstack([np.where((xx - 38) ** 2 + (yy - 38) ** 2 <= 30 * 30, 220, 50).astype(np.uint8)] * 3)
    small = np.dstack([np.where((xx - 38) ** 2 + (yy - 38) ** 2 <= 12 * 12, 230, 40).astype(np.uint8)] * 3)
    r_big = detect_piece(big, hough={})[0]
    assert r_big["has_piece"] and r_big["method"] == "hough" and r_big["confidence"] == 0.9 and abs(r_big["radius"] - 30) <= 2
    r_small = detect_piece(small, hough=dict(min_radius_ratio=0.12, max_radius_ratio=0.25))[0]
    assert r_small["has_piece"] and r_small["method"] == "tower_top" and r_small["confidence"] == 0.75
    flat = np.full((77, 77, 3), 100, np.uint8)
    assert detect_piece(flat, hough={})[0]["has_piece"] is False


@pytest.mark.parametrize("rad", [23, 25, 28, 30, 32, 35, 37])
def test_hough_band_of_the_reference_piece_stats(oracle, rad):
    """piece_stats.txt of the reference (its detector on a real board, 77-px squares) lists every piece as method
    'hough', confidence 90 %, radius 23..37 px.  Not a pin (those are real images), but a plausibility band: the
    restated chain gives the same method and confidence and recovers the radius of drawn pieces across that band,
    with the shipped detector settings (min_radius 25 %, max_radius 55 %)."""
    from ref_logic import detect_piece
    yy, xx = np.mgrid[0:77, 0:77]
    rng = np.random.default_rng(rad)
    img = np.where((xx - 38) ** 2 + (yy - 38) ** 2 <= rad * rad, 215, 70).astype(np.int16) + rng.integers(-5, 6, (77, 77))
    sq = np.dstack([np.clip(img, 0, 255).astype(np.uint8)] * 3)
    res = detect_piece(sq, hough=dict(min_radius_ratio=0.25, max_radius_ratio=0.55))[0]
    assert res["has_piece"] and res["method"] == "hough" and res["confidence"] == 0.9
    assert abs(res["radius"] - rad) <= 2 and abs(res["center"][0] - 38) <= 2 and abs(res["center"][1] - 38) <= 2
