"""Shared test helpers: synthetic scenes through the oracle."""
import numpy as np

from chessboard_vision_amd import synth as S
from oracle import cbv_oracle as O


def oracle_scene(name):
    d = S.SCENES[name]
    sc = O.Scene()
    sc.bg_lo, sc.bg_span, sc.noise, sc.radius = d["bg_lo"], d["bg_span"], d["noise"], d["radius"]
    for k in ("light", "dark", "white", "black"):
        for i in range(3):
            getattr(sc, k)[i] = d[k][i]
    return sc


def oracle_frame(w, h, scene="normal", stream_id=0, frame_idx=0, frames_per_ply=32):
    pts = S.scaled_corners(w, h)
    Hinv = O.get_perspective_transform(pts, S.BOARD_UNIT_QUAD)
    board = S.board_array(S.position_for_frame(frame_idx, frames_per_ply))
    return O.synth_frame(S.frame_seed(stream_id, frame_idx), w, h, Hinv, board, oracle_scene(scene))


def random_frame(w, h, seed, smooth=True):
    rng = np.random.default_rng(seed)
    if not smooth:
        return rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    # low-frequency colour field + noise: exercises every histogram bin region without being white noise
    yy, xx = np.mgrid[:h, :w]
    img = np.empty((h, w, 3), np.float64)
    for c in range(3):
        fx, fy, ph = rng.uniform(0.5, 3), rng.uniform(0.5, 3), rng.uniform(0, 6.28)
        img[..., c] = 127 + 110 * np.sin(fx * xx / w * 6.28 + ph) * np.cos(fy * yy / h * 6.28)
    img += rng.normal(0, 6, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)
