"""Synthetic camera frames for tests and bench (ours).

Pure-Python description of the scene: palettes, the board quad (the
reference's calibration.json corners scaled to the frame), and a scripted
game so occupancy changes along a stream.  The pixels themselves are produced
by the native generators (HIP kernel in the product, C in the oracle), which
are bit-identical by construction (integer hash + double arithmetic without
contraction).
"""
import numpy as np

# calibration.json corners of the reference (1920x1080 feed): TL, TR, BR, BL
CALIB_CORNERS_1080P = ((556, 112), (1560, 108), (1562, 1024), (550, 1005))
CALIB_GRID_X = (0, 79, 157, 234, 310, 386, 464, 541, 620)
CALIB_GRID_Y = (0, 80, 158, 235, 311, 388, 465, 542, 620)

# color_profile.json of the reference
SHIPPED_PROFILE = {"hue_shift": -86, "sat_scale": 0.91, "val_scale": 2.46, "contrast": 1.48,
                   "brightness": -30, "radical_mode": 0, "target_hue": 0, "hue_window": 26}

# piece_detector_settings.json of the reference: min_radius 25, max_radius 55 (percent of the square)
SHIPPED_DETECTOR = {"min_radius_ratio": 0.25, "max_radius_ratio": 0.55}

# Scene palettes (BGR).  "normal": a well-lit board, used with an empty colour
# profile.  "dim": an under-exposed camera, the situation the shipped profile
# (contrast 1.48, brightness -30, val x2.46) was calibrated for.
SCENES = {
    "normal": dict(bg_lo=60, bg_span=31, light=(140, 160, 180), dark=(60, 85, 115),
                   white=(245, 245, 240), black=(20, 20, 25), noise=3, radius=0.42),
    "dim": dict(bg_lo=24, bg_span=8, light=(62, 66, 70), dark=(40, 43, 48),
                white=(84, 84, 82), black=(21, 21, 22), noise=1, radius=0.42),
    # every byte uniform over 0..255 (mid-gray everywhere + noise of amplitude 127 wraps nothing: 128 +- 127): the worst
    # case for data-dependent gathers; there is no board to detect in it
    "white_noise": dict(bg_lo=128, bg_span=1, light=(128, 128, 128), dark=(128, 128, 128),
                        white=(128, 128, 128), black=(128, 128, 128), noise=127, radius=0.42),
}


def scaled_corners(w, h):
    """Board quad for a w x h frame, in warp_image's order TL, TR, BL, BR
    (what board_detection.reorder returns, board_detection.py:49-58)."""
    tl, tr, br, bl = [(x * w / 1920.0, y * h / 1080.0) for (x, y) in CALIB_CORNERS_1080P]
    pts = np.array([tl, tr, bl, br], dtype=np.float32)
    return pts


BOARD_UNIT_QUAD = np.array([[0, 0], [8, 0], [0, 8], [8, 8]], dtype=np.float32)

START_ROWS = ("bbbbbbbb", "bbbbbbbb", "........", "........", "........", "........", "wwwwwwww", "wwwwwwww")

# Ruy Lopez, 16 plies; each ply is a list of (from, to) squares in algebraic.
SCRIPT = [
    [("e2", "e4")], [("e7", "e5")], [("g1", "f3")], [("b8", "c6")], [("f1", "b5")], [("a7", "a6")],
    [("b5", "a4")], [("g8", "f6")], [("e1", "g1"), ("h1", "f1")], [("f8", "e7")], [("f1", "e1")],
    [("b7", "b5")], [("a4", "b3")], [("d7", "d6")], [("c2", "c3")], [("e8", "g8"), ("h8", "f8")],
]

START_PIECES = {
    "a1": "R", "b1": "N", "c1": "B", "d1": "Q", "e1": "K", "f1": "B", "g1": "N", "h1": "R",
    "a8": "r", "b8": "n", "c8": "b", "d8": "q", "e8": "k", "f8": "b", "g8": "n", "h8": "r",
}
for _f in "abcdefgh":
    START_PIECES[_f + "2"] = "P"
    START_PIECES[_f + "7"] = "p"


def _sq(name):
    return "abcdefgh".index(name[0]), int(name[1]) - 1  # (file, rank) a1 = (0, 0)


def position_after(plies):
    """dict {(file, rank): fen_char} after the first `plies` plies of SCRIPT (cyclic:
    after the script ends the game restarts from the initial position)."""
    plies = plies % (len(SCRIPT) + 1)
    pos = {_sq(k): v for k, v in START_PIECES.items()}
    for ply in SCRIPT[:plies]:
        for frm, to in ply:
            pos[_sq(to)] = pos.pop(_sq(frm))
    return pos


def board_array(pos):
    """64 bytes row-major from rank 8: 0 empty, 1 white, 2 black."""
    b = np.zeros(64, np.uint8)
    for (f, r), ch in pos.items():
        b[(7 - r) * 8 + f] = 1 if ch.isupper() else 2
    return b


def position_for_frame(frame_idx, frames_per_ply=32):
    return position_after(frame_idx // frames_per_ply)


def frame_seed(stream_id, frame_idx):
    return (0xC0FFEE + (stream_id << 32) + frame_idx) & 0xFFFFFFFFFFFFFFFF
