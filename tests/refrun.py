"""Loading and comparing the reference-run fixtures (tests/golden/ref_*.json|npz, recorded by
tests/golden/make_reference_runs.py from the reference's own classes).  Inputs are regenerated here from the
committed scene description through the oracle's frame generator; nothing under /root/reference is read."""
import hashlib
import json
import os

import numpy as np

from chessboard_vision_amd import synth as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def load_npz(name):
    return np.load(os.path.join(GOLD, name))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(positions):
    """{(file, rank)} -> int, bit = rank * 8 + file (the fixtures' convention)."""
    m = 0
    for (f, r) in positions:
        m |= 1 << (r * 8 + f)
    return m


def unbits(m):
    return {(b % 8, b // 8) for b in range(64) if (m >> b) & 1}


def plain(v):
    if isinstance(v, (list, tuple)):
        return [plain(x) for x in v]
    if isinstance(v, (np.bool_, bool)):
        return bool(v)
    if isinstance(v, np.integer):
        return int(v)
    if isinstance(v, np.floating):
        return float(v)
    return v


def result_rows(results):
    """Same row layout as the generator's: [file, rank, has_piece, method, center, radius, confidence, center_border_diff]."""
    return [[p[0], p[1], bool(r["has_piece"]), r["method"], plain(r["center"]), plain(r["radius"]), plain(r["confidence"]),
             plain(r["center_border_diff"])] for p, r in results.items()]


def detector_state(det):
    """State summary of a PieceDetector-like object (reference_squares, cached_results, detection_history)."""
    keys = sorted(det.reference_squares.keys())
    return {"ref_keys_bits": bits(keys),
            "ref_sha256": sha(np.concatenate([np.asarray(det.reference_squares[k]).ravel() for k in keys])) if keys else "",
            "cached_bits": bits(det.cached_results.keys()),
            "cached_has_bits": bits(k for k, v in det.cached_results.items() if v["has_piece"]),
            "history": {"%d,%d" % k: [bool(x) for x in v] for k, v in sorted(det.detection_history.items())}}


def check_set_for(frame_idx, frames_per_ply):
    """The generator's stand-in for the session's smart-scan set (make_reference_runs.py::check_set_for)."""
    ply = frame_idx // frames_per_ply
    s = set(S.position_after(ply).keys())
    for nxt in S.SCRIPT[ply % (len(S.SCRIPT) + 1):][:2]:
        for _, to in nxt:
            s.add(("abcdefgh".index(to[0]), int(to[1]) - 1))
    return s


def planes_sha(plane_dict):
    keys = sorted(plane_dict.keys())
    return sha(np.concatenate([np.asarray(plane_dict[k]).ravel() for k in keys]))
