"""NoiseHandler — the hand / move-stabilisation state machine that consumes the
`visual_changes` set PieceDetector.detect_all_pieces emits each frame
(game_session.py:157-165).  Same class surface and return values as the
reference's noise_handler.py:21-243; written as a transition
table so the device kernel (k_noise, k_squares.hip) and this class share one
specification.

States: IDLE (nothing moves), NOISE_ACTIVE (more than NOISE_THRESHOLD squares
changed: a hand is over the board, moves are blocked), MOVE_PENDING (1-3
squares changed: wait until they stay put for STABILITY_FRAMES frames).
"""
from enum import Enum, auto


class NoiseState(Enum):
    IDLE = auto()
    NOISE_ACTIVE = auto()
    MOVE_PENDING = auto()


# message codes shared with the device kernel (cbv_noise_result.msg)
MESSAGES = ("waiting", "hand_detected", "detecting", "noise_cleared", "clearing", "stabilizing", "hand_active",
            "interrupted_by_hand", "move_ready", "stable_ready", "counting", "updated")


class NoiseHandler:
    NOISE_THRESHOLD = 3
    STABILITY_FRAMES = 12
    COOLDOWN_FRAMES = 5

    def __init__(self):
        self._reset()

    def _reset(self):
        self.state = NoiseState.IDLE
        self.pending_squares = set()
        self.stable_count = 0
        self.cooldown_count = 0
        self.last_lifted_square = None

    def reset(self):
        self._reset()

    def is_blocked(self):
        return self.state == NoiseState.NOISE_ACTIVE

    def get_state_name(self):
        return {NoiseState.IDLE: "IDLE", NoiseState.NOISE_ACTIVE: "NOISE", NoiseState.MOVE_PENDING: "PENDING"}.get(self.state, "UNKNOWN")

    # -- helpers ------------------------------------------------------------
    def _start_pending(self, changed):
        self.state = NoiseState.MOVE_PENDING
        self.pending_squares = changed.copy()
        self.stable_count = 1

    def _track_lifted(self, changed):
        self.last_lifted_square = list(changed)[0] if len(changed) == 1 else None

    def process(self, changed_squares):
        """One frame.  Returns (NoiseState, dict) exactly like the reference."""
        n = len(changed_squares)
        noisy = n > self.NOISE_THRESHOLD
        if self.state == NoiseState.IDLE:
            if n == 0:
                return NoiseState.IDLE, {"message": "waiting"}
            if noisy:
                self.state = NoiseState.NOISE_ACTIVE
                self.cooldown_count = 0
                return NoiseState.NOISE_ACTIVE, {"message": "hand_detected", "changed_count": n}
            self._start_pending(changed_squares)
            self._track_lifted(changed_squares)
            return NoiseState.MOVE_PENDING, {"message": "detecting", "squares": self.pending_squares,
                                             "lifted": self.last_lifted_square, "stable": False,
                                             "progress": self.stable_count / self.STABILITY_FRAMES}

        if self.state == NoiseState.NOISE_ACTIVE:
            if noisy:
                self.cooldown_count = 0
                return NoiseState.NOISE_ACTIVE, {"message": "hand_active", "changed_count": n}
            self.cooldown_count += 1
            done = self.cooldown_count >= self.COOLDOWN_FRAMES
            if n == 0:
                if done:
                    self.state = NoiseState.IDLE
                    self.cooldown_count = 0
                    return NoiseState.IDLE, {"message": "noise_cleared"}
                return NoiseState.NOISE_ACTIVE, {"message": "clearing", "cooldown": self.cooldown_count,
                                                 "progress": self.cooldown_count / self.COOLDOWN_FRAMES}
            if done:
                self._start_pending(changed_squares)
                return NoiseState.MOVE_PENDING, {"message": "detecting", "squares": self.pending_squares, "stable": False}
            return NoiseState.NOISE_ACTIVE, {"message": "stabilizing", "changed_count": n}

        if self.state == NoiseState.MOVE_PENDING:
            if noisy:
                self.state = NoiseState.NOISE_ACTIVE
                self.pending_squares = set()
                self.stable_count = 0
                self.cooldown_count = 0
                return NoiseState.NOISE_ACTIVE, {"message": "interrupted_by_hand", "changed_count": n}
            if n == 0:
                self.stable_count += 1
                if self.stable_count >= self.STABILITY_FRAMES:
                    squares = self.pending_squares.copy()
                    self._reset()
                    return NoiseState.IDLE, {"message": "move_ready", "squares": squares, "stable": True}
                return NoiseState.MOVE_PENDING, {"message": "stabilizing", "squares": self.pending_squares, "stable": False,
                                                 "progress": self.stable_count / self.STABILITY_FRAMES}
            if changed_squares == self.pending_squares:
                self.stable_count += 1
                if self.stable_count >= self.STABILITY_FRAMES:
                    return NoiseState.MOVE_PENDING, {"message": "stable_ready", "squares": self.pending_squares.copy(),
                                                     "stable": True, "progress": 1.0}
                return NoiseState.MOVE_PENDING, {"message": "counting", "squares": self.pending_squares,
                                                 "lifted": self.last_lifted_square if len(self.pending_squares) == 1 else None,
                                                 "stable": False, "progress": self.stable_count / self.STABILITY_FRAMES}
            self._start_pending(changed_squares)
            self._track_lifted(changed_squares)
            return NoiseState.MOVE_PENDING, {"message": "updated", "squares": self.pending_squares,
                                             "lifted": self.last_lifted_square, "stable": False,
                                             "progress": self.stable_count / self.STABILITY_FRAMES}
        return self.state, {}


# ---------------------------------------------------------------------------
# device results -> the reference's (NoiseState, dict) tuples
# ---------------------------------------------------------------------------
_DEV_STATES = (NoiseState.IDLE, NoiseState.NOISE_ACTIVE, NoiseState.MOVE_PENDING)
_DEV_MESSAGES = ("waiting", "hand_detected", "detecting", "noise_cleared", "clearing", "stabilizing", "hand_active", "detecting",
                 "interrupted_by_hand", "move_ready", "stabilizing", "stable_ready", "counting", "updated")


def decode_device_result(r, index_to_pos):
    """cbv_noise_result -> (NoiseState, data dict) as NoiseHandler.process returns it.
    `index_to_pos[i]` is the (file, rank) of roi i."""
    squares = {index_to_pos[i] for i in range(len(index_to_pos)) if (r.squares >> i) & 1}
    lifted = index_to_pos[r.lifted] if r.lifted >= 0 else None
    m, msg = r.msg, _DEV_MESSAGES[r.msg]
    if m in (0, 3):
        data = {"message": msg}
    elif m in (1, 5, 6, 8):
        data = {"message": msg, "changed_count": r.count}
    elif m == 4:
        data = {"message": msg, "cooldown": r.count, "progress": r.count / NoiseHandler.COOLDOWN_FRAMES}
    elif m == 7:
        data = {"message": msg, "squares": squares, "stable": False}
    elif m == 9:
        data = {"message": msg, "squares": squares, "stable": True}
    elif m == 10:
        data = {"message": msg, "squares": squares, "stable": False, "progress": r.count / NoiseHandler.STABILITY_FRAMES}
    elif m == 11:
        data = {"message": msg, "squares": squares, "stable": True, "progress": 1.0}
    else:  # 2 detecting (from IDLE), 12 counting, 13 updated
        data = {"message": msg, "squares": squares, "lifted": lifted, "stable": False,
                "progress": r.count / NoiseHandler.STABILITY_FRAMES}
    return _DEV_STATES[r.state], data


def run_on_device(change_sets, pos_to_index, ctx=None, state=None):
    """NoiseHandler.process for a whole sequence of change sets in one device call (cbv_noise_run)."""
    import ctypes as C
    import numpy as np
    from . import _native as N
    ctx = ctx or N.context()
    index_to_pos = [None] * (max(pos_to_index.values()) + 1)
    for p, i in pos_to_index.items():
        index_to_pos[i] = p
    bits = np.array([sum(1 << pos_to_index[p] for p in s) for s in change_sets], dtype=np.uint64)
    st = state or N.NoiseDevState()
    out = (N.NoiseResult * len(bits))()
    ctx.check(ctx.lib.cbv_noise_run(ctx.h, N.ptr(bits), len(bits), C.byref(st), out))
    return [decode_device_result(r, index_to_pos) for r in out], st
