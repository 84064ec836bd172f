"""GameState — drop-in for game_state.py on the native rules engine.

`process_occupancy_change` is one native call (cbv_game_process_occupancy,
include/cbv_chess.h) that restates game_state.py:40-195: diff the vision
occupancy against the board, recognise normal move (auto-queen promotion),
castling, en passant and capture, and push the move when it is legal.  The
statuses are the reference's strings.  `self.board` is a
`chess_rules.Board` with the python-chess surface the application touches
(game_session.py:147-380).
"""
import ctypes as C

from . import chess_rules as chess


def _bits(squares):
    bits = 0
    for f, r in squares:
        bits |= 1 << chess.square(f, r)
    return bits


class GameState:
    def __init__(self):
        self.board = chess.Board()

    def get_fen(self):
        return self.board.fen()

    def get_turn(self):
        return self.board.turn

    def get_turn_name(self):
        return "white" if self.board.turn == chess.WHITE else "black"

    def get_legal_moves(self):
        return list(self.board.legal_moves)

    def get_legal_moves_from(self, file, rank):
        src = chess.square(file, rank)
        return [m for m in self.board.legal_moves if m.from_square == src]

    def get_board_occupancy(self):
        """{(file, rank)} of the occupied squares, a1 = (0, 0) (game_state.py:26-38)."""
        bits = self.board.occupancy_bits()
        return {(s & 7, s >> 3) for s in range(64) if (bits >> s) & 1}

    def process_occupancy_change(self, vision_occupancy_grid):
        """(move, status) like game_state.py:40-112; the board is advanced when a move is confirmed."""
        return self.process_occupancy_bits(_bits(vision_occupancy_grid))

    def process_occupancy_bits(self, square_bits):
        """Same, from an occupancy word (bit = python-chess square), e.g.
        chess_rules.roi_bits_to_squares(frame_result.stable_occupied)."""
        lib = chess._L()
        code = C.c_uint16(chess.MOVE_NONE)
        status = lib.cbv_game_process_occupancy(self.board._h, square_bits, C.byref(code))
        return chess.Move._from_code(code.value), lib.cbv_game_status_name(status).decode()

    def reset(self):
        self.board.reset()

    def set_fen(self, fen):
        self.board.set_fen(fen)


class StableMoveTracker:
    """The move-acceptance logic of GameSession._process_stable_move / _infer_move (game_session.py:181-265)
    without its UI and network side: an occupancy must stay identical for STABILITY_REQUIRED frames, differ from
    the board by at most 4 squares, the cooldown since the last move must have passed and the NoiseHandler must not
    report NOISE_ACTIVE; then exactly one legal move has to explain the difference.

    `on_move_detected(move) -> bool` is the reference's hook (True = apply locally); `after_move()` is where the
    session refreshes the detector references and resets the noise handler (game_session.py:220-223)."""

    STABILITY_REQUIRED = 20
    MOVE_COOLDOWN = 2.0

    def __init__(self, game, clock=None, on_move_detected=None, after_move=None):
        import time
        self.game = game
        self.clock = clock or time.time
        self.on_move_detected = on_move_detected or (lambda move: True)
        self.after_move = after_move or (lambda: None)
        self.stable_occupancy = None
        self.stable_count = 0
        self.last_move_time = 0

    def infer_move(self, vision_occupied):
        """(move or None, number of candidate moves) — game_session.py:229-265."""
        lib = chess._L()
        code = C.c_uint16(chess.MOVE_NONE)
        n = lib.cbv_game_infer_move(self.game.board._h, _bits(vision_occupied), C.byref(code))
        return chess.Move._from_code(code.value), n

    def process(self, vision_occupied, noise_active=False):
        """One frame; returns the move that was pushed on the board, or None."""
        expected = self.game.get_board_occupancy()
        total_diff = len(expected - vision_occupied) + len(vision_occupied - expected)
        if total_diff > 4:                       # a hand or noise: start over
            self.stable_count = 0
            self.stable_occupancy = set()
        elif self.stable_occupancy == vision_occupied:
            self.stable_count += 1
        else:
            self.stable_occupancy = set(vision_occupied)
            self.stable_count = 1
        now = self.clock()
        if self.stable_count < self.STABILITY_REQUIRED or not (now - self.last_move_time) > self.MOVE_COOLDOWN or noise_active:
            return None
        move, _ = self.infer_move(vision_occupied)
        if move is None or not self.on_move_detected(move):
            return None
        if move not in self.game.board.legal_moves:
            return None
        self.game.board.push(move)
        self.last_move_time = now
        self.after_move()
        self.stable_count = 0
        return move


def smart_scan_squares(game):
    """The `squares_to_check` set GameSession builds between full scans (game_session.py:130-152): occupied squares
    plus, for every legal move, (file, 7 - rank) of its destination — the reference's own conversion, kept as is."""
    out = set(game.get_board_occupancy())
    for m in game.board.legal_moves:
        out.add((chess.square_file(m.to_square), 7 - chess.square_rank(m.to_square)))
    return out
