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
