"""The slice of python-chess that game_state.py / game_session.py use, on the
native rules engine (csrc/chess_rules.cpp, include/cbv_chess.h).

python-chess is a third-party dependency of the reference (requirements.txt:
`chess`) that is absent here; this module keeps its names and behaviour for
the calls the reference makes: `chess.Board()`, `.fen()`, `.set_fen()`,
`.reset()`, `.turn`, `.piece_at()`, `.legal_moves` (iteration, `in`),
`.push()`, `.pop()`, `.peek()`, `.move_stack`, `.is_capture()`,
`.is_en_passant()`, `chess.Move`, `chess.square*`, colour / piece constants
and square names.  Host code only; the shared library is required (there is no
Python fallback).
"""
import ctypes as C

from . import _native as N

WHITE, BLACK = True, False
PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = range(1, 7)
PIECE_SYMBOLS = [None, "p", "n", "b", "r", "q", "k"]
FILE_NAMES = "abcdefgh"
RANK_NAMES = "12345678"
SQUARES = list(range(64))
SQUARE_NAMES = [f + r for r in RANK_NAMES for f in FILE_NAMES]
STARTING_FEN = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
MOVE_NONE = 0xFFFF

for _i, _n in enumerate(SQUARE_NAMES):
    globals()[_n.upper()] = _i


def square(file_index, rank_index):
    return rank_index * 8 + file_index


def square_file(sq):
    return sq & 7


def square_rank(sq):
    return sq >> 3


def square_name(sq):
    return SQUARE_NAMES[sq]


def parse_square(name):
    return SQUARE_NAMES.index(name)


_lib = None


def _L():
    global _lib
    if _lib is None:
        lib = N.load()
        vp, i32, u64, u16 = C.c_void_p, C.c_int, C.c_uint64, C.c_uint16
        proto = {
            "cbv_board_create": (vp, []), "cbv_board_destroy": (None, [vp]), "cbv_board_reset": (None, [vp]),
            "cbv_board_set_fen": (i32, [vp, C.c_char_p]), "cbv_board_fen": (i32, [vp, C.c_char_p, i32]),
            "cbv_board_turn": (i32, [vp]), "cbv_board_set_turn": (None, [vp, i32]), "cbv_board_piece_at": (i32, [vp, i32]), "cbv_board_occupancy": (u64, [vp]),
            "cbv_board_legal_moves": (i32, [vp, C.POINTER(u16), i32]), "cbv_board_is_legal": (i32, [vp, u16]),
            "cbv_board_is_capture": (i32, [vp, u16]), "cbv_board_is_en_passant": (i32, [vp, u16]),
            "cbv_board_is_check": (i32, [vp]), "cbv_board_push": (i32, [vp, u16]), "cbv_board_pop": (u16, [vp]),
            "cbv_board_ply": (i32, [vp]), "cbv_board_peek": (u16, [vp]), "cbv_board_perft": (u64, [vp, i32]),
            "cbv_game_process_occupancy": (i32, [vp, u64, C.POINTER(u16)]), "cbv_game_status_name": (C.c_char_p, [i32]),
            "cbv_roi_bits_to_squares": (u64, [u64]), "cbv_game_infer_move": (i32, [vp, u64, C.POINTER(u16)]),
        }
        for name, (res, args) in proto.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class Piece:
    def __init__(self, piece_type, color):
        self.piece_type, self.color = piece_type, color

    def symbol(self):
        s = PIECE_SYMBOLS[self.piece_type]
        return s.upper() if self.color else s

    def __eq__(self, other):
        return isinstance(other, Piece) and (self.piece_type, self.color) == (other.piece_type, other.color)

    def __hash__(self):
        return self.piece_type + (0 if self.color else 8)

    def __repr__(self):
        return "Piece.from_symbol(%r)" % self.symbol()


class Move:
    def __init__(self, from_square, to_square, promotion=None):
        self.from_square, self.to_square, self.promotion = from_square, to_square, promotion

    @classmethod
    def from_uci(cls, uci):
        if len(uci) not in (4, 5):
            raise ValueError("expected uci string to be of length 4 or 5: %r" % uci)
        promo = PIECE_SYMBOLS.index(uci[4]) if len(uci) == 5 else None
        return cls(parse_square(uci[0:2]), parse_square(uci[2:4]), promo)

    @classmethod
    def _from_code(cls, code):
        if code == MOVE_NONE:
            return None
        promo = (code >> 12) & 7
        return cls(code & 63, (code >> 6) & 63, promo or None)

    def _code(self):
        return self.from_square | (self.to_square << 6) | ((self.promotion or 0) << 12)

    def uci(self):
        s = SQUARE_NAMES[self.from_square] + SQUARE_NAMES[self.to_square]
        return s + PIECE_SYMBOLS[self.promotion] if self.promotion else s

    def __eq__(self, other):
        return isinstance(other, Move) and self._code() == other._code()

    def __hash__(self):
        return self._code()

    def __repr__(self):
        return "Move.from_uci(%r)" % self.uci()

    __str__ = uci


class LegalMoveGenerator:
    """`board.legal_moves`: iterable, sized, supports `move in ...`."""

    def __init__(self, board):
        self.board = board

    def _codes(self):
        buf = (C.c_uint16 * 256)()
        n = _L().cbv_board_legal_moves(self.board._h, buf, 256)
        return [buf[i] for i in range(n)]

    def __iter__(self):
        return iter([Move._from_code(c) for c in self._codes()])

    def __len__(self):
        return len(self._codes())

    def count(self):
        return len(self)

    def __bool__(self):
        return len(self) > 0

    def __contains__(self, move):
        return bool(_L().cbv_board_is_legal(self.board._h, move._code()))


class Board:
    def __init__(self, fen=STARTING_FEN):
        self._h = C.c_void_p(_L().cbv_board_create())
        if fen != STARTING_FEN:
            self.set_fen(fen)

    def __del__(self):
        try:
            if self._h:
                _L().cbv_board_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def turn(self):
        return bool(_L().cbv_board_turn(self._h))

    @turn.setter
    def turn(self, white):
        _L().cbv_board_set_turn(self._h, 1 if white else 0)

    @property
    def legal_moves(self):
        return LegalMoveGenerator(self)

    @property
    def move_stack(self):
        """Moves played since the last set_fen/reset (read-only copy)."""
        lib, n, out = _L(), _L().cbv_board_ply(self._h), []
        for _ in range(n):
            out.append(Move._from_code(lib.cbv_board_pop(self._h)))
        for m in reversed(out):
            lib.cbv_board_push(self._h, m._code())
        return list(reversed(out))

    def fen(self):
        buf = C.create_string_buffer(128)
        _L().cbv_board_fen(self._h, buf, 128)
        return buf.value.decode()

    def set_fen(self, fen):
        if _L().cbv_board_set_fen(self._h, fen.encode()) != 0:
            raise ValueError("invalid fen: %r" % fen)

    def reset(self):
        _L().cbv_board_reset(self._h)

    def piece_at(self, sq):
        p = _L().cbv_board_piece_at(self._h, sq)
        return Piece(p & 7, not (p & 8)) if p else None

    def occupancy_bits(self):
        return _L().cbv_board_occupancy(self._h)

    def push(self, move):
        _L().cbv_board_push(self._h, move._code())

    def push_uci(self, uci):
        """Parse and play a UCI move; ValueError when it is not legal here (lichess_session.py uses it to sync)."""
        move = Move.from_uci(uci)
        if move not in self.legal_moves:
            raise ValueError("illegal uci: %r in %s" % (uci, self.fen()))
        self.push(move)
        return move

    def pop(self):
        m = Move._from_code(_L().cbv_board_pop(self._h))
        if m is None:
            raise IndexError("pop from empty move stack")
        return m

    def peek(self):
        m = Move._from_code(_L().cbv_board_peek(self._h))
        if m is None:
            raise IndexError("peek at empty move stack")
        return m

    def is_capture(self, move):
        return bool(_L().cbv_board_is_capture(self._h, move._code()))

    def is_en_passant(self, move):
        return bool(_L().cbv_board_is_en_passant(self._h, move._code()))

    def is_check(self):
        return bool(_L().cbv_board_is_check(self._h))

    def perft(self, depth):
        return _L().cbv_board_perft(self._h, depth)

    def __str__(self):
        rows = []
        for r in range(7, -1, -1):
            rows.append(" ".join((self.piece_at(square(f, r)).symbol() if self.piece_at(square(f, r)) else ".") for f in range(8)))
        return "\n".join(rows)


def roi_bits_to_squares(bits):
    """Occupancy word of cbv_frame_result (bit = 8 * row + col of the warped board) -> python-chess square bits."""
    return _L().cbv_roi_bits_to_squares(bits)
