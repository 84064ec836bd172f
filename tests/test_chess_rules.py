"""the rules engine behind GameState (csrc/chess_rules.cpp) and the
GameState drop-in.  Host code; runs without a GPU.

Pinned by (a) the published perft node counts (chessprogramming.org "Perft
Results": initial position, Kiwipete and positions 3-5) — a move generator that
reproduces them generates exactly the legal moves, including castling, en
passant, promotions and pins; (b) the ten cases of the reference's own
test_game_state.py:9-158 (FEN / UCI / status literals read from it as data);
(c) python-chess's documented Board.fen() conventions.
"""
import pytest

from chessboard_vision_amd import chess_rules as chess
from chessboard_vision_amd.game_state import GameState

PERFT = [
    (chess.STARTING_FEN, [20, 400, 8902, 197281]),
    ("r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", [48, 2039, 97862]),
    ("8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1", [14, 191, 2812, 43238]),
    ("r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1", [6, 264, 9467]),
    ("r2q1rk1/pP1p2pp/Q4n2/bbp1p3/Np6/1B3NBn/pPPP1PPP/R3K2R b KQ - 0 1", [6, 264, 9467]),
    ("rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8", [44, 1486, 62379]),
    ("r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10", [46, 2079, 89890]),
]


@pytest.mark.parametrize("fen,counts", PERFT)
def test_perft_known_answers(fen, counts):
    b = chess.Board(fen)
    assert [b.perft(d + 1) for d in range(len(counts))] == counts
    assert b.fen().split()[0] == fen.split()[0], "perft must leave the board untouched"


def test_start_position_move_order_is_python_chess_order():
    ucis = [m.uci() for m in chess.Board().legal_moves]
    assert ucis == ["g1h3", "g1f3", "b1c3", "b1a3", "h2h3", "g2g3", "f2f3", "e2e3", "d2d3", "c2c3", "b2b3", "a2a3",
                    "h2h4", "g2g4", "f2f4", "e2e4", "d2d4", "c2c4", "b2b4", "a2a4"]


def test_fen_conventions():
    b = chess.Board()
    assert b.fen() == chess.STARTING_FEN
    b.push(chess.Move.from_uci("e2e4"))
    # the en passant square is printed only when an en passant capture is legal
    assert b.fen() == "rnbqkbnr/pppppppp/8/8/4P3/8/PPPP1PPP/RNBQKBNR b KQkq - 0 1"
    b.push(chess.Move.from_uci("a7a6"))
    b.push(chess.Move.from_uci("e4e5"))
    b.push(chess.Move.from_uci("d7d5"))
    assert b.fen() == "rnbqkbnr/1pp1pppp/p7/3pP3/8/8/PPPP1PPP/RNBQKBNR w KQkq d6 0 3"
    b.push(chess.Move.from_uci("e5d6"))
    assert b.fen() == "rnbqkbnr/1pp1pppp/p2P4/8/8/8/PPPP1PPP/RNBQKBNR b KQkq - 0 3"
    assert b.pop().uci() == "e5d6" and b.fen() == "rnbqkbnr/1pp1pppp/p7/3pP3/8/8/PPPP1PPP/RNBQKBNR w KQkq d6 0 3"
    # rook and king moves drop castling rights; halfmove clock counts quiet moves
    c = chess.Board("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1")
    c.push(chess.Move.from_uci("h1g1"))
    assert c.fen() == "r3k2r/8/8/8/8/8/8/R3K1R1 b Qkq - 1 1"
    c.push(chess.Move.from_uci("e8e7"))
    assert c.fen() == "r6r/4k3/8/8/8/8/8/R3K1R1 w Q - 2 2"
    c.push(chess.Move.from_uci("e1c1"))
    assert c.fen() == "r6r/4k3/8/8/8/8/8/2KR2R1 b - - 3 2"
    assert [m.uci() for m in c.move_stack] == ["h1g1", "e8e7", "e1c1"] and c.peek().uci() == "e1c1"
    # rights without their rook are not printed
    assert chess.Board("4k3/8/8/8/8/8/8/4K2R w KQkq - 0 1").fen() == "4k3/8/8/8/8/8/8/4K2R w K - 0 1"
    with pytest.raises(ValueError):
        chess.Board("not a fen")
    # promotion with capture, and its undo
    p = chess.Board("1n2k3/P7/8/8/8/8/8/4K3 w - - 0 1")
    assert {m.uci() for m in p.legal_moves if m.from_square == chess.A7} == {"a7a8q", "a7a8r", "a7a8b", "a7a8n", "a7b8q", "a7b8r", "a7b8b", "a7b8n"}
    p.push(chess.Move.from_uci("a7b8q"))
    assert p.fen() == "1Q2k3/8/8/8/8/8/8/4K3 b - - 0 1" and p.is_check()
    p.pop()
    assert p.fen() == "1n2k3/P7/8/8/8/8/8/4K3 w - - 0 1"


def _moved(gs, remove, add):
    occ = gs.get_board_occupancy()
    for sq in remove:
        occ.remove(sq)
    for sq in add:
        occ.add(sq)
    return occ


# --- the reference's own cases (test_game_state.py), restated as data -------------------------------------
def test_initial_occupancy():  # :9-16
    occ = GameState().get_board_occupancy()
    assert len(occ) == 32 and (0, 0) in occ and (4, 1) in occ and (4, 4) not in occ


def test_normal_move_logic():  # :18-32
    gs = GameState()
    move, status = gs.process_occupancy_change(_moved(gs, [(4, 1)], [(4, 3)]))
    assert status == "move_confirmed" and move.uci() == "e2e4"
    assert gs.board.piece_at(chess.E2) is None and gs.board.piece_at(chess.E4) is not None


def test_illegal_move_logic():  # :34-46
    gs = GameState()
    move, status = gs.process_occupancy_change(_moved(gs, [(4, 1)], [(4, 4)]))
    assert status == "illegal_move" and move is None and gs.board.piece_at(chess.E2) is not None


def test_turn_switching():  # :48-63
    gs = GameState()
    gs.process_occupancy_change(_moved(gs, [(4, 1)], [(4, 3)]))
    assert gs.board.turn == chess.BLACK and gs.get_turn_name() == "black"
    move, msg = gs.process_occupancy_change(_moved(gs, [(4, 6)], [(4, 4)]))
    assert msg == "move_confirmed" and move.uci() == "e7e5"


def test_capture_logic_simple():  # :65-82
    gs = GameState()
    gs.board.push(chess.Move.from_uci("e2e4"))
    gs.board.push(chess.Move.from_uci("d7d5"))
    move, status = gs.process_occupancy_change(_moved(gs, [(4, 3)], []))
    assert status == "capture_confirmed" and move.uci() == "e4d5"
    piece = gs.board.piece_at(chess.D5)
    assert piece is not None and piece.color == chess.WHITE


def test_kingside_castling():  # :84-104
    gs = GameState()
    gs.set_fen("r1bqk2r/pppp1ppp/2n2n2/2b1p3/2B1P3/5N2/PPPP1PPP/RNBQK2R w KQkq - 4 4")
    move, status = gs.process_occupancy_change(_moved(gs, [(4, 0), (7, 0)], [(6, 0), (5, 0)]))
    assert status == "castling_confirmed" and move.uci() == "e1g1"
    assert gs.board.piece_at(chess.G1).piece_type == chess.KING and gs.board.piece_at(chess.F1).piece_type == chess.ROOK


def test_queenside_castling():  # :106-120
    gs = GameState()
    gs.set_fen("r3kbnr/pppqpppp/2n5/3p1b2/3P1B2/2N5/PPPQPPPP/R3KBNR w KQkq - 6 5")
    move, status = gs.process_occupancy_change(_moved(gs, [(4, 0), (0, 0)], [(2, 0), (3, 0)]))
    assert status == "castling_confirmed" and move.uci() == "e1c1"


def test_en_passant():  # :122-143
    gs = GameState()
    gs.set_fen("rnbqkbnr/ppp1pppp/8/3pP3/8/8/PPPP1PPP/RNBQKBNR w KQkq d6 0 3")
    move, status = gs.process_occupancy_change(_moved(gs, [(4, 4), (3, 4)], [(3, 5)]))
    assert status == "en_passant_confirmed" and move.uci() == "e5d6"
    assert gs.board.piece_at(chess.D5) is None and gs.board.piece_at(chess.D6).piece_type == chess.PAWN


def test_get_turn():  # :145-151
    gs = GameState()
    assert gs.get_turn() == chess.WHITE and gs.get_turn_name() == "white"
    gs.board.push(chess.Move.from_uci("e2e4"))
    assert gs.get_turn() == chess.BLACK and gs.get_turn_name() == "black"


def test_reset():  # :153-156
    gs = GameState()
    gs.board.push(chess.Move.from_uci("e2e4"))
    gs.reset()
    assert gs.get_fen() == chess.STARTING_FEN


# --- beyond the reference's cases --------------------------------------------------------------------------
def test_other_statuses_and_promotion():
    gs = GameState()
    assert gs.process_occupancy_change(gs.get_board_occupancy()) == (None, "no_valid_change")
    # two attackers could have taken on d5: e4xd5 or c4xd5 -> the vanished square decides, so this is NOT ambiguous
    gs.set_fen("rnbqkbnr/ppp1pppp/8/3p4/2P1P3/8/PP1P1PPP/RNBQKBNR w KQkq - 0 3")
    move, status = gs.process_occupancy_change(_moved(gs, [(2, 3)], []))
    assert (move.uci(), status) == ("c4d5", "capture_confirmed")
    # one attacker, two victims it could have taken -> ambiguous_capture, board untouched
    gs.set_fen("4k3/8/8/2p1p3/3P4/8/8/4K3 w - - 0 1")
    fen = gs.get_fen()
    assert gs.process_occupancy_change(_moved(gs, [(3, 3)], [])) == (None, "ambiguous_capture") and gs.get_fen() == fen
    # a vanished piece with no capture available -> falls through to no_valid_change (game_state.py:99-112)
    gs.reset()
    assert gs.process_occupancy_change(_moved(gs, [(4, 1)], [])) == (None, "no_valid_change")
    # 1-1 move of a pawn to the last rank auto-promotes to a queen (game_state.py:187-193)
    gs.set_fen("4k3/P7/8/8/8/8/8/4K3 w - - 0 1")
    move, status = gs.process_occupancy_change(_moved(gs, [(0, 6)], [(0, 7)]))
    assert (move.uci(), status) == ("a7a8q", "move_confirmed") and gs.get_fen() == "Q3k3/8/8/8/8/8/8/4K3 b - - 0 1"
    # 2-2 pattern that is not castling
    gs.reset()
    assert gs.process_occupancy_change(_moved(gs, [(4, 1), (3, 1)], [(4, 3), (3, 3)])) == (None, "no_valid_change")
    assert len(gs.get_legal_moves()) == 20 and {m.uci() for m in gs.get_legal_moves_from(6, 0)} == {"g1f3", "g1h3"}


def test_scripted_game_through_occupancy_words():
    """The pipeline's occupancy words drive the game: ROI-numbered bits -> square bits -> moves -> FEN with piece
    identity (what generate_fen on occupancy alone cannot give, game_state.py:7-8)."""
    from chessboard_vision_amd import synth as S
    gs = GameState()
    played = []
    for ply in range(1, len(S.SCRIPT) + 1):
        pos = S.position_after(ply)
        roi_bits = 0
        for (f, r) in pos:
            roi_bits |= 1 << ((7 - r) * 8 + f)
        move, status = gs.process_occupancy_bits(chess.roi_bits_to_squares(roi_bits))
        assert move is not None and status in ("move_confirmed", "castling_confirmed"), (ply, status)
        played.append(move.uci())
    assert played == ["e2e4", "e7e5", "g1f3", "b8c6", "f1b5", "a7a6", "b5a4", "g8f6", "e1g1", "f8e7", "f1e1", "b7b5",
                      "a4b3", "d7d6", "c2c3", "e8g8"]
    assert gs.get_fen() == "r1bq1rk1/2p1bppp/p1np1n2/1p2p3/4P3/1BP2N2/PP1P1PPP/RNBQR1K1 w - - 1 9"


def test_infer_move_and_stable_tracker():
    """GameSession._infer_move / _process_stable_move (game_session.py:181-265) on the native engine."""
    from chessboard_vision_amd.game_state import StableMoveTracker, smart_scan_squares
    gs = GameState()
    now = [100.0]
    events = []
    tr = StableMoveTracker(gs, clock=lambda: now[0], after_move=lambda: events.append("synced"))
    start = gs.get_board_occupancy()
    assert tr.infer_move(start) == (None, 0)
    e4 = _moved(gs, [(4, 1)], [(4, 3)])
    assert tr.infer_move(e4)[0].uci() == "e2e4"
    # 19 stable frames are not enough, the 20th pushes the move
    for i in range(19):
        assert tr.process(e4) is None
    assert tr.process(e4).uci() == "e2e4" and events == ["synced"] and gs.get_turn_name() == "black"
    # cooldown: the reply is seen at once but accepted only 2 s after the previous move
    e5 = _moved(gs, [(4, 6)], [(4, 4)])
    for i in range(25):
        assert tr.process(e5) is None
    now[0] += 2.5
    assert tr.process(e5).uci() == "e7e5"
    # a hand over the board (more than 4 differing squares) resets the count; NOISE_ACTIVE blocks acceptance
    now[0] += 10
    nf3 = _moved(gs, [(6, 0)], [(5, 2)])
    for i in range(19):
        tr.process(nf3)
    hand = set(nf3) - {(0, 1), (1, 1), (2, 1)} | {(3, 3), (3, 4)}
    assert tr.process(hand) is None and tr.stable_count == 0
    for i in range(19):
        assert tr.process(nf3) is None
    assert tr.process(nf3, noise_active=True) is None
    assert tr.process(nf3).uci() == "g1f3"
    # ambiguity: after 1.e4 d5 a vanished e4 pawn with d5 still occupied is exd5 only; a vanished piece that could
    # capture two ways yields two candidates and no move
    amb = GameState()
    amb.set_fen("4k3/8/8/2p1p3/3P4/8/8/4K3 w - - 0 1")
    t2 = StableMoveTracker(amb, clock=lambda: 1e9)
    assert t2.infer_move(_moved(amb, [(3, 3)], [])) == (None, 2)
    # promotion is inferred as a queen
    pr = GameState()
    pr.set_fen("4k3/P7/8/8/8/8/8/4K3 w - - 0 1")
    assert StableMoveTracker(pr).infer_move(_moved(pr, [(0, 6)], [(0, 7)]))[0].uci() == "a7a8q"
    # smart scan set of the start position: 32 occupied squares + the (file, 7 - rank) images of ranks 3 and 4
    sq = smart_scan_squares(GameState())
    assert len(sq) == 32 + 16 and (0, 5) in sq and (0, 4) in sq and (0, 2) not in sq


def test_board_surface_used_by_the_session_layer():
    """push_uci (lichess_session.py sync), assignable turn (test_race_condition.py:45), Piece symbols, str()."""
    b = chess.Board()
    assert b.push_uci("e2e4").uci() == "e2e4" and b.turn == chess.BLACK
    with pytest.raises(ValueError):
        b.push_uci("e2e4")
    b.turn = chess.WHITE
    assert b.turn == chess.WHITE and b.fen().split()[1] == "w"
    assert b.piece_at(chess.E4).symbol() == "P" and b.piece_at(chess.E8).symbol() == "k" and b.piece_at(chess.E3) is None
    assert str(chess.Board()).splitlines()[0] == "r n b q k b n r" and len(chess.SQUARES) == 64
    assert chess.square_name(chess.square(4, 3)) == "e4" and chess.parse_square("h8") == 63
