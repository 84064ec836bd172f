/*
 * cbv_chess.h — C-ABI of the game-rules row: what
 * game_state.py needs from python-chess (an absent third-party dependency of
 * the reference, requirements.txt: "chess"), restated as host C++, plus
 * GameState.process_occupancy_change (game_state.py:40-112) itself so that the
 * u64 occupancy words of cbv_frame_result can be turned into moves without
 * leaving native code.  Host only: no function here touches the GPU.
 *
 * Squares are python-chess indices: a1 = 0, b1 = 1, ..., h8 = 63
 * (chess.square(file, rank) = rank * 8 + file).  Occupancy words use the same
 * numbering (bit s = square s); note that cbv_frame_result numbers squares by
 * ROI (8 * row + col of the warped image) — cbv_roi_bits_to_squares converts.
 *
 * Moves are 16-bit: from | to << 6 | promotion << 12, promotion being the
 * python-chess piece type (0 none, 2 knight, 3 bishop, 4 rook, 5 queen).
 * Castling is the king's move (e1g1), as python-chess encodes it for
 * standard chess.
 */
#ifndef CBV_CHESS_H
#define CBV_CHESS_H
#include <stdint.h>

#ifndef CBV_API
#define CBV_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cbv_board cbv_board;
typedef uint16_t cbv_move;

#define CBV_MOVE_NONE 0xFFFFu
#define CBV_MAX_MOVES 256

/* chess.Board() / board.reset() / board.set_fen(fen) / board.fen()   (game_state.py:5,7,197-203) */
CBV_API cbv_board* cbv_board_create(void);
CBV_API void cbv_board_destroy(cbv_board* b);
CBV_API void cbv_board_reset(cbv_board* b);
/* returns 0, or -1 when the text is not a FEN (the board is left unchanged) */
CBV_API int cbv_board_set_fen(cbv_board* b, const char* fen);
/* python-chess Board.fen(): en passant square only when an en passant capture is legal; castling
 * letters only for rights whose king and rook still stand on their squares.  Returns the length. */
CBV_API int cbv_board_fen(const cbv_board* b, char* out, int cap);
/* board.turn: 1 = white (chess.WHITE is True), 0 = black */
CBV_API int cbv_board_turn(const cbv_board* b);
CBV_API void cbv_board_set_turn(cbv_board* b, int white); /* board.turn = chess.WHITE (test_race_condition.py:45) */
/* board.piece_at(sq): 0 = empty, else piece_type (1 pawn .. 6 king) | 8 when black */
CBV_API int cbv_board_piece_at(const cbv_board* b, int square);
/* bit s = a piece stands on square s   (get_board_occupancy, game_state.py:26-38) */
CBV_API uint64_t cbv_board_occupancy(const cbv_board* b);
/* list(board.legal_moves) in python-chess's generation order; returns the count */
CBV_API int cbv_board_legal_moves(const cbv_board* b, cbv_move* out, int cap);
CBV_API int cbv_board_is_legal(const cbv_board* b, cbv_move m);      /* move in board.legal_moves */
CBV_API int cbv_board_is_capture(const cbv_board* b, cbv_move m);    /* board.is_capture(move) */
CBV_API int cbv_board_is_en_passant(const cbv_board* b, cbv_move m); /* board.is_en_passant(move) */
CBV_API int cbv_board_is_check(const cbv_board* b);
/* board.push(move) (not validated, like python-chess) / board.pop() / len(board.move_stack) / board.peek() */
CBV_API int cbv_board_push(cbv_board* b, cbv_move m);
CBV_API cbv_move cbv_board_pop(cbv_board* b);
CBV_API int cbv_board_ply(const cbv_board* b);
CBV_API cbv_move cbv_board_peek(const cbv_board* b);
/* leaf count of the legal move tree (the published perft numbers are the known-answer test of the generator) */
CBV_API uint64_t cbv_board_perft(cbv_board* b, int depth);

/* GameState.process_occupancy_change (game_state.py:40-112): compare the vision occupancy with the board,
 * recognise normal move / castling / en passant / capture, push the move when it is legal. */
enum {
    CBV_GAME_NO_VALID_CHANGE = 0, CBV_GAME_MOVE_CONFIRMED, CBV_GAME_ILLEGAL_MOVE, CBV_GAME_CASTLING_CONFIRMED,
    CBV_GAME_EN_PASSANT_CONFIRMED, CBV_GAME_CAPTURE_CONFIRMED, CBV_GAME_AMBIGUOUS_CAPTURE
};
CBV_API int cbv_game_process_occupancy(cbv_board* b, uint64_t vision_occupancy, cbv_move* move_out);
/* the status strings the reference returns, indexed by the codes above */
CBV_API const char* cbv_game_status_name(int status);

/* GameSession._infer_move (game_session.py:229-265): the moves that explain the difference between the board and
 * the vision occupancy — (vanished origin, appeared destination) pairs that are legal (queen promotion tried when
 * the plain move is not), plus legal captures from a vanished origin onto a square vision sees occupied.  Returns
 * the number of distinct candidates; *move_out is set only when it is exactly one.  The board is not changed. */
CBV_API int cbv_game_infer_move(const cbv_board* b, uint64_t vision_occupancy, cbv_move* move_out);

/* ROI-numbered bits of cbv_frame_result (bit 8 * row + col, row 0 = rank 8 of the warped board, or its 180-degree
 * turn when the pipeline was configured with rot180) -> python-chess square bits (grid_extractor.py:31-40 numbering: file = col, rank = 7 - row) */
CBV_API uint64_t cbv_roi_bits_to_squares(uint64_t roi_bits);

#ifdef __cplusplus
}
#endif
#endif /* CBV_CHESS_H */
