"""Board map -> FEN text; same functions and outputs as fen_generator.py:12-89."""

COLUMNS = "abcdefgh"
ROWS = "12345678"

PIECE_TO_FEN = {
    "white-pawn": "P", "white-knight": "N", "white-bishop": "B", "white-rook": "R", "white-queen": "Q", "white-king": "K",
    "black-pawn": "p", "black-knight": "n", "black-bishop": "b", "black-rook": "r", "black-queen": "q", "black-king": "k",
}


def get_chess_square(x, y, board_size):
    """Pixel (x, y) of the warped board -> ('e4', (grid_x, grid_y)); grid_y = 0 is rank 8."""
    sq = board_size // 8
    gx, gy = x // sq, y // sq
    if not (0 <= gx < 8 and 0 <= gy < 8):
        return "fora dos limites", (-1, -1)
    return COLUMNS[gx] + ROWS[7 - gy], (gx, gy)


def map_detections_to_board(detections, board_size):
    """Keep the most confident detection per square."""
    board_map = {}
    for det in detections:
        cx, cy = det["center"]
        _, (gx, gy) = get_chess_square(cx, cy, board_size)
        if gx == -1 or gy == -1:
            continue
        entry = {"fen": PIECE_TO_FEN.get(det["class"], "?"), "conf": det["conf"], "class": det["class"]}
        cur = board_map.get((gx, gy))
        if cur is None or entry["conf"] > cur["conf"]:
            board_map[(gx, gy)] = entry
    return board_map


def generate_fen(board_map, current_turn="w"):
    grid = [["" for _ in range(8)] for _ in range(8)]
    for (gx, gy), data in board_map.items():
        grid[gy][gx] = data["fen"]
    ranks = []
    for row in grid:
        txt, run = "", 0
        for cell in row:
            if cell == "":
                run += 1
                continue
            if run:
                txt += str(run)
                run = 0
            txt += cell
        if run:
            txt += str(run)
        ranks.append(txt)
    return "/".join(ranks) + f" {current_turn} - - 0 1"


def occupancy_to_grid(occupied):
    """{(file, rank)} with a1 = (0,0) -> 8x8 list of 0/1, row 0 = rank 8."""
    return [[1 if (f, 7 - gy) in occupied else 0 for f in range(8)] for gy in range(8)]
