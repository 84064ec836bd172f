"""From a camera-like stream to chess moves on one MI355X: synthetic 1080p frames of a scripted game go through
enhance -> warp -> 64-square detect (BoardPipeline) and the NoiseHandler state machine (on the device); an
occupancy that stayed put for 20 frames is handed to GameState.process_occupancy_change, which recognises the
move and keeps the FEN with piece identity.  (The session layer's own rule, StableMoveTracker, reproduces the
reference's _infer_move including its habit of calling a move ambiguous whenever the moved piece could also
have captured something; process_occupancy_change does not have that problem.)

    python examples/stream_to_moves.py            # needs the built library and a gfx950 GPU
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chessboard_vision_amd import synth as S  # noqa: E402
from chessboard_vision_amd.game_state import GameState  # noqa: E402
from chessboard_vision_amd.stream import BoardPipeline  # noqa: E402

W, H, FRAMES_PER_PLY, BATCH = 1920, 1080, 30, 120
PLIES = len(S.SCRIPT)

pipe = BoardPipeline(W, H, BATCH)
pipe.configure(S.scaled_corners(W, H), profile=S.SHIPPED_PROFILE, grid_lines=(S.CALIB_GRID_X, S.CALIB_GRID_Y),
               enhance_region=True,  # a fixed camera over one board: enhance what the warp samples (identical results)
               **S.SHIPPED_DETECTOR)
game = GameState()
stable, last = 0, None

t0 = time.perf_counter()
frame = 0
total = FRAMES_PER_PLY * (PLIES + 1)
while frame < total:
    n = min(BATCH, total - frame)
    pipe.synth(0, n, frame0=frame, scene="dim", frames_per_ply=FRAMES_PER_PLY)   # stands in for the camera
    pipe.run(0, n)
    results, noise = pipe.results(0, n), pipe.noise_results(0, n)
    for i in range(n):
        state, _ = noise[i]
        occ = pipe.occupied(results[i])
        stable = stable + 1 if occ == last else 1
        last = occ
        if stable == 20 and state.name != "NOISE_ACTIVE" and occ != game.get_board_occupancy():
            move, status = game.process_occupancy_change(occ)
            print("frame %4d  %-5s %-20s %s" % (frame + i, move.uci() if move else "-", status, game.get_fen()))
    frame += n
dt = time.perf_counter() - t0
print("%d frames in %.2f s (%.0f frames/s including synthesis and host logic); final FEN %s" % (total, dt, total / dt, game.get_fen()))
