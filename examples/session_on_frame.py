"""The reference's own per-frame loop, call for call, on the drop-in classes: what GameSession.on_frame does with every
camera frame (game_session.py:124-180) — warp_image, split_board, the smart-scan set from the rules engine,
detect_all_pieces, NoiseHandler.process, the stable-occupancy rule — with synthetic 1080p frames of a scripted game
standing in for the camera (host numpy arrays, exactly what cv2.VideoCapture.read() yields).  Prints every recognised
move with the FEN and, at the end, what one frame cost.

    python examples/session_on_frame.py           # needs the built library and a gfx950 GPU
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chessboard_vision_amd import synth as S  # noqa: E402
from chessboard_vision_amd.board_detection import warp_image  # noqa: E402
from chessboard_vision_amd.frame_enhancer import ImageEnhancer  # noqa: E402
from chessboard_vision_amd.game_state import GameState, smart_scan_squares  # noqa: E402
from chessboard_vision_amd.grid_extractor import SmartGridExtractor  # noqa: E402
from chessboard_vision_amd.noise_handler import NoiseHandler, NoiseState  # noqa: E402
from chessboard_vision_amd.piece_detector import PieceDetector  # noqa: E402
from chessboard_vision_amd.stream import BoardPipeline  # noqa: E402

W, H, FRAMES_PER_PLY = 1920, 1080, 30
points_ordered = S.scaled_corners(W, H)

# the camera: frames of the scripted game, rendered on the device and fetched as host arrays
cam = BoardPipeline(W, H, FRAMES_PER_PLY)
cam.configure(points_ordered, profile={})

enhancer = ImageEnhancer()                      # game_session.py:86
enhancer.profile = dict(S.SHIPPED_PROFILE)      # color_profile.json of the reference
grid = SmartGridExtractor()
grid.grid_lines_x, grid.grid_lines_y = list(S.CALIB_GRID_X), list(S.CALIB_GRID_Y)
piece_detector = PieceDetector()
piece_detector.min_radius_ratio, piece_detector.max_radius_ratio = S.SHIPPED_DETECTOR["min_radius_ratio"], S.SHIPPED_DETECTOR["max_radius_ratio"]
noise = NoiseHandler()
game = GameState()
stable_occupancy, stable_count, frame_count = set(), 0, 0
t_frames, n_frames = 0.0, 0

for ply in range(len(S.SCRIPT) + 1):
    cam.synth(0, FRAMES_PER_PLY, frame0=ply * FRAMES_PER_PLY, scene="dim", frames_per_ply=FRAMES_PER_PLY)
    for i in range(FRAMES_PER_PLY):
        img = cam.download(0, i)                # "success, img = cap.read()"
        t0 = time.perf_counter()
        frame_count += 1
        img = enhancer.process_pipeline(img)    # the north star's composed chain puts the enhancement in front
        warped, _, board_size = warp_image(img, points_ordered)
        squares = grid.split_board(warped)
        # smart scan (game_session.py:130-152): a full scan every 30th frame, else occupied squares + legal destinations
        squares_to_check = None if frame_count % 30 == 0 else smart_scan_squares(game)
        piece_detections, visual_changes = piece_detector.detect_all_pieces(squares, use_delta=True, squares_to_check=squares_to_check)
        vision_occupied = {pos for pos, info in piece_detections.items() if info["has_piece"]}
        noise_state, _ = noise.process(visual_changes)
        # _process_stable_move's stability rule (game_session.py:181-205)
        expected = game.get_board_occupancy()
        if len(expected - vision_occupied) + len(vision_occupied - expected) > 4:
            stable_count, stable_occupancy = 0, set()
        elif stable_occupancy == vision_occupied:
            stable_count += 1
        else:
            stable_occupancy, stable_count = set(vision_occupied), 1
        t_frames += time.perf_counter() - t0
        n_frames += 1
        if stable_count == 20 and noise_state != NoiseState.NOISE_ACTIVE and vision_occupied != expected:
            move, status = game.process_occupancy_change(vision_occupied)
            print("frame %4d  %-5s %-20s %s" % (frame_count, move.uci() if move else "-", status, game.get_fen()))
            if move:
                piece_detector.update_references(squares)   # game_session.py:219-223
                noise.reset()
print("%d frames, %.3f ms per frame through process_pipeline -> warp_image -> split_board -> detect_all_pieces -> NoiseHandler "
      "(host arrays in, dicts out); final FEN %s" % (n_frames, t_frames / n_frames * 1e3, game.get_fen()))
