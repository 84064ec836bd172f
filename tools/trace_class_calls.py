"""A few process_pipeline / warp_image / detect_all_pieces calls on one 1080p host frame, for
`rocprofv3 --kernel-trace --memory-copy-trace`: the GPU-side timeline of one class-API frame (tools/class_timeline.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chessboard_vision_amd import synth as S
from chessboard_vision_amd.board_detection import warp_image
from chessboard_vision_amd.frame_enhancer import ImageEnhancer
from chessboard_vision_amd.grid_extractor import SmartGridExtractor
from chessboard_vision_amd.piece_detector import PieceDetector
from chessboard_vision_amd.stream import BoardPipeline
w, h = 1920, 1080
pts = S.scaled_corners(w, h)
p = BoardPipeline(w, h, 2); p.configure(pts, profile={}); p.synth(0, 2, scene="dim")
f = p.download(0, 0)
p.close()
e = ImageEnhancer(); e.profile = dict(S.SHIPPED_PROFILE)
ge = SmartGridExtractor(); ge.grid_lines_x, ge.grid_lines_y = list(S.CALIB_GRID_X), list(S.CALIB_GRID_Y)
det = PieceDetector()
chk = set(S.position_for_frame(0).keys())
for _ in range(12):
    enh = e.process_pipeline(f)
    warped = warp_image(enh, pts)[0]
    det.detect_all_pieces(ge.split_board(warped), squares_to_check=chk)
