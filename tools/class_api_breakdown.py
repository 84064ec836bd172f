"""Where the time of the class-API calls goes (one 1080p host frame per call): Python glue, allocation, the C call."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from chessboard_vision_amd import _native as N, synth as S
from chessboard_vision_amd.board_detection import get_perspective_transform, warp_image, warp_perspective
from chessboard_vision_amd.frame_enhancer import ImageEnhancer
from chessboard_vision_amd.grid_extractor import SmartGridExtractor
from chessboard_vision_amd.piece_detector import PieceDetector
from chessboard_vision_amd.stream import BoardPipeline

def med(fn, n=60, warm=5):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return round(statistics.median(ts) * 1e6, 1)

w, h = 1920, 1080
pts = S.scaled_corners(w, h)
p = BoardPipeline(w, h, 4); p.configure(pts, profile=S.SHIPPED_PROFILE); p.synth(0, 4, scene="dim")
frames = [p.download(0, i) for i in range(4)]
f = frames[0]
c = N.context()
out = {}
out["np.float32(points) x2 + getPerspectiveTransform"] = med(lambda: get_perspective_transform(np.float32(pts), np.float32([[0, 0], [620, 0], [0, 620], [620, 620]])))
out["np.empty board"] = med(lambda: np.empty((620, 620, 3), np.uint8))
M = get_perspective_transform(np.float32(pts), np.float32([[0, 0], [620, 0], [0, 620], [620, 620]]))
board = np.empty((620, 620, 3), np.uint8)
Mc = np.ascontiguousarray(M, dtype=np.float64)
out["cbv_warp_perspective (preallocated out)"] = med(lambda: c.check(c.lib.cbv_warp_perspective(c.h, N.ptr(f), w, h, f.strides[0], N.ptr(Mc), 620, 620, 0, N.ptr(board), board.strides[0])))
out["warp_perspective()"] = med(lambda: warp_perspective(f, M, (620, 620)))
out["warp_image()"] = med(lambda: warp_image(f, pts))
k = [0]
def rot():
    k[0] += 1
    return warp_image(frames[k[0] & 3], pts)
out["warp_image() rotating 4 frames"] = med(rot)
e = ImageEnhancer(); e.profile = S.SHIPPED_PROFILE
enh_out = np.empty_like(f)
prm = e._params()
out["cbv_process_pipeline (preallocated out)"] = med(lambda: c.check(c.lib.cbv_process_pipeline(c.h, N.ptr(f), w, h, f.strides[0], prm, N.ptr(enh_out), enh_out.strides[0])))
out["process_pipeline()"] = med(lambda: e.process_pipeline(f))
out["np.empty frame"] = med(lambda: np.empty((h, w, 3), np.uint8))
out["e._params()"] = med(lambda: e._params())
ge = SmartGridExtractor(); ge.grid_lines_x, ge.grid_lines_y = list(S.CALIB_GRID_X), list(S.CALIB_GRID_Y)
warped = warp_image(e.process_pipeline(f), pts)[0]
out["split_board"] = med(lambda: ge.split_board(warped))
sq = ge.split_board(warped)
det = PieceDetector()
det.update_references(sq)
out["detect_all_pieces (steady, check=None)"] = med(lambda: det.detect_all_pieces(sq))
chk = set(S.position_for_frame(0).keys())
out["detect_all_pieces (steady, 32 squares to check)"] = med(lambda: det.detect_all_pieces(sq, squares_to_check=chk))
from chessboard_vision_amd._squares import plan_of
img, lay = plan_of(sq)
det._prm.check_given = 0
out["cbv_squares_detect_all C call + tolist"] = med(lambda: det._state.detect_all(img, lay, det._prm))
det._prm.check_given = 1; det._prm.check = lay.mask(chk)
out["cbv_squares_detect_all C call + tolist (32 to check)"] = med(lambda: det._state.detect_all(img, lay, det._prm))
out["plan_of(SquareDict)"] = med(lambda: plan_of(sq))
out["plan_of(plain dict)"] = med(lambda: plan_of(dict(sq)))
for k_, v in out.items():
    print("%-60s %8.1f us" % (k_, v))
