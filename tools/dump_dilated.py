import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from chessboard_vision_amd.board_detection import find_chessboard_corners
from helpers import oracle_frame
for (w, h) in ((1280, 720), (1920, 1080)):
    f = oracle_frame(w, h, "normal", frame_idx=0)
    got, dil = find_chessboard_corners(f, debug=True)
    print(w, h, got.reshape(-1, 2).tolist() if got.size else None)
    np.save("gpurun_out/dil_%d.npy" % w, np.packbits(dil > 0, axis=1))
