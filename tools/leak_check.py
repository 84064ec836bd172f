"""Create / configure / run / close pipelines and detector objects in a loop and watch the device's free memory and the
process's RSS: neither may creep.  One-off (GPU box)."""
import os
import resource
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from chessboard_vision_amd import synth as S  # noqa: E402
from chessboard_vision_amd.frame_enhancer import ImageEnhancer  # noqa: E402
from chessboard_vision_amd.piece_detector import PieceDetector  # noqa: E402
from chessboard_vision_amd.change_detector import ChangeDetector  # noqa: E402
from chessboard_vision_amd.grid_extractor import GridExtractor  # noqa: E402
from chessboard_vision_amd.board_detection import warp_image  # noqa: E402
from chessboard_vision_amd.stream import BoardPipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(1)
w, h = 640, 480
frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
pts = S.scaled_corners(w, h)
log = []
for it in range(n):
    p = BoardPipeline(w, h, 16)
    p.configure(pts, profile=S.SHIPPED_PROFILE, chunk=4, lanes=1 + it % 3, enhance_region=bool(it % 2), keep_enhanced=(it % 5 == 0), **S.SHIPPED_DETECTOR)
    p.synth(0, 16, scene="dim")
    p.run(0, 16)
    p.results(0, 16)
    if it % 3 == 0:
        p.configure(pts, profile={}, chunk=8)  # reconfigure in place
        p.run(0, 8)
        p.results(0, 8)
    ring = p.host_ring() if hasattr(p, "host_ring") and it % 4 == 0 else None
    p.close()
    e = ImageEnhancer()
    e.profile = S.SHIPPED_PROFILE
    enh = e.process_pipeline(frame)
    warped, _, _ = warp_image(enh, pts)
    sq = GridExtractor().split_board(warped)
    pd, cd = PieceDetector(), ChangeDetector()
    pd.detect_all_pieces(sq)
    cd.calibrate(sq)
    cd.detect_changes_detailed(sq)
    del pd, cd, e
    free, total = torch.cuda.mem_get_info()
    rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024
    log.append((free >> 20, rss))
    if it % 10 == 9:
        print("it %d: device free %d MiB, max RSS %d MiB" % (it + 1, free >> 20, rss), flush=True)
lo = min(f for f, _ in log[5:])
hi = max(f for f, _ in log[5:])
print("device free after warm-up: %d..%d MiB (spread %d); max RSS %d -> %d MiB" % (lo, hi, hi - lo, log[5][1], log[-1][1]))
