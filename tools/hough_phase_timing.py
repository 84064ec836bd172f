"""Per-phase cycle counts of k_hough on the bench scene (needs a build with -DHG_TIMING)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chessboard_vision_amd import synth as S
from chessboard_vision_amd.stream import BoardPipeline

n = 32
p = BoardPipeline(1920, 1080, n)
p.configure(S.scaled_corners(1920, 1080), profile=S.SHIPPED_PROFILE, grid_lines=(S.CALIB_GRID_X, S.CALIB_GRID_Y),
            min_radius_ratio=0.25, chunk=32, lanes=1, use_hough=2)
p.synth(0, n, scene="dim", frames_per_ply=4)
p.run(0, n)
p.results(0, n)
rows = []
for slot in (0, 13, 31):
    for r in p.hough(slot):
        if r.flags & 2:
            continue
        t = [r.circles[2 + i // 4][i % 4] for i in range(8)] + [r.circles[4][0]]
        rows.append(t + [r.n_edges, r.n_centres, r.circles[4][1], r.circles[4][2], r.found])
a = np.array(rows)
names = ["P0 load", "P1 sobel", "P2 nms", "P3 hyst", "P4 vote", "P5 maxima", "P6 radius", "P7 pick", "(tail)", "edges", "centres", "weak", "circ", "found"]
for i, nm in enumerate(names):
    print("%-10s mean %10.1f  p50 %10.1f  max %10.1f" % (nm, a[:, i].mean(), np.median(a[:, i]), a[:, i].max()))
print("total cycles mean", a[:, :9].sum(axis=1).mean())
