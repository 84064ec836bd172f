"""Per-workgroup phase timeline of k_sharpen_box (needs a build with SH_TIMING=1): how much of a CU's time has some
workgroup computing, how much has all resident workgroups waiting for their tile."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from chessboard_vision_amd import _native as N  # noqa: E402
from chessboard_vision_amd import synth as S  # noqa: E402
from chessboard_vision_amd.stream import BoardPipeline  # noqa: E402

n = 32
p = BoardPipeline(1920, 1080, n)
p.configure(S.scaled_corners(1920, 1080), profile=S.SHIPPED_PROFILE, chunk=32, lanes=1, use_hough=0)
p.synth(0, n, scene="dim")
p.run(0, n)
p.results(0, 1)
lib = N.load()
lib.cbv_debug_sharpen_stamps.restype = C.c_int
lib.cbv_debug_sharpen_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.cbv_debug_sharpen_stamps(None, 0, 1)
p.run(0, n)
p.results(0, 1)
buf = np.zeros((1 << 16, 5), np.uint64)
k = lib.cbv_debug_sharpen_stamps(buf.ctypes.data, 1 << 16, 1)
a = buf[:k].astype(np.int64)
hw = a[:, 0]
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0xF) << 7)   # cu_id, se_id, (xcc/sh bits)
t = a[:, 1:] - a[:, 1].min()
print("workgroups", k, "span clk", int(t.max()), "distinct CU keys", len(np.unique(cu)))
d = np.diff(t, axis=1)
for name, col in (("load issue -> issued", 0), ("wait for tile (barrier)", 1), ("compute + store + reduce", 2)):
    print("%-28s mean %8.0f  p50 %8.0f  p90 %8.0f clk" % (name, d[:, col].mean(), np.median(d[:, col]), np.percentile(d[:, col], 90)))
print("lifetime mean %.0f clk" % (t[:, 3] - t[:, 0]).mean())
# per CU: fraction of the busy span in which at least one workgroup is in its compute phase
fr_comp, fr_any, res = [], [], []
for c in np.unique(cu):
    m = cu == c
    tt = t[m]
    lo, hi = tt[:, 0].min(), tt[:, 3].max()
    ev = []
    for r in tt:
        ev.append((r[2], 1))
        ev.append((r[3], -1))
    ev.sort()
    cur, last, busy = 0, lo, 0
    for x, dlt in ev:
        if cur > 0:
            busy += x - last
        last = x
        cur += dlt
    fr_comp.append(busy / max(1, hi - lo))
    # average residency
    res.append((tt[:, 3] - tt[:, 0]).sum() / max(1, hi - lo))
print("per CU: some workgroup computing %.1f %% of the time; mean resident workgroups %.2f" % (100 * np.mean(fr_comp), np.mean(res)))
