"""Timeline of ONE 1080p frame through the pipeline (configs[1], the live-camera case): run under rocprofv3 to get the
kernel trace, then analyse it.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 tools/single_frame_trace.py run
    python tools/single_frame_trace.py show out/t_kernel_trace.csv
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import time
    from chessboard_vision_amd import _native as N
    from chessboard_vision_amd import synth as S
    from chessboard_vision_amd.stream import BoardPipeline
    ctx = N.context(0)
    w, h = 1920, 1080
    pipe = BoardPipeline(w, h, 4, ctx)
    pipe.configure(S.scaled_corners(w, h), profile=S.SHIPPED_PROFILE, grid_lines=(S.CALIB_GRID_X, S.CALIB_GRID_Y), **S.SHIPPED_DETECTOR)
    pipe.synth(0, 4, scene="dim")
    pipe.run(0, 1)
    pipe.calibrate_changes(0)
    pipe.reset_state()
    ts = []
    for i in range(12):
        a = time.perf_counter()
        pipe.run(i % 4, 1)
        pipe.results(i % 4, 1)
        ts.append((time.perf_counter() - a) * 1e3)
    print("run + results of one frame, ms:", " ".join("%.3f" % t for t in ts))


def show(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    # the last frame: from the last k_color_lab_hist (first kernel of the chain) on, with a reset / fill launch right before it
    start = max(i for i, x in enumerate(rows) if x[2] == "k_color_lab_hist")
    while start > 0 and rows[start][0] - rows[start - 1][1] < 40000 and ("fillBuffer" in rows[start - 1][2] or rows[start - 1][2] == "k_reset_aux"):
        start -= 1
    rows = rows[start:]
    t0 = rows[0][0]
    busy = 0
    prev_end = t0
    print("%-28s %10s %10s %10s" % ("kernel", "start us", "dur us", "gap us"))
    for s, e, n in rows:
        print("%-28s %10.1f %10.1f %10.1f" % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        busy += e - s
        prev_end = max(prev_end, e)
    span = prev_end - t0
    print("span %.1f us, kernels %.1f us (%.0f %%), gaps %.1f us over %d launches" % (span / 1e3, busy / 1e3, 100.0 * busy / span, (span - busy) / 1e3, len(rows)))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
