"""Who runs beside whom in the timed region: from a rocprofv3 kernel trace of the bench, per queue (= lane / scan stream)
the kernels of the LAST full step, how long each takes there against its time in a single-lane pass, how much of the
step some bilateral launch is on the chip, and the windows in which none is.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --steps 4 --warmup 2 --no-profile-pass \
        --cpu-frames 0 --no-4k --no-noise-leg --no-class-api --no-region-leg
    python tools/lane_timeline.py out/t_kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.split("(")[0].replace("void ", "").split("<")[0]


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Queue_Id"]), int(r["Grid_Size_Z"] or 1)))
    rows.sort()
    # batched bilateral launches (64 frames): the timed region's
    bil = [x for x in rows if x[2] == "k_bilateral" and x[1] - x[0] > 500_000]
    if len(bil) < 16:
        sys.exit("no batched bilateral launches in the trace")
    last = bil[-8:]  # one 512-frame step = 8 chunks of 64
    t0, t1 = last[0][0], last[-1][1]
    span = (t1 - t0) / 1e3
    print("window: the last 8 batched bilateral launches, %.1f us (%.2f us per frame over 512 frames)" % (span, span / 512))
    # union of bilateral intervals
    iv = sorted((s, e) for s, e, *_ in last)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    gaps = []
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e) / 1e3)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("some bilateral launch on the chip: %.1f us = %.1f %% of the window; windows with none: %s us" %
          (busy / 1e3, 100.0 * busy / (t1 - t0), ", ".join("%.0f" % g for g in gaps)))
    overl = 0
    for i in range(len(iv)):
        for j in range(i + 1, len(iv)):
            overl += max(0, min(iv[i][1], iv[j][1]) - max(iv[i][0], iv[j][0]))
    print("two bilateral launches at once: %.1f us" % (overl / 1e3))
    # per kernel: durations inside the window, by queue
    per = defaultdict(list)
    for s, e, n, q, z in rows:
        if s >= t0 and e <= t1 + 2_000_000 and e - s > 20_000:
            per[n].append((e - s) / 1e3)
    print("%-26s %6s %10s %10s %10s" % ("kernel (in the window)", "n", "avg us", "min us", "max us"))
    for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print("%-26s %6d %10.1f %10.1f %10.1f" % (n, len(v), sum(v) / len(v), min(v), max(v)))
    # the lanes' timelines
    qs = sorted({x[3] for x in last})
    for q in sorted({x[3] for x in rows if t0 <= x[0] <= t1}):
        line = [(s, e, n) for s, e, n, qq, z in rows if qq == q and s >= t0 and s <= t1 and e - s > 20_000]
        print("queue %d:" % q, " ".join("%s[%.0f+%.0f]" % (n.replace("k_", "")[:8], (s - t0) / 1e3, (e - s) / 1e3) for s, e, n in line[:40]))


if __name__ == "__main__":
    main(sys.argv[1])
