"""How much do kernels of the pipeline's two lanes (HIP streams) overlap in time?  Reads a rocprofv3 --kernel-trace CSV
of a bench run and reports, for the timed launches: the union of busy time, the time in which kernels of >= 2 different
queues run at once, and per-kernel durations next to their stand-alone durations (single-lane profile pass).

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile-pass
    python tools/lane_overlap.py out/t_kernel_trace.csv
"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    if not name.startswith("k_") or name == "k_synth":
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], name))
rows.sort()
# keep the steady part: from the first k_bilateral of the last third of the trace
t_lo = rows[len(rows) // 3][0]
rows = [x for x in rows if x[0] >= t_lo]
ev = []
for s, e, q, n in rows:
    ev.append((s, 1, q))
    ev.append((e, -1, q))
ev.sort()
active = collections.Counter()
last = ev[0][0]
busy = multi = 0
for t, d, q in ev:
    nq = sum(1 for v in active.values() if v > 0)
    if nq >= 1:
        busy += t - last
    if nq >= 2:
        multi += t - last
    active[q] += d
    last = t
span = rows[-1][1] - rows[0][0]
print("kernels: %d on %d queues; span %.3f ms, some kernel running %.3f ms (%.1f %%), kernels of >= 2 queues at once %.3f ms (%.1f %% of the span)"
      % (len(rows), len({q for _, _, q, _ in rows}), span / 1e6, busy / 1e6, 100.0 * busy / span, multi / 1e6, 100.0 * multi / span))
# how long are TWO launches of the dominant kernel in flight at once (both lanes in the bilateral: no complementary work)?
bl = sorted((s, e) for s, e, q, n in rows if n == "k_bilateral")
both = 0
for i, (s1, e1) in enumerate(bl):
    for s2, e2 in bl[i + 1:]:
        if s2 >= e1:
            break
        both += min(e1, e2) - s2
one = sum(e - s for s, e in bl) - 2 * both
print("k_bilateral in flight: one launch %.3f ms (%.1f %% of the span), two launches at once %.3f ms (%.1f %%), none %.1f %%"
      % (one / 1e6, 100.0 * one / span, both / 1e6, 100.0 * both / span, 100.0 * (span - one - both) / span))
per = collections.defaultdict(list)
for s, e, q, n in rows:
    per[n].append(e - s)
tot = sum(sum(v) for v in per.values())
print("sum of kernel durations %.3f ms = %.2f x the span (> 1 means time-sliced or overlapped execution)" % (tot / 1e6, tot / span))
print("%-26s %8s %12s" % ("kernel", "launches", "avg us live"))
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("%-26s %8d %12.1f" % (n, len(v), sum(v) / len(v) / 1e3))
