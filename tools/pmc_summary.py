"""Summarise rocprofv3 --pmc counter CSVs (tools/pmc_sq.sh): one table per CSV, the longest dispatch of every k_*
kernel (the chunked launch), plus derived ratios when the counters are there.

    python tools/pmc_summary.py gpurun_out/pmc_sq/p1.csv gpurun_out/pmc_sq/p2.csv
"""
import collections
import csv
import sys


def load(path):
    rows = collections.defaultdict(dict)  # (kernel, dispatch) -> {counter: value, "_dur": ns}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if not k.startswith("k_"):
            continue
        d = rows[(k, int(r["Dispatch_Id"]))]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["_dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d["_grid"] = int(r["Grid_Size"])
    best = {}
    for (k, _), d in rows.items():
        # the batched launch = the longest dispatch of the kernel (grids do not tell: a single-frame bilateral launch uses
        # 512 workgroups of 512 lanes, the 64-frame one 256 persistent workgroups of 768)
        if k not in best or d["_dur"] > best[k]["_dur"]:
            best[k] = d
    return best


def main():
    merged = collections.defaultdict(dict)
    for path in sys.argv[1:]:
        best = load(path)
        names = sorted({c for d in best.values() for c in d if c[0] != "_"})
        print(path)
        print("%-24s" % "kernel" + "".join("%17s" % n.replace("SQ_", "")[:16] for n in names) + "%10s" % "dur_us")
        for k, d in sorted(best.items(), key=lambda kv: -kv[1]["_dur"]):
            print("%-24s" % k[:24] + "".join("%17.4g" % d.get(c, float("nan")) for c in names) + "%10.1f" % (d["_dur"] / 1e3))
            merged[k].update(d)
        print()
    print("derived (largest dispatch of each kernel)")
    print("%-24s%14s%14s%14s%14s" % ("kernel", "lds_conflict", "valu_active", "wait_any", "valu/px"))
    for k, d in sorted(merged.items(), key=lambda kv: -kv[1]["_dur"]):
        conf = d.get("SQ_LDS_BANK_CONFLICT", float("nan")) / max(d.get("SQ_LDS_IDX_ACTIVE", float("nan")), 1)
        va = d.get("SQ_ACTIVE_INST_VALU", float("nan")) / max(d.get("SQ_WAVE_CYCLES", float("nan")), 1)
        wa = d.get("SQ_WAIT_ANY", float("nan")) / max(d.get("SQ_WAVE_CYCLES", float("nan")), 1)
        print("%-24s%14.3f%14.3f%14.3f%14.4g" % (k[:24], conf, va, wa, d.get("SQ_INSTS_VALU", float("nan"))))


if __name__ == "__main__":
    main()
