"""profiles/<round>/path_valu.json from the two SQ counter passes of tools/pmc_sq.sh: for every kernel of the hot path,
vector instructions and VALU-busy / LDS-busy time per frame, and their sums over the path.  bench.py sets the sums
beside the live time per frame (`path_valu`): the path as a whole is bound by vector issue, not by any one kernel.

    python tools/pmc_path_json.py gpurun_out/pmc_sq/p1.csv gpurun_out/pmc_sq/p2.csv 64 > profiles/r02/path_valu.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import load

PATH_KERNELS = ["k_reset_aux", "k_color_lab_hist", "k_clahe_lut", "k_clahe_apply", "k_bilateral", "k_sharpen_box", "k_norm_lut",
                "k_warp", "k_squares_pre5_stats", "k_hough", "k_scan", "k_noise", "k_pack_results"]
GHZ, SIMDS, CUS = 2.4, 1024, 256

p1, p2, frames = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
rows, tot = {}, {"valu_wave_instructions": 0.0, "valu_busy_us": 0.0, "lds_busy_us": 0.0, "alone_us": 0.0}
for k in PATH_KERNELS:
    if k not in p1 or k not in p2:
        continue
    a, b = p1[k], p2[k]
    r = {
        "valu_wave_instructions": a["SQ_INSTS_VALU"] / frames,
        # SQ_ACTIVE_INST_VALU: quad-cycles summed over waves -> cycles per SIMD -> us at 2.4 GHz
        "valu_busy_us": a["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / (GHZ * 1e3) / frames,
        "lds_busy_us": b["SQ_LDS_IDX_ACTIVE"] / CUS / (GHZ * 1e3) / frames,
        "alone_us": a["_dur"] / 1e3 / frames,
    }
    for f in tot:
        tot[f] += r[f]
    rows[k] = {f: round(v, 3) for f, v in r.items()}
out = {
    "source": "rocprofv3 --pmc SQ_* (tools/pmc_sq.sh: bench.py --frames %d --steps 1 --warmup 0 --lanes 1 --cpu-frames 0 "
              "--no-profile-pass), the %d-frame dispatch of every kernel, divided by %d" % (frames, frames, frames),
    "per_frame": rows,
    "sum_per_frame": {f: round(v, 3) for f, v in tot.items()},
    "note": "busy times assume %.1f GHz, %d SIMDs, %d CUs; SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md); "
            "valu_busy of a kernel is the time its vector instructions occupy a SIMD's issue port, averaged over the chip" % (GHZ, SIMDS, CUS),
}
json.dump(out, sys.stdout, indent=1)
