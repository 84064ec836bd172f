"""profiles/r02/bilateral_counters.json from the two SQ counter passes of tools/pmc_sq.sh: vector lane-ops per pixel
(SQ_INSTS_VALU counts wave instructions; x 64 lanes / pixels of the dispatch), LDS conflict share, VALU-active share.
bench.py reads `valu_lane_ops_per_px` from it for its roofline.valu object.

    python tools/pmc_bilateral_json.py gpurun_out/pmc_sq/p1.csv gpurun_out/pmc_sq/p2.csv 64 1920 1080 > profiles/r02/bilateral_counters.json
"""
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import load

p1, p2, frames, w, h = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
a, b = load(p1)["k_bilateral"], load(p2)["k_bilateral"]
px = frames * w * h
out = {
    "source": "rocprofv3 --pmc SQ_* (tools/pmc_sq.sh: bench.py --frames %d --steps 1 --warmup 0 --lanes 1 --cpu-frames 0 --no-profile-pass), "
              "the %d-frame k_bilateral dispatch" % (frames, frames),
    "frames_per_dispatch": frames, "width": w, "height": h,
    "SQ_INSTS_VALU": a["SQ_INSTS_VALU"], "valu_lane_ops_per_px": round(a["SQ_INSTS_VALU"] * 64 / px, 2),
    "SQ_INSTS_LDS": b["SQ_INSTS_LDS"], "lds_wave_instructions_per_px": round(b["SQ_INSTS_LDS"] * 64 / px, 3),
    "SQ_LDS_IDX_ACTIVE": b["SQ_LDS_IDX_ACTIVE"], "SQ_LDS_BANK_CONFLICT": b["SQ_LDS_BANK_CONFLICT"],
    "lds_conflict_share_of_lds_cycles": round(b["SQ_LDS_BANK_CONFLICT"] / b["SQ_LDS_IDX_ACTIVE"], 4),
    "SQ_ACTIVE_INST_VALU_quadcycles": a["SQ_ACTIVE_INST_VALU"], "SQ_WAVE_CYCLES_quadcycles": a["SQ_WAVE_CYCLES"],
    "SQ_WAIT_INST_LDS_quadcycles": a["SQ_WAIT_INST_LDS"], "dispatch_us": round(a["_dur"] / 1e3, 1),
    "valu_busy_share_per_simd": round(a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (a["_dur"] * 2.4), 4),
    "lds_busy_share_per_cu": round(b["SQ_LDS_IDX_ACTIVE"] / 256 / (b["_dur"] * 2.4), 4),
    "note": "busy shares assume 2.4 GHz; SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)",
}
json.dump(out, sys.stdout, indent=1)
