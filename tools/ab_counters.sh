#!/bin/bash
# A/B of two builds (tools/bin/libA.so, libB.so) with the numbers that decide a kernel experiment: path frames/s, the
# per-kernel times of the single-lane pass, and the SQ counters of the 64-frame dispatches (two --pmc passes each).
#   tools/ab_counters.sh [scene-args]
L=chessboard-vision_amd/lib/libcbv_hip.so
cp $L /tmp/lib_orig.so
export TMPDIR=/tmp
for v in A B; do
  cp tools/bin/lib$v.so $L
  out=$PWD/gpurun_out/abc_$v
  rm -rf $out; mkdir -p $out
  python bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-4k --no-region-leg --no-class-api "$@" > $out/bench.json 2>/dev/null
  python - "$v" $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d["kernels"]
print(sys.argv[1], "fps", d["value"], "bilateral alone us/frame", k["k_bilateral"]["ms_per_frame"] * 1e3, "live ms/launch", d["roofline"]["avg_launch_ms"],
      "noise:", d["noise_scene"]["value"] if d.get("noise_scene") else None, d["noise_scene"]["bilateral_ms_per_frame_alone"] * 1e3 if d.get("noise_scene") else None, flush=True)
PY
  A="--frames 64 --steps 1 --warmup 0 --lanes 1 --cpu-frames 0 --no-profile-pass"
  (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $out/p1 -o r -- python3 $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2> $out/p1.err)
  (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD --output-format csv -d $out/p2 -o r -- python3 $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2> $out/p2.err)
  find $out/p1 -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $out/p1.csv
  find $out/p2 -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $out/p2.csv
  rm -rf $out/p1 $out/p2
  python tools/pmc_summary.py $out/p1.csv $out/p2.csv > $out/sq_counters.txt
  grep -A8 "^derived" $out/sq_counters.txt | head -9
done
cp /tmp/lib_orig.so $L
