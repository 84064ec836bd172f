#!/bin/bash
# like tools/ab.sh, but prints the per-kernel stand-alone times of the single-lane profile pass as well
R=${1:-2}; shift
L=chessboard-vision_amd/lib/libcbv_hip.so
cp $L /tmp/lib_orig.so
for i in $(seq $R); do
  for v in A B; do
    cp tools/bin/lib$v.so $L
    python bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-4k "$@" > /tmp/ab.json 2>/dev/null
    python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
k = d["kernels"]
print(sys.argv[1], d["value"], " ".join("%s=%.3f" % (n[2:8], 1e3 * k[n]["ms_per_frame"]) for n in k), flush=True)
PY
  done
done
cp /tmp/lib_orig.so $L
