#!/bin/bash
# A/B of two builds of the library on the GPU box: alternating bench runs in one session (tools/README.md).
#   tools/ab.sh [rounds] [extra bench args]      expects tools/bin/libA.so and tools/bin/libB.so
R=${1:-3}; shift
L=chessboard-vision_amd/lib/libcbv_hip.so
cp $L /tmp/lib_orig.so
for i in $(seq $R); do
  for v in A B; do
    cp tools/bin/lib$v.so $L
    python bench.py --steps 30 --warmup 3 --cpu-frames 0 --no-4k --no-profile-pass "$@" > /tmp/ab.json 2>/dev/null
    python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["occupancy_check"], flush=True)
PY
  done
done
cp /tmp/lib_orig.so $L
