#!/bin/bash
# path frames/s for a few (chunk, lanes, splits) of the 512-frame step (two passes, so box drift shows)
for pass in 1 2; do
for cfg in "64 2 4" "64 2 2" "64 3 2" "32 2 2" "128 2 2" "96 2 2" "64 2 8"; do
  set -- $cfg
  python bench.py --steps 30 --warmup 3 --cpu-frames 0 --no-4k --no-profile-pass --no-noise-leg --no-class-api --no-region-leg --chunk $1 --lanes $2 --splits $3 > /tmp/sw.json 2>/dev/null || { echo "chunk=$1 lanes=$2 splits=$3 FAILED"; continue; }
  python - "$cfg" <<'PY'
import json, sys
d = json.loads(open("/tmp/sw.json").read().strip().splitlines()[-1])
print("chunk lanes splits = %-10s %9.1f frames/s  %7.3f ms/step  ok=%s" % (sys.argv[1], d["value"], d["ms_per_step"], d["occupancy_check"]), flush=True)
PY
done
done
