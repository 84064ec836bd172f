#!/bin/bash
# A/B of two whole trees on the GPU box, alternating bench runs in one session: A = tools/bin/old_tree (an earlier commit,
# `git archive <commit> ... | tar -x -C tools/bin/old_tree` + its built library), B = this tree.
#   tools/ab_trees.sh [rounds] [extra bench args]
R=${1:-3}; shift
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then B=tools/bin/old_tree/bench.py; else B=bench.py; fi
    python $B --steps 30 --warmup 3 --cpu-frames 0 --no-4k --no-profile-pass "$@" > /tmp/ab.json 2>/dev/null
    python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["occupancy_check"], flush=True)
PY
  done
done
