#!/bin/bash
# throughput of the default workload over (lanes, chunk); prints one line per point
for lanes in 1 2 3 4; do for chunk in 16 32 64; do
  python bench.py --lanes $lanes --chunk $chunk --cpu-frames 0 --no-profile-pass --steps 4 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lanes $lanes chunk $chunk', d['value'], d['roofline']['avg_launch_ms'])"
done; done
