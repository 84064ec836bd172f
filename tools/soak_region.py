"""One-off soak of region-limited enhancement (tests/test_gpu_region.py::run_random_region_case over many seeds)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_region import run_random_region_case  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fails, t0 = 0, time.time()
for i in range(n):
    try:
        run_random_region_case(100000 + i)
    except AssertionError as e:
        fails += 1
        print("seed", 100000 + i, "FAILED", str(e)[:200], flush=True)
    if i % 50 == 49:
        print("it %d / %d, %d failures, %.0f s" % (i + 1, n, fails, time.time() - t0), flush=True)
print("soak_region: %d cases, %d failures" % (n, fails))
