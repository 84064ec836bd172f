"""Which NUMA node the GPU hangs off, which CPUs this process may run on, and what one class-API warp call costs."""
import ctypes, glob, os, sys, time, statistics
_libc = ctypes.CDLL(None)
def getcpu():
    return _libc.sched_getcpu()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
aff = sorted(os.sched_getaffinity(0))
print("affinity: %d cpus, %s..%s" % (len(aff), aff[:4], aff[-4:]))
for n in sorted(glob.glob("/sys/devices/system/node/node*")):
    try:
        print(os.path.basename(n), "cpulist", open(n + "/cpulist").read().strip())
    except Exception as e:
        print(n, e)
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        print(d, "numa_node", open(d + "/numa_node").read().strip(), "vendor", open(d + "/vendor").read().strip(), os.path.basename(os.path.realpath(d)))
    except Exception as e:
        pass
print("ROCR_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"), "HIP_VISIBLE_DEVICES", os.environ.get("HIP_VISIBLE_DEVICES"), "CUDA_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES"))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), "cpuset", open("/sys/fs/cgroup/cpuset.cpus.effective").read().strip())
except Exception as e:
    print("cgroup", e)
import numpy as np
from chessboard_vision_amd import _native as N, synth as S
from chessboard_vision_amd.board_detection import get_perspective_transform
def med(fn, n=40, warm=5):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return round(statistics.median(ts) * 1e6, 1), round(min(ts) * 1e6, 1)
w, h = 1920, 1080
c = N.context()
print(c.name, "running on cpu", getcpu())
f = np.random.default_rng(0).integers(0, 255, (h, w, 3), dtype=np.uint8)
dst = np.float32([[0, 0], [620, 0], [0, 620], [620, 620]])
M = np.ascontiguousarray(get_perspective_transform(S.scaled_corners(w, h), dst))
board = np.empty((620, 620, 3), np.uint8)
call = lambda: c.check(c.lib.cbv_warp_perspective(c.h, N.ptr(f), w, h, f.strides[0], N.ptr(M), 620, 620, 0, N.ptr(board), board.strides[0]))
print("warp call (us, median/min):", med(call))
if len(sys.argv) > 1:
    # pin to the cpus of one node, re-allocate the frame there, and time again
    for n in sorted(glob.glob("/sys/devices/system/node/node*")):
        lst = open(n + "/cpulist").read().strip()
        cpus = set()
        for part in lst.split(","):
            if "-" in part:
                a, b = part.split("-"); cpus |= set(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        cpus &= set(aff)
        if not cpus:
            continue
        os.sched_setaffinity(0, cpus)
        f2 = f.copy(); b2 = np.empty((620, 620, 3), np.uint8)
        call2 = lambda: c.check(c.lib.cbv_warp_perspective(c.h, N.ptr(f2), w, h, f2.strides[0], N.ptr(M), 620, 620, 0, N.ptr(b2), b2.strides[0]))
        print(os.path.basename(n), "pinned + reallocated:", med(call2), "on cpu", getcpu())
