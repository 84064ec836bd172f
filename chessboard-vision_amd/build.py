"""Builds libcbv_hip.so (gfx950 only) in-tree with hipcc.

    python -m chessboard_vision_amd.build        # or __graft_entry__.build()

Each translation unit is compiled to an object in parallel, then linked.
-ffp-contract=off is part of the contract: the float stages are specified as
one rounding per operation, never fused.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcbv_hip.so")
SOURCES = ["cbv_api.cpp", "cbv_tables.cpp", "k_enhance.hip", "k_bilateral.hip", "k_warp.hip", "k_analysis.hip",
           "k_squares.hip", "k_hough.hip", "k_canny.hip", "chess_rules.cpp", "contours.cpp"]
HEADERS = ["cbv_internal.h", "cbv_device.h", os.path.join("..", "..", "include", "cbv.h")]
FLAGS = [*(["-DHG_TIMING"] if os.environ.get("HG_TIMING") else []), *(["-DBL_TWO_COPIES"] if os.environ.get("BL_TWO_COPIES") else []), *(["-DSH_TIMING"] if os.environ.get("SH_TIMING") else []), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-x", "hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
