"""Does the cost of a pageable copy depend on how many OTHER host buffers were copied before (a cache of pinned
ranges that fills up, a locked-memory limit)?  Copies K distinct 6.2 MB buffers once each, then times copies of a
fresh buffer and of the first one."""
import ctypes as C, os, sys, time, statistics, resource
import numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
dev, st = C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(dev), 64 << 20) == 0
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
print("RLIMIT_MEMLOCK", resource.getrlimit(resource.RLIMIT_MEMLOCK))
N = 6220800
def cp(buf, n=N, d2h=False, off=0):
    a = time.perf_counter()
    if d2h: hip.hipMemcpyAsync(buf.ctypes.data + off, dev, n, 2, st)
    else: hip.hipMemcpyAsync(dev, buf.ctypes.data + off, n, 1, st)
    hip.hipStreamSynchronize(st)
    return (time.perf_counter() - a) * 1e6
def med(fn, n=9):
    return statistics.median(fn() for _ in range(n))
bufs = []
first = np.full(N, 7, np.uint8)
print("first buffer, first copy %.0f us, then %.0f us" % (cp(first), med(lambda: cp(first))))
for K in (1, 2, 4, 8, 16, 32, 64, 96):
    while len(bufs) < K:
        b = np.full(N, len(bufs) & 255, np.uint8)
        cp(b)                       # copied once: pinned, maybe cached
        cp(b, d2h=True)
        bufs.append(b)
    fresh = np.full(N, 9, np.uint8)
    t_fresh1 = cp(fresh)
    t_fresh = med(lambda: cp(fresh))
    t_rows = med(lambda: cp(fresh, 5324800, off=604800))
    t_board = med(lambda: cp(fresh, 1153200, d2h=True))
    t_first = med(lambda: cp(first))
    t_old = med(lambda: cp(bufs[0]))
    print("after %3d other buffers: fresh buffer 1st copy %6.0f, repeated %6.0f | its rows %6.0f | board D2H %6.0f | very first buffer %6.0f | buffer 0 %6.0f us" % (K, t_fresh1, t_fresh, t_rows, t_board, t_first, t_old))
