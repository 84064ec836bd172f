"""hipMemcpyAsync + synchronise between pageable host memory and the device for a sweep of sizes, both directions:
microseconds and GB/s per size.  Shows where the runtime switches between its staging path and pinning the user's pages
(a box-dependent knee would explain class-API calls that are 3x slower in some processes)."""
import ctypes as C, os, sys, time, statistics
import numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
dev, st = C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(dev), 64 << 20) == 0
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
if len(sys.argv) > 1:  # big allocations first, like a pipeline would make
    junk = [C.c_void_p() for _ in range(6)]
    for j in junk:
        hip.hipMalloc(C.byref(j), 400 << 20)
host = np.random.default_rng(0).integers(0, 255, 64 << 20, dtype=np.uint8)
def med(fn, n=15):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return statistics.median(ts) * 1e6
print("%10s %12s %12s" % ("bytes", "H2D us (GB/s)", "D2H us (GB/s)"))
for kb in (64, 256, 512, 1024, 1126, 2048, 3072, 4096, 5200, 6075, 8192, 16384, 32768):
    n = kb << 10
    for off in (0, 604800):
        h = host.ctypes.data + off
        a = med(lambda: (hip.hipMemcpyAsync(dev, h, n, 1, st), hip.hipStreamSynchronize(st)))
        b = med(lambda: (hip.hipMemcpyAsync(h, dev, n, 2, st), hip.hipStreamSynchronize(st)))
        print("%10d +%6d %7.1f (%5.1f) %7.1f (%5.1f)" % (n, off, a, n / a / 1e3, b, n / b / 1e3))
