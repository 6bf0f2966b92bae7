"""Pageable host->device copy rate: one thread vs two threads on two streams (what a second staging
thread in the ABI call could gain).  python tools/h2d_probe.py [GiB per buffer]"""
import ctypes as C, sys, threading, time
import numpy as np
hip = C.CDLL("libamdhip64.so.7")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gib * (1 << 30))
host = [np.ones(n // 4, np.float32) for _ in range(2)]
dev, st = [], []
for i in range(2):
    d, s = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d), n) == 0 and hip.hipStreamCreate(C.byref(s)) == 0
    dev.append(d); st.append(s)
def copy(i):
    assert hip.hipSetDevice(0) == 0
    assert hip.hipMemcpyAsync(dev[i], host[i].ctypes.data_as(C.c_void_p), n, 1, st[i]) == 0
    assert hip.hipStreamSynchronize(st[i]) == 0
copy(0); copy(1)
t = time.perf_counter(); copy(0); copy(1); t1 = time.perf_counter() - t
th = [threading.Thread(target=copy, args=(i,)) for i in range(2)]
t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; t2 = time.perf_counter() - t
print("pageable, one thread : %.1f GB/s" % (2 * n / t1 / 1e9))
print("pageable, two threads: %.1f GB/s" % (2 * n / t2 / 1e9))
t = time.perf_counter()
for i in range(2):
    assert hip.hipHostRegister(host[i].ctypes.data_as(C.c_void_p), n, 0) == 0
tr = time.perf_counter() - t
t = time.perf_counter(); copy(0); copy(1); t3 = time.perf_counter() - t
print("hipHostRegister      : %.3f s for %.1f GiB (%.1f GB/s)" % (tr, 2 * gib, 2 * n / tr / 1e9))
print("registered, one thread: %.1f GB/s" % (2 * n / t3 / 1e9))
