"""Host->device copy rate WHILE the RL loop runs on the same GPU (what block k+1's upload sees under
block k's iterations): pageable / registered source, one or two copy threads, 64 MB .. 2 GB pieces.
    python tools/h2d_under_load.py"""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libmultiviewnative_amd import native
lib = native.lib()
hip = C.CDLL("libamdhip64.so.7")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
n = 1 << 31  # 2 GiB per buffer
host = [np.ones(n // 4, np.float32) for _ in range(2)]
dev, st = [], []
for i in range(2):
    d, s = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d), n) == 0 and hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    dev.append(d); st.append(s)

def copy(i, piece=n):
    assert hip.hipSetDevice(0) == 0
    base = host[i].ctypes.data
    for off in range(0, n, piece):
        assert hip.hipMemcpyAsync(C.c_void_p(dev[i].value + off), C.c_void_p(base + off), min(piece, n - off), 1, st[i]) == 0
    assert hip.hipStreamSynchronize(st[i]) == 0

shape = (512, 512, 512)
eng = lib.engine(shape, 2)
rng = np.random.default_rng(0)
k = np.zeros((15, 15, 15), np.float32); k[7, 7, 7] = 1
for v in range(2):
    eng.set_view(v, rng.uniform(10, 20, shape).astype(np.float32), np.full(shape, 0.5, np.float32), k, k)
eng.set_psi(np.full(shape, 15.0, np.float32))
eng.iterate(2, 0.006, 1e-4, sync=True)
stop = False
def load():
    while not stop:
        eng.iterate(20, 0.006, 1e-4, sync=True)

def measure(tag):
    copy(0); copy(1)
    t = time.perf_counter(); copy(0); copy(1); t1 = time.perf_counter() - t
    th = [threading.Thread(target=copy, args=(i,)) for i in range(2)]
    t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; t2 = time.perf_counter() - t
    t = time.perf_counter(); copy(0, 64 << 20); copy(1, 64 << 20); t3 = time.perf_counter() - t
    print("%-34s one thread %.1f GB/s, two threads %.1f GB/s, one thread in 64 MB pieces %.1f GB/s" % (
        tag, 2 * n / t1 / 1e9, 2 * n / t2 / 1e9, 2 * n / t3 / 1e9), flush=True)

measure("pageable, idle GPU")
th = threading.Thread(target=load); th.start(); time.sleep(0.2)
measure("pageable, RL loop running")
stop = True; th.join()
for i in range(2):
    assert hip.hipHostRegister(host[i].ctypes.data_as(C.c_void_p), n, 0) == 0
measure("registered, idle GPU")
stop = False
th = threading.Thread(target=load); th.start(); time.sleep(0.2)
measure("registered, RL loop running")
stop = True; th.join()
t = time.perf_counter(); eng.iterate(20, 0.006, 1e-4, sync=True); dt = time.perf_counter() - t
print("RL loop alone: %.2f ms per 2-view iteration" % (dt / 20 * 1e3))
eng.close()
