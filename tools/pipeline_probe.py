"""Why does the upload of block k+1 slow down under block k's iterations?  Times 12 x 512 MB pageable
host->device copies (what stage_view does) while (a) nothing runs, (b) a resident 6-view RL loop runs,
(c) another thread sits in blocking inplace_gpu_deconvolve calls."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libmultiviewnative_amd import native
from libmultiviewnative_amd.abi import WorkspaceHolder
lib = native.lib()
hip = C.CDLL("libamdhip64.so.7")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
shape, V = (512, 512, 512), 6
n = 4 * 512 ** 3
rng = np.random.default_rng(0)
views = [rng.random(shape, dtype=np.float32) * 50 + 10 for _ in range(V)]
w = [np.full(shape, 1.0 / V, np.float32) for _ in range(V)]
k = np.zeros((31, 31, 31), np.float32); k[15, 15, 15] = 1
dev = []
for i in range(2):
    d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), n) == 0; dev.append(d)
s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0

def copies(tag):
    hip.hipSetDevice(0)
    t = time.perf_counter()
    for i in range(12):
        src = (views if i % 2 == 0 else w)[(i // 2) % V]
        assert hip.hipMemcpyAsync(dev[i % 2], src.ctypes.data_as(C.c_void_p), n, 1, s) == 0
    assert hip.hipStreamSynchronize(s) == 0
    dt = time.perf_counter() - t
    print("%-52s 12 x 512 MB in %.1f ms = %.1f GB/s" % (tag, dt * 1e3, 12 * n / dt / 1e9), flush=True)

copies("idle (first touch)"); copies("idle")
eng = lib.engine(shape, V)
for v in range(V):
    eng.set_view(v, views[v], w[v], k, k)
eng.set_psi(np.full(shape, 35.0, np.float32))
eng.iterate(2, 0.006, 1e-4, sync=True)
stop = False
def load():
    while not stop:
        eng.iterate(10, 0.006, 1e-4, sync=True)
th = threading.Thread(target=load); th.start(); time.sleep(0.1)
copies("resident 6-view RL loop running"); copies("resident 6-view RL loop running")
stop = True; th.join(); eng.close()
h = WorkspaceHolder(views, [k] * V, [k] * V, w, 0.006, 1e-4, 10)
lib.set_pad_mode("none")
psi = np.full(shape, 35.0, np.float32)
lib.gpu_deconvolve_inplace(psi, h, 0)
stop = False
def calls():
    while not stop:
        lib.gpu_deconvolve_inplace(psi, h, 0)
th = threading.Thread(target=calls); th.start(); time.sleep(0.05)
for _ in range(3):
    copies("blocking ABI calls running in another thread")
stop = True; th.join()
lib.set_pad_mode(None)
