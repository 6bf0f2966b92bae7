"""Times the reference ABI call itself (host buffers in, host buffer out): what Fiji sees."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libmultiviewnative_amd import native
from libmultiviewnative_amd.abi import WorkspaceHolder
lib = native.lib()
shape, V, its = (512, 512, 512), 6, 10
rng = np.random.default_rng(0)
views = [rng.random(shape, dtype=np.float32) * 50 + 10 for _ in range(V)]
w = [np.full(shape, 1.0 / V, np.float32) for _ in range(V)]
ax = np.arange(31) - 15.0
g = np.exp(-0.5 * (ax[:, None, None] / 3) ** 2 - 0.5 * (ax[None, :, None] / 2) ** 2 - 0.5 * (ax[None, None, :] / 2) ** 2)
psf = (g / g.sum()).astype(np.float32)
h = WorkspaceHolder(views, [psf] * V, [np.ascontiguousarray(psf[::-1, ::-1, ::-1])] * V, w, 0.006, 1e-4, its)
psi0 = np.full(shape, 35.0, np.float32)
for rep in range(3):
    t = time.perf_counter()
    out = lib.gpu_deconvolve(psi0, h, 0)
    dt = time.perf_counter() - t
    print("inplace_gpu_deconvolve 512^3 x %d views x %d iterations: %.3f s end to end (call %d)" % (V, its, dt, rep), flush=True)
print("finite:", bool(np.isfinite(out).all()))
