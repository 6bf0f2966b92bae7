"""Times the reference ABI call itself (host buffers in, host buffer out): what Fiji sees for a
sequence of blocks -- blocking inplace_gpu_deconvolve calls against the same blocks pipelined through
mvn_deconvolve_submit / mvn_deconvolve_wait (uploads of block k+1 under the iterations of block k).
    python tools/abi_end_to_end.py [edge=512] [blocks=4] [pad_mode=none]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libmultiviewnative_amd import native
from libmultiviewnative_amd.abi import WorkspaceHolder
lib = native.lib()
edge = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else "none"
shape, V, its = (edge, edge, edge), 6, 10
rng = np.random.default_rng(0)
views = [rng.random(shape, dtype=np.float32) * 50 + 10 for _ in range(V)]
w = [np.full(shape, 1.0 / V, np.float32) for _ in range(V)]
ax = np.arange(31) - 15.0
g = np.exp(-0.5 * (ax[:, None, None] / 3) ** 2 - 0.5 * (ax[None, :, None] / 2) ** 2 - 0.5 * (ax[None, None, :] / 2) ** 2)
psf = (g / g.sum()).astype(np.float32)
h = WorkspaceHolder(views, [psf] * V, [np.ascontiguousarray(psf[::-1, ::-1, ::-1])] * V, w, 0.006, 1e-4, its)
lib.set_pad_mode(mode)
psis = [np.full(shape, 35.0 + b, np.float32) for b in range(nblocks)]
# warm both engine lanes (allocation, plans, PSF spectra)
warm = [p.copy() for p in psis[:2]]  # kept alive until waited for: the worker writes into them
for t in [lib.deconvolve_submit(p, h) for p in warm]:
    lib.deconvolve_wait(t)
blocking = [p.copy() for p in psis]
t0 = time.perf_counter()
for p in blocking:
    lib.gpu_deconvolve_inplace(p, h, 0)
t_block = time.perf_counter() - t0
piped = [p.copy() for p in psis]
t0 = time.perf_counter()
tickets = [lib.deconvolve_submit(p, h) for p in piped]
for t in tickets:
    lib.deconvolve_wait(t)
t_pipe = time.perf_counter() - t0
same = all(np.array_equal(a, b) for a, b in zip(blocking, piped))
print("%d blocks of %d^3 x %d views x %d iterations, pad_mode %s: blocking %.3f s (%.2f blocks/s), submit/wait %.3f s "
      "(%.2f blocks/s) = x%.2f; results identical: %s" % (nblocks, edge, V, its, mode, t_block, nblocks / t_block, t_pipe,
                                                        nblocks / t_pipe, t_block / t_pipe, same), flush=True)
lib.set_pad_mode(None)
