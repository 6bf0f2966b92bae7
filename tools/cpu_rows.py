"""CPU rows of BASELINE.md section 3: the oracle (port of inplace_cpu_deconvolve) on the host
cores for the two small configs -- 64^3 / 1 view / 3^3 PSF / 5 iterations and 256^3 / 1 view /
15^3 PSF / 10 iterations -- with 1 thread and with all threads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import binding as orc
from libmultiviewnative_amd.abi import WorkspaceHolder
import bench

for shape, psf, its in (((64, 64, 64), 3, 5), ((256, 256, 256), 15, 10)):
    view, k1, k2 = bench.make_view(shape, 0, psf)
    w = np.ones(shape, np.float32)
    h = WorkspaceHolder([view], [k1], [k2], [w], bench.LAMBDA, bench.MIN_VALUE, its)
    psi0 = np.full(shape, np.float32(35.0), np.float32)
    for threads in (1, -1):
        orc.cpu_deconvolve(psi0, h, threads)
        t = time.perf_counter()
        orc.cpu_deconvolve(psi0, h, threads)
        dt = time.perf_counter() - t
        setup_s, loop_s = orc.last_timing()
        print("%s %d iterations, %s threads (%s FFT): %.1f ms total, loop %.1f ms = %.2f it/s" % (
            "x".join(map(str, shape)), its, orc.threads(threads), orc.fft_backend(), dt * 1e3, loop_s * 1e3,
            its / loop_s), flush=True)
