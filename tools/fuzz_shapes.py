"""Differential fuzz: random shapes / kernel extents / view counts through the HIP library against
the CPU oracle (RL result) and numpy (forward transform).  python tools/fuzz_shapes.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from libmultiviewnative_amd import native
from libmultiviewnative_amd.abi import WorkspaceHolder
from oracle import binding as orc
from ref_fixtures import realistic_views

lib = native.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
pool = [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 16, 17, 20, 24, 25, 27, 31, 32, 33, 36, 40, 48, 49, 62, 64,
        66, 72, 80, 96, 100, 127, 128, 130, 160, 192, 256]
if os.environ.get("FUZZ_FIXED"):  # mostly lengths with compile-time kernels, mixed over the three axes
    pool = [64, 96, 128, 160, 192, 256, 288, 320, 384, 512, 576, 640, 48, 30, 18]
pad_mode = os.environ.get("FUZZ_PAD", "none")  # "zero": the default policy, oracle on hand-padded stacks
worst = 0.0
for i in range(n):
    while True:
        shape = tuple(int(rng.choice(pool)) for _ in range(3))
        if np.prod(shape) <= int(os.environ.get("FUZZ_MAX_VOXELS", "3000000")):
            break
    V = int(rng.integers(1, 4))
    # FUZZ_DEEP=1: PSF depths up to 35 along dim0 (the direct dim0 leg takes <= 33, deeper ones the fused FFT pass)
    k0s = [1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 21, 30, 31, 33, 35] if os.environ.get("FUZZ_DEEP") else [1, 2, 3, 4, 5, 7]
    ks = tuple(int(min(s, rng.choice(k0s if d == 0 else [1, 2, 3, 4, 5, 7]))) for d, s in enumerate(shape))
    lam = float(rng.choice([0.0, 0.006, 0.1]))
    its = int(rng.integers(1, 5))
    _, views, k1, k2, w, psi0 = realistic_views(shape, V, ks, seed=int(rng.integers(1 << 30)))
    h = WorkspaceHolder(views, k1, k2, w, lam, 1e-4, its)
    got = lib.gpu_deconvolve(psi0, h)
    ref = orc.cpu_deconvolve(psi0, h, 4)
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    x = rng.standard_normal(shape).astype(np.float32)
    spec = lib.rfft3(x)
    sref = np.fft.rfftn(x.astype(np.float64))
    ferr = float(np.abs(spec - sref).max() / np.abs(sref).max())
    worst = max(worst, err)
    flag = "" if (err <= 1e-4 and ferr <= 1e-5 and np.isfinite(got).all()) else "   <-- CHECK"
    print("%3d shape %-16s V=%d k=%-10s lam=%-5g its=%d  rl rel err %.2e  fft rel err %.2e%s"
          % (i, shape, V, ks, lam, its, err, ferr, flag), flush=True)
print("worst RL rel err", worst)
