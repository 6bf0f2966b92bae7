"""Every tap-count template of the fused middle pass (PSF depths 1 .. 31 along dim0) on random plane counts, through the
blocking ABI call and the resident engine's slab range, against the CPU oracle.   python tools/fuzz_mid_fused.py [seed]
(on the GPU box; MVN_MID_FUSED=2 forces the line layout for shallow volumes too)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MVN_MID_FUSED"] = "2"
os.environ["MVN_PAD_MODE"] = "none"
import numpy as np  # noqa: E402
from libmultiviewnative_amd import native  # noqa: E402
from libmultiviewnative_amd.abi import WorkspaceHolder  # noqa: E402
from oracle import binding as orc  # noqa: E402
from ref_fixtures import realistic_views  # noqa: E402

lib = native.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
bad = 0
for k0 in range(1, 32):
    d0 = int(rng.integers(max(k0, 4) + 1, 41))
    shape = (d0, 512, 512)
    ks = (k0, int(rng.choice([1, 3, 5])), int(rng.choice([1, 3, 4])))
    lam = float(rng.choice([0.0, 0.006]))
    its = int(rng.integers(1, 3))
    _, views, k1, k2, w, psi0 = realistic_views(shape, 2, ks, seed=int(rng.integers(1 << 30)))
    h = WorkspaceHolder(views, k1, k2, w, lam, 1e-4, its)
    c0 = lib.l.mvn_mid_fused_launch_count()
    got = lib.gpu_deconvolve(psi0, h)
    fused = lib.l.mvn_mid_fused_launch_count() - c0
    ref = orc.cpu_deconvolve(psi0, h, -1)
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    worst = max(worst, err)
    # (volumes of fewer than K + 4 planes keep the transform along dim0: no fused launches there)
    ok = err <= 1e-5 and np.isfinite(got).all() and fused in (0, its * 2 * 2)
    bad += 0 if ok else 1
    print("k0 %2d shape %-15s k=%-11s lam=%-5g its=%d  fused launches %2d  rl rel err %.2e%s"
          % (k0, shape, ks, lam, its, fused, err, "" if ok else "   <-- CHECK"), flush=True)
print("worst RL rel err %.2e, %d to check" % (worst, bad))
sys.exit(1 if bad else 0)
