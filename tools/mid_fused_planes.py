"""Where the fused middle pass starts to pay: ms per view update of a resident engine on (d0, 512, 512) volumes, the
three-pass middle (MVN_MID_FUSED=0) against the fused pass forced on (MVN_MID_FUSED=2), by planes and PSF depth.
    python tools/mid_fused_planes.py            (on the GPU box; prints a table)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from libmultiviewnative_amd import native  # noqa: E402
from ref_fixtures import realistic_views  # noqa: E402


def ms_per_view_update(lib, shape, k0, mode, its=20):
    os.environ["MVN_MID_FUSED"] = mode
    _, views, k1, k2, w, psi0 = realistic_views(shape, 2, (k0, 5, 5), seed=1)
    e = lib.engine(shape, 2)
    try:
        for v in range(2):
            e.set_view(v, views[v], w[v], k1[v], k2[v])
        e.set_psi(psi0)
        e.iterate(3, 0.006, 1e-4)
        e.sync()
        c0 = lib.l.mvn_mid_fused_launch_count()
        t = time.perf_counter()
        e.iterate(its, 0.006, 1e-4)
        e.sync()
        dt = time.perf_counter() - t
        fused = lib.l.mvn_mid_fused_launch_count() - c0
    finally:
        e.close()
    return dt / (its * 2) * 1e3, fused


def main():
    lib = native.lib()
    print("| planes | PSF planes | three passes, ms per view update | fused pass | ratio |")
    print("|---|---|---|---|---|")
    for k0 in (31, 15, 5):
        for d0 in (32, 48, 64, 96, 128, 256):
            if d0 < k0 + 4:
                continue
            a, fa = ms_per_view_update(lib, (d0, 512, 512), k0, "0")
            b, fb = ms_per_view_update(lib, (d0, 512, 512), k0, "2")
            assert fa == 0 and fb > 0, (fa, fb)
            print("| %d | %d | %.4f | %.4f | %.2f |" % (d0, k0, a, b, b / a), flush=True)


if __name__ == "__main__":
    main()
