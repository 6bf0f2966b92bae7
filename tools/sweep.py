"""Tile/threads sweep on the GPU: re-plans with different env knobs and prints per-kernel times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libmultiviewnative_amd import native
lib = native.lib()
shape = tuple(int(x) for x in os.environ.get("AB_SHAPE", "512 512 512").split())
rng = np.random.default_rng(0)
view = rng.uniform(10, 20, shape).astype(np.float32)
wts = np.full(shape, 0.5, np.float32)
k = np.zeros((15, 15, 15), np.float32); k[7, 7, 7] = 0.5; k[6, 7, 7] = 0.25; k[8, 7, 7] = 0.25
psi0 = np.full(shape, 15.0, np.float32)
configs = [dict(), dict(MVN_T_FUSED="16"), dict(MVN_T_AXIS="8"), dict(MVN_T_ROWS="8"),
           dict(MVN_THREADS="256"), dict(MVN_THREADS="256", MVN_T_ROWS="8", MVN_T_AXIS="8")]
if len(sys.argv) > 1:
    configs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]]
lam = float(os.environ.get("SWEEP_LAMBDA", "0.006"))
for cfg in configs:
    for kk in ("MVN_T_ROWS", "MVN_T_AXIS", "MVN_T_FUSED", "MVN_THREADS", "MVN_NO_FIXED"):
        os.environ.pop(kk, None)
    os.environ.update(cfg)
    lib.check(lib.l.mvn_plan_store_clear())
    eng = lib.engine(shape, 1)
    eng.set_view(0, view, wts, k, k)
    eng.set_psi(psi0)
    eng.iterate(1, lam, 1e-4)
    ms = eng.time_iterate(5, lam, 1e-4) / 5
    eng.profile(True)
    eng.iterate(3, lam, 1e-4)
    eng.sync()
    prof = {n: round(t / c, 4) for n, (t, c) in eng.profile_read().items() if c}
    eng.profile(False)
    eng.close()
    print(cfg, "view-iter %.3f ms" % ms, prof, flush=True)
    for d in (() if os.environ.get("AB_NO_FFT") else (0, 1)):
        ms, per = lib.fft3_profile(shape, d, 10)
        print("   fft3 dir", d, "%.3f ms" % ms, {k: round(v, 4) for k, v in per.items()}, flush=True)
