"""Per-kernel timing of one RL view-iteration at an arbitrary shape: python tools/shape_probe.py d0 d1 d2 [psf]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libmultiviewnative_amd import native
lib = native.lib()
shape = tuple(int(x) for x in sys.argv[1:4])
kedge = int(sys.argv[4]) if len(sys.argv) > 4 else 15
print("plan", lib.plan_describe(shape), flush=True)
rng = np.random.default_rng(0)
view = rng.uniform(10, 20, shape).astype(np.float32)
k = np.zeros((kedge,) * 3, np.float32); c = kedge // 2
k[c, c, c] = 0.5; k[c - 1, c, c] = 0.25; k[c + 1, c, c] = 0.25
eng = lib.engine(shape, 1)
eng.set_view(0, view, np.full(shape, 0.5, np.float32), k, k)
eng.set_psi(np.full(shape, 15.0, np.float32))
eng.iterate(1, 0.006, 1e-4)
ms = eng.time_iterate(3, 0.006, 1e-4) / 3
B = 4 * shape[0] * shape[1] * 2 * (shape[2] // 2 + 1)
eng.profile(True); eng.iterate(2, 0.006, 1e-4); eng.sync()
prof = {n: round(t / c_, 4) for n, (t, c_) in eng.profile_read().items() if c_}
eng.close()
print(shape, "view-iter %.3f ms  -> %.0f GB/s on the 25B model" % (ms, 25 * B / ms / 1e6), prof, flush=True)
for d in (0, 1):
    t, per = lib.fft3_profile(shape, d, 3)
    print("   fft3 dir", d, "%.3f ms" % t, {kk: round(v, 4) for kk, v in per.items()}, flush=True)
