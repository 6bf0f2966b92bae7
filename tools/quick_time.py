"""Quick on-GPU timing probe: 3-D FFT and RL sweeps with per-kernel event timing."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from libmultiviewnative_amd import native

lib = native.lib()
print("backend", lib.backend_name())
for shape in [(256, 256, 256), (512, 512, 512)]:
    B = 4 * shape[0] * shape[1] * 2 * (shape[2] // 2 + 1)
    for d in (0, 1):
        ms, per = lib.fft3_profile(shape, d, 10)
        print("fft3", shape, "dir", d, "%.3f ms" % ms, "%.0f GB/s (6B model)" % (6 * B / ms / 1e6),
              {k: round(v, 4) for k, v in per.items()})
    V = 2
    eng = lib.engine(shape, V)
    rng = np.random.default_rng(0)
    k = np.zeros((15, 15, 15), np.float32); k[7, 7, 7] = 0.5; k[6, 7, 7] = 0.25; k[8, 7, 7] = 0.25
    for v in range(V):
        eng.set_view(v, rng.uniform(10, 20, shape).astype(np.float32), np.full(shape, 0.5, np.float32), k, k)
    eng.set_psi(np.full(shape, 15.0, np.float32))
    eng.iterate(1, 0.006, 1e-4)
    ms = eng.time_iterate(5, 0.006, 1e-4) / 5
    print("RL sweep", shape, "V=%d" % V, "%.3f ms/iter" % ms, "%.0f GB/s (25B model)" % (25 * B * V / ms / 1e6))
    eng.profile(True)
    eng.iterate(3, 0.006, 1e-4)
    eng.sync()
    for name, (tms, n) in eng.profile_read().items():
        if n:
            print("   %-12s n=%3d avg %.3f ms" % (name, n, tms / n))
    eng.profile(False)
    eng.close()
