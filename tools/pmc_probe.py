"""A handful of launches of every hot kernel at 512^3 for rocprofv3 --pmc / --kernel-trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libmultiviewnative_amd import native
lib = native.lib()
shape = tuple(int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 512, 512)))
rng = np.random.default_rng(0)
eng = lib.engine(shape, 1)
kd = min(31, shape[0] if shape[0] >= 35 else 15, shape[0])  # 31 planes (the headline's depth) where the volume allows
k = np.zeros((kd, 31, 31), np.float32); k[kd // 2, 15, 15] = 0.5; k[kd // 2 - 1, 15, 15] = 0.25; k[kd // 2 + 1, 15, 15] = 0.25
eng.set_view(0, rng.uniform(10, 20, shape).astype(np.float32), np.full(shape, 0.5, np.float32), k, k)
eng.set_psi(np.full(shape, 15.0, np.float32))
eng.iterate(3, 0.006, 1e-4)
eng.close()
print("probe done")
