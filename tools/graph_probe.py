"""Launch-bound regime: ms per RL view-iteration on small volumes, captured-graph replay (default) vs
direct launches (MVN_GRAPH=0 in the environment).  python tools/graph_probe.py [edge ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libmultiviewnative_amd import native
lib = native.lib()
edges = [int(x) for x in sys.argv[1:]] or [32, 64, 128, 256]
for n in edges:
    shape = (n, n, n)
    V = 2
    rng = np.random.default_rng(0)
    k = np.zeros((9, 9, 9), np.float32); k[4, 4, 4] = 0.5; k[3, 4, 4] = 0.25; k[5, 4, 4] = 0.25
    eng = lib.engine(shape, V)
    for v in range(V):
        eng.set_view(v, rng.uniform(10, 20, shape).astype(np.float32), np.full(shape, 0.5, np.float32), k, k)
    eng.set_psi(np.full(shape, 15.0, np.float32))
    eng.iterate(4, 0.006, 1e-4)       # warm-up: plans, graph capture + instantiation
    its = 40
    ms = eng.time_iterate(its, 0.006, 1e-4) / (its * V)
    print("%d^3: %.4f ms per view-iteration (MVN_GRAPH=%s)" % (n, ms, os.environ.get("MVN_GRAPH", "1")), flush=True)
    eng.close()
