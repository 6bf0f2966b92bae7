"""What ONE slab of an N-device run costs in compute, measured on one GPU: a group of one slab (its own neighbour both
ways) on the planes an N-way cut of 512 x 512 x 512 gives a device, 6 views, 31^3 PSFs - own planes computed, 2 x 15
halo planes exchanged with itself by device copies.  Link time is not in these numbers.

    python tools/slab_cost.py [planes ...]        (default 512 256 128 64)
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from libmultiviewnative_amd import native
from libmultiviewnative_amd.abi import WorkspaceHolder

lib = native.lib()
V, psf = 6, 31
for nz in [int(a) for a in sys.argv[1:]] or [512, 256, 128, 64]:
    shape = (nz, 512, 512)
    views, k1s, k2s = [], [], []
    for v in range(V):
        view, k1, k2 = bench.make_view(shape, v, psf)
        views.append(view); k1s.append(k1); k2s.append(k2)
    w = np.full(shape, 1.0 / V, np.float32)
    h = WorkspaceHolder(views, k1s, k2s, [w] * V, bench.LAMBDA, bench.MIN_VALUE, 10)
    g = lib.group([0], shape, psf // 2, V)
    g.load(np.full(shape, np.float32(bench.start_value()), np.float32), h)
    g.iterate(2, bench.LAMBDA, bench.MIN_VALUE)
    ms = g.iterate(10, bench.LAMBDA, bench.MIN_VALUE) / 10
    g.close()
    e = lib.engine(shape, V)
    for v in range(V):
        e.set_view(v, views[v], w, k1s[v], k2s[v])
    e.set_psi(np.full(shape, np.float32(bench.start_value()), np.float32))
    e.iterate(2, bench.LAMBDA, bench.MIN_VALUE, sync=True)
    plain = e.time_iterate(10, bench.LAMBDA, bench.MIN_VALUE) / 10
    e.close()
    print(json.dumps({"own_planes": nz, "halo_planes": 2 * (psf // 2), "slab_ms_per_iteration": round(ms, 3),
                      "plain_engine_on_the_same_planes_ms": round(plain, 3),
                      "as_one_of_n_devices": 512 // nz}), flush=True)
