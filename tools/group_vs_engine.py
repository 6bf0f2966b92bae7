"""Where does a group of slab engines (mvn_group_*) differ from one engine?  Diagnostic for the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
os.environ["MVN_NYQ_PACKED"] = "1"
from libmultiviewnative_amd import native
from libmultiviewnative_amd.abi import WorkspaceHolder
from ref_fixtures import realistic_views
lib = native.lib()
cases = [((64, 64, 512), (7, 5, 5)), ((64, 512, 512), (7, 5, 5)), ((512, 64, 64), (31, 5, 5)), ((128, 512, 512), (31, 31, 31)),
         ((512, 512, 512), (31, 31, 31))]
for shape, ks in cases:
    V = 1
    _, views, k1, k2, w, psi0 = realistic_views(shape, V, ks, seed=3)
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 1)
    for devs in ([0], [0, 0]):
        for its in (1, 3):
            e = lib.engine(shape, V)
            e.set_view(0, views[0], w[0], k1[0], k2[0]); e.set_psi(psi0); e.iterate(its, 0.006, 1e-4, sync=True)
            one = e.get_psi(); e.close()
            g = lib.group(devs, shape, ks[0] // 2, V)
            g.load(psi0, h); g.iterate(its, 0.006, 1e-4); got = g.get_psi(); g.close()
            d = np.abs(got.astype(np.float64) - one)
            planes = np.nonzero(d.reshape(shape[0], -1).max(1) > 0)[0]
            print(shape, ks, devs, its, "bit_equal", bool(np.array_equal(got, one)), "max_rel %.2e" % (d.max() / np.abs(one).max()),
                  "planes differing", len(planes), planes[:6], planes[-6:] if len(planes) else "", flush=True)
