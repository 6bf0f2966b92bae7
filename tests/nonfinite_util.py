"""Non-finite inputs through the RL loop: shared by the emulation test and the -m gpu test."""
import numpy as np

from libmultiviewnative_amd.abi import WorkspaceHolder
from oracle import binding as orc
from ref_fixtures import realistic_views


def nonfinite_cases(binding, shape, kshape, pos, its_list=(1, 2), nviews=2, seed=8):
    """One Inf voxel - in psi, then in a view - through the sequential sweep and the simultaneous step, against
    the oracle (shared by the emulation and the -m gpu test).  The reference's FFT convolution turns the voxel
    into a volume of NaN and the update clamps every voxel to minValue (inc/cpu_convolve.h:256-268,
    inc/cpu_kernels.h:40-47,76-83).  Inf in psi: the voxel itself stays NaN, so every later convolution is
    flooded as well and the whole run is exact arithmetic -> bit-equal.  Inf in a view: that view's update is the
    exact minValue blend of a psi the other views have updated with ordinary (rounded) arithmetic -> 1e-5."""
    for where in ("psi", "view"):
        _, views, k1, k2, w, psi0 = realistic_views(shape, nviews, kshape, seed=seed)
        psi = psi0.copy()
        if where == "psi":
            psi[pos] = np.inf
        else:
            views[nviews - 1][pos] = np.inf
        for its in its_list:
            h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, its)
            got = binding.gpu_deconvolve(psi, h, pad_mode="none")
            ref = orc.cpu_deconvolve(psi, h, 8)
            e = binding.engine(shape, nviews)
            try:
                for v in range(nviews):
                    e.set_view(v, views[v], w[v], k1[v], k2[v])
                e.set_psi(psi)
                for _ in range(its):
                    e.compute_delta(0.006, 1e-4)
                    e.apply_delta()
                got_sim = e.get_psi()
            finally:
                e.close()
            ref_sim = orc.cpu_deconvolve_simultaneous(psi, h, 8)
            for tag, g, r in (("sequential", got, ref), ("simultaneous", got_sim, ref_sim)):
                what = (shape, kshape, where, its, tag)
                assert np.array_equal(np.isnan(g), np.isnan(r)), what
                if where == "psi":
                    assert np.array_equal(g, r, equal_nan=True), what
                    assert np.isnan(r).sum() == 1  # the voxel itself; everything else was clamped and blended
                else:
                    assert np.abs(g - r).max() <= 1e-5 * np.abs(r).max(), what
            # the flood really happened: the last view's update was the minValue blend everywhere
            if where == "view" and its == 1:
                _, vclean, _, _, _, _ = realistic_views(shape, nviews, kshape, seed=seed)
                hc = WorkspaceHolder(vclean, k1, k2, w, 0.006, 1e-4, 1)
                assert np.abs(orc.cpu_deconvolve(psi, hc, 8) - ref).max() > 1e-2 * np.abs(ref).max()
