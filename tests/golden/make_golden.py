"""Generates tests/golden/*.npz -- small committed input/expected-output vectors.

    python tests/golden/make_golden.py

Two kinds of vectors (data only, no reference code):
 * fixture_a.npz  -- the reference's self-contained synthetic fixture (tests/test_fixtures.hpp:21-305:
   8^3 ramp, its zero-bordered 10^3 copy, the 3^3 and 4x3x2 kernels) together with the expected
   convolution results.  The expectations are computed here INDEPENDENTLY of any FFT, by the direct
   zero-padded spatial convolution the reference's own fixture constructor uses
   (tests/test_algorithms.hpp:10-58 via scipy.signal.convolve in float64).
 * rl_small.npz   -- multi-view RL inputs (seeded synthetic views, PSFs, weights, start value) and the
   result after N iterations computed by the float64 numpy restatement of the reference's CPU loop
   (tests/numpy_restatement.py, independent of both the C oracle and the HIP path), for the
   sequential (reference) and the simultaneous sweep, lambda = 0 and 0.006.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from numpy_restatement import deconvolve  # noqa: E402
from ref_fixtures import Fixture3D, realistic_views, spatial_convolve_same  # noqa: E402


def main():
    fx = Fixture3D()
    out = {"image": fx.image, "padded_image": fx.padded_image, "padded_one": fx.padded_one}
    for name in ("identity", "horizont", "vertical", "depth", "all1", "asymm_cross", "asymm_one",
                 "asymm_identity"):
        k = getattr(fx, name)
        out["kernel_" + name] = k
        if not name.startswith("asymm"):
            # expected interior of conv(padded_image, k): zero-padded 'same' convolution of the ramp
            out["expect_" + name] = spatial_convolve_same(fx.image, k).astype(np.float32)
            out["sum_" + name] = np.float64(spatial_convolve_same(fx.image, k).sum())
    np.savez_compressed(os.path.join(HERE, "fixture_a.npz"), **out)

    shape, nv, kshape, its = (12, 10, 14), 3, (3, 5, 3), 3
    _, views, k1, k2, w, psi0 = realistic_views(shape, nv, kshape, seed=7)
    rl = {"psi0": psi0, "iterations": np.int32(its), "min_value": np.float32(1e-4)}
    for v in range(nv):
        rl["view%d" % v], rl["weights%d" % v] = views[v], w[v]
        rl["kernel1_%d" % v], rl["kernel2_%d" % v] = k1[v], k2[v]
    for lam in (0.0, 0.006):
        tag = "lam%g" % lam
        rl["expect_sequential_" + tag] = deconvolve(psi0, views, k1, k2, w, lam, 1e-4, its).astype(np.float32)
        rl["expect_simultaneous_" + tag] = deconvolve(psi0, views, k1, k2, w, lam, 1e-4, its,
                                                      simultaneous=True).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "rl_small.npz"), **rl)
    for f in ("fixture_a.npz", "rl_small.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
