"""The CPU oracle against the committed golden vectors (tests/golden/)."""
import numpy as np
import pytest

from golden_util import fixture_a, rl_small
from oracle import binding as orc


@pytest.mark.parametrize("name", ["identity", "horizont", "vertical", "depth", "all1"])
def test_fixture_a_convolutions(name):
    g = fixture_a()
    out = orc.cpu_convolution(g["padded_image"], g["kernel_" + name], 1)[1:9, 1:9, 1:9]
    want = g["expect_" + name]
    assert np.abs(out - want).max() <= 3e-6 * np.abs(want).max()
    assert abs(float(out.astype(np.float64).sum()) - float(g["sum_" + name])) / float(g["sum_" + name]) < 1e-7


@pytest.mark.parametrize("lam", [0.0, 0.006])
def test_rl_small(lam):
    psi0, h, seq, sim = rl_small(lam)
    got = orc.cpu_deconvolve(psi0, h, 1)
    assert np.abs(got - seq).max() <= 5e-5 * np.abs(seq).max()
    got = orc.cpu_deconvolve_simultaneous(psi0, h, 1)
    assert np.abs(got - sim).max() <= 5e-5 * np.abs(sim).max()
