"""Oracle pointwise kernels vs the constants of tests/test_gpu_kernels_impl.cu."""
import numpy as np

from oracle import binding as orc


def test_quotient_one_fifth():
    # tests/test_gpu_kernels_impl.cu:57-164: 1/5 exact
    view = np.ones(1000, np.float32)
    blurred = np.full(1000, 5, np.float32)
    out = orc.compute_quotient(view, blurred)
    assert np.all(out == np.float32(np.float64(1) / np.float64(5)))


def test_quotient_zero_and_nan_flow_on():
    out = orc.compute_quotient(np.array([0, 1, -1, 2], np.float32), np.array([0, 0, 0, np.nan], np.float32))
    assert np.isnan(out[0]) and np.isposinf(out[1]) and np.isneginf(out[2]) and np.isnan(out[3])


def test_final_values_constants():
    # tests/test_gpu_kernels_impl.cu:171-351: psi=5, integral=42, w=.1, min=1e-4
    n = 64
    psi = np.full(n, 5, np.float32)
    integral = np.full(n, 42, np.float32)
    w = np.full(n, 0.1, np.float32)
    out = orc.final_values(psi, integral, w, 1e-4, 0.0)
    want = np.float32(0.1) * (np.float32(210) - np.float32(5)) + np.float32(5)
    assert np.all(out == want) and abs(float(want) - 25.5) < 1e-5
    lam = 0.006
    out = orc.final_values(psi, integral, w, 1e-4, lam)
    linv = np.float32(np.float32(1) / np.float64(lam))
    v = np.float32(np.float64(linv) * (np.sqrt(1.0 + 2.0 * lam * 210.0) - 1.0))
    want = np.float32(0.1) * (v - np.float32(5)) + np.float32(5)
    assert np.all(out == want)


def test_final_values_clamps():
    psi = np.array([1, 1, 1, 1, 2], np.float32)
    integral = np.array([-3, np.nan, np.inf, 0, 1e-9], np.float32)
    w = np.ones(5, np.float32)
    for lam in (0.0, 0.006):
        out = orc.final_values(psi, integral, w, 1e-3, lam)
        m = np.float32(1e-3)
        want = w * (m - psi) + psi  # the blend is float arithmetic: 1*(0.001-1)+1 != 0.001
        assert np.array_equal(out, want), out  # 2e-9 (regularised or not) < minValue -> clamped too


def test_final_values_random_vs_numpy():
    from numpy_restatement import update
    rng = np.random.default_rng(0)
    n = 4096
    psi = rng.uniform(0.5, 10, n).astype(np.float32)
    integral = rng.uniform(-0.1, 1, n).astype(np.float32)  # test_gpu_kernels_impl.cu:353-486
    w = rng.uniform(0, 1, n).astype(np.float32)
    for lam in (0.0, 0.006):
        got = orc.final_values(psi, integral, w, 1e-4, lam)
        ref = update(psi.astype(np.float64), integral.astype(np.float64), w.astype(np.float64), lam, 1e-4)
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-6


def test_double_reciprocal_equals_correctly_rounded_float_division():
    # the product computes float(1.0/double(x)) as the IEEE single-precision quotient 1.0f/x
    # (53 >= 2*24+2 bits: no double-rounding error); pinned here against the oracle's double path
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 32, 2_000_000, dtype=np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x)]
    ones = np.ones_like(x)
    with np.errstate(all="ignore"):
        want = np.float32(1.0) / x
    got = orc.compute_quotient(ones, x)
    assert np.array_equal(got, want, equal_nan=True)
