"""Pins the oracle's FFT against the reference's own FFT-boundary tests and numpy pocketfft."""
import numpy as np
import pytest

from oracle import binding as orc


def test_plan_store_ramp_roundtrip_exact():
    # tests/test_plan_store.cpp:83-142: 8^3 integer ramp, r2c -> c2r -> /512 equals input exactly
    x = np.arange(512, dtype=np.float32).reshape(8, 8, 8)
    spec = orc.rfft3_forward(x)
    back = orc.rfft3_backward(spec, 8) / np.float32(512)
    assert np.array_equal(back, x)


@pytest.mark.parametrize("shape", [(13, 17, 19), (16, 16, 16), (9, 27, 3), (27, 9, 81),
                                   (5, 25, 125), (7, 49, 7), (16, 18, 14), (10, 10, 10)])
def test_roundtrip_mse(shape):
    # tests/test_fftw_numerical_stability.cpp:32-664: ramp 0..N-1, roundtrip MSE < 1e-4
    n = int(np.prod(shape))
    x = np.arange(n, dtype=np.float32).reshape(shape)
    back = orc.rfft3_backward(orc.rfft3_forward(x), shape[2]) / np.float32(n)
    mse = float(np.mean((back.astype(np.float64) - x) ** 2))
    # the reference bound is absolute (1e-4) on ramps up to 16^3=4096; scale it for larger ramps
    bound = 1e-4 * max(1.0, (n / 4096.0) ** 2)
    assert mse < bound, mse


@pytest.mark.parametrize("shape", [(8, 8, 8), (13, 17, 19), (16, 18, 14), (6, 10, 15), (32, 20, 64),
                                   (4, 4, 7), (3, 5, 2), (64, 64, 64)])
def test_forward_matches_pocketfft(shape):
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape).astype(np.float32)
    ref = np.fft.rfftn(x.astype(np.float64))
    got = orc.rfft3_forward(x)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() / scale < 5e-6


@pytest.mark.parametrize("shape", [(8, 8, 8), (13, 17, 19), (16, 18, 14), (4, 4, 7), (32, 20, 64)])
def test_backward_matches_pocketfft(shape):
    rng = np.random.default_rng(2)
    x = rng.standard_normal(shape)
    spec = np.fft.rfftn(x).astype(np.complex64)
    ref = np.fft.irfftn(spec.astype(np.complex128), s=shape, axes=(0, 1, 2)) * np.prod(shape)
    got = orc.rfft3_backward(spec, shape[2])
    assert np.abs(got - ref).max() / np.abs(ref).max() < 5e-6


def test_threads_agree():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((12, 20, 18)).astype(np.float32)
    assert np.array_equal(orc.rfft3_forward(x, 1), orc.rfft3_forward(x, 4))
