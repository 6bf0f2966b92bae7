"""End-to-end oracle RL loop: closed forms from the reference's synthetic bench data, the
numpy float64 restatement, and the loop invariants of tests/test_gpu_deconvolve_impl.cu."""
import numpy as np
import pytest

from libmultiviewnative_amd.abi import WorkspaceHolder
from oracle import binding as orc
from ref_fixtures import realistic_views, synthetic_views
from numpy_restatement import deconvolve as np_deconvolve


def f_reg(x, lam=0.006):
    return (np.sqrt(1 + 2 * lam * x) - 1) / lam


@pytest.mark.parametrize("n_views,want", [(1, 29.40588), (6, 37.72946), (8, 43.75619)])
def test_synthetic_closed_form(n_views, want):
    # SURVEY.md 8c golden (3): bench/synthetic_data.hpp:59-96 data, lambda=.006: after the last
    # view psi == f((16+4v)(v+2)/(v+1)) in every voxel
    shape = (16, 16, 16)
    views, k1, k2, w = synthetic_views(shape, n_views, 3, 5)
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-3, 2)
    psi = orc.cpu_deconvolve(np.full(shape, 3.0, np.float32), h, 1)
    v = n_views - 1
    assert abs(f_reg((16 + 4 * v) * (v + 2) / (v + 1)) - want) < 1e-4
    assert np.abs(psi - want).max() < 2e-4 * want


def test_zero_iterations_returns_input():
    # tests/test_gpu_deconvolve_impl.cu:333-376
    shape = (8, 8, 8)
    views, k1, k2, w = synthetic_views(shape, 2, 3, 3)
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 0)
    psi0 = np.random.default_rng(0).uniform(1, 2, shape).astype(np.float32)
    assert np.array_equal(orc.cpu_deconvolve(psi0, h, 1), psi0)


def test_zero_psi_recovers_through_nan_guard():
    # SURVEY.md 8c hazard: start_psi == 0 -> view/0 = Inf -> FFT(Inf) = NaN -> !(NaN>0) -> min
    shape = (8, 8, 8)
    views, k1, k2, w = synthetic_views(shape, 1, 3, 3)
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-3, 1)
    psi = orc.cpu_deconvolve(np.zeros(shape, np.float32), h, 1)
    assert np.all(psi == np.float32(1e-3))


@pytest.mark.parametrize("lam", [0.0, 0.006])
@pytest.mark.parametrize("shape,kshape,nv", [((16, 20, 18), (5, 5, 5), 3), ((13, 17, 19), (3, 5, 3), 2)])
def test_realistic_vs_numpy(shape, kshape, nv, lam):
    _, views, k1, k2, w, psi0 = realistic_views(shape, nv, kshape)
    h = WorkspaceHolder(views, k1, k2, w, lam, 1e-4, 4)
    got = orc.cpu_deconvolve(psi0, h, 2)
    ref = np_deconvolve(psi0, views, k1, k2, w, lam, 1e-4, 4)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 5e-5
    assert np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)) < 5e-6


def test_n_iterations_equals_n_times_one():
    # SURVEY.md section 5 checkpoint/resume row; tests/test_cpu_deconvolve.cpp:66-92
    shape = (12, 10, 14)
    _, views, k1, k2, w, psi0 = realistic_views(shape, 2, (3, 3, 3))
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 3)
    a = orc.cpu_deconvolve(psi0, h, 1)
    h.with_iterations(1)
    b = psi0
    for _ in range(3):
        b = orc.cpu_deconvolve(b, h, 1)
    assert np.array_equal(a, b)


def test_serial_equals_parallel():
    # tests/test_cpu_deconvolve.cpp: serial == parallel exactly
    shape = (12, 10, 14)
    _, views, k1, k2, w, psi0 = realistic_views(shape, 2, (3, 3, 3))
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 2)
    assert np.array_equal(orc.cpu_deconvolve(psi0, h, 1), orc.cpu_deconvolve(psi0, h, 4))


def test_simultaneous_mode():
    shape = (12, 10, 14)
    _, views, k1, k2, w, psi0 = realistic_views(shape, 3, (3, 3, 3))
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 3)
    got = orc.cpu_deconvolve_simultaneous(psi0, h, 2)
    ref = np_deconvolve(psi0, views, k1, k2, w, 0.006, 1e-4, 3, simultaneous=True)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 5e-5
    # one view: simultaneous == sequential
    h1 = WorkspaceHolder(views[:1], k1[:1], k2[:1], w[:1], 0.006, 1e-4, 3)
    assert np.array_equal(orc.cpu_deconvolve_simultaneous(psi0, h1, 1), orc.cpu_deconvolve(psi0, h1, 1))
    # partial sums over view shards add up to the full step
    full = orc.simultaneous_step(psi0, h, 0, 3)
    parts = orc.simultaneous_step(psi0, h, 0, 2) + orc.simultaneous_step(psi0, h, 2, 3)
    assert np.abs(full - parts).max() <= 1e-5 * np.abs(full).max() + 1e-7
