"""Oracle convolution vs the reference's convolution fixtures
(tests/test_cpu_symm_convolve.cpp, test_cpu_asymm_convolve.cpp, test_padd_utils.cpp)."""
import numpy as np
import pytest

from oracle import binding as orc
from ref_fixtures import Fixture3D, GOLDEN_SUMS, spatial_convolve_same

FX = Fixture3D()


def test_fixture_sanity():
    # tests/test_multi_array_fixture.cpp:11-17
    assert FX.image[4, 4, 4] == 292
    assert FX.padded_image.shape == (10, 10, 10)


def test_wrapped_insert_centre_to_origin():
    # inc/padd_utils.h:11-40; tests/test_padd_utils.cpp wrapped insert
    k = np.arange(27, dtype=np.float32).reshape(3, 3, 3)
    t = orc.wrapped_insert(k, (8, 9, 10))
    assert t[0, 0, 0] == k[1, 1, 1]
    assert t[7, 8, 9] == k[0, 0, 0]
    assert t[1, 1, 1] == k[2, 2, 2]
    assert t[7, 0, 1] == k[0, 1, 2]
    assert np.count_nonzero(t) == 26  # k[0,0,0]==0
    # asymmetric / even extents: centre index k/2 lands on the origin
    k2 = np.arange(1, 25, dtype=np.float32).reshape(4, 3, 2)
    t2 = orc.wrapped_insert(k2, (8, 8, 8))
    assert t2[0, 0, 0] == k2[2, 1, 1]
    assert t2[6, 7, 7] == k2[0, 0, 0]
    assert t2.sum() == k2.sum()


@pytest.mark.parametrize("name", ["identity", "horizont", "vertical", "depth", "all1"])
@pytest.mark.parametrize("nthreads", [1, 4])
def test_symm_convolve_sums(name, nthreads):
    # tests/test_cpu_symm_convolve.cpp:15-230: inplace_cpu_convolution on padded_image_ (10^3),
    # interior 8^3 sum == spatial-convolution sum within 1e-5 %
    kernel = getattr(FX, name)
    out = orc.cpu_convolution(FX.padded_image, kernel, nthreads)
    got = float(out[FX.interior].astype(np.float64).sum())
    want = GOLDEN_SUMS[name]
    assert abs(got - want) / want * 100 < 1e-5 * 100  # reference: BOOST_REQUIRE_CLOSE(.., 1e-5 %)... 
    # voxel-wise against the zero-padded 'same' spatial convolution
    ref = spatial_convolve_same(FX.image, kernel)
    assert np.abs(out[FX.interior] - ref).max() < 2e-3 * max(1.0, np.abs(ref).max() / 1e3)


def test_golden_sums_from_definition():
    for name, want in GOLDEN_SUMS.items():
        ref = spatial_convolve_same(FX.image, getattr(FX, name))
        assert abs(ref.sum() - want) < 1e-6 * want


def test_oracle_spatial_convolve_matches_scipy():
    ref = spatial_convolve_same(FX.image, FX.asymm_cross)
    got = orc.spatial_convolve(FX.image, FX.asymm_cross)
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("name", ["asymm_cross", "asymm_one", "asymm_identity"])
def test_asymm_delta_reproduces_kernel(name):
    # tests/test_cpu_asymm_convolve.cpp:15-54: padded delta (10^3, at [5,5,5]) (*) asymmetric
    # (4x3x2) kernel; interior sum == kernel sum (.001 %), and the kernel re-appears in the
    # interior at [shape/2 - k/2, shape/2 - k/2 + k) (kernel voxel k/2 lands on the delta).
    kernel = getattr(FX, name)
    out = orc.cpu_convolution(FX.padded_one, kernel, 1)
    one = out[FX.interior]
    assert abs(float(one.sum()) - float(kernel.sum())) / float(kernel.sum()) * 100 < 1e-3
    pos = tuple(slice(s // 2 - k // 2, s // 2 - k // 2 + k) for s, k in zip(one.shape, kernel.shape))
    seg = one[pos]
    assert seg.shape == kernel.shape
    assert np.array_equal(np.floor(seg + 0.5), kernel)
    assert np.abs(seg - kernel).max() < 1e-4


def test_trivial_kernel_gives_zero():
    out = orc.cpu_convolution(FX.padded_image, FX.trivial, 1)
    assert np.abs(out).max() == 0


@pytest.mark.parametrize("shape,kshape", [((16, 18, 14), (3, 3, 3)), ((13, 17, 19), (5, 3, 7)),
                                          ((20, 12, 9), (4, 3, 2))])
def test_cyclic_convolution_vs_numpy(shape, kshape):
    from numpy_restatement import cyclic_convolve
    rng = np.random.default_rng(5)
    im = rng.uniform(0, 10, shape).astype(np.float32)
    k = rng.uniform(0, 1, kshape).astype(np.float32)
    got = orc.cpu_convolution(im, k, 2)
    ref = cyclic_convolve(im, k)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6
