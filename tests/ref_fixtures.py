"""The reference's self-contained synthetic fixture, regenerated from its definition
(``tests/test_fixtures.hpp:21-305``, ``default_3D_fixture = convolutionFixture3D<3,8>``) and
the synthetic bench data (``bench/synthetic_data.hpp:59-96``).  Data only -- no reference code.
"""
import numpy as np


class Fixture3D:
    def __init__(self, kdim=3, idim=8):
        h = kdim // 2
        self.image = np.arange(idim ** 3, dtype=np.float32).reshape(idim, idim, idim)  # :197-204
        self.one = np.zeros((idim,) * 3, np.float32)
        self.one[idim // 2, idim // 2, idim // 2] = 1
        p = idim + 2 * h
        self.padded_image = np.zeros((p,) * 3, np.float32)  # :206-210
        self.padded_image[h:h + idim, h:h + idim, h:h + idim] = self.image
        self.padded_one = np.zeros((p,) * 3, np.float32)
        self.padded_one[p // 2, p // 2, p // 2] = 1
        self.interior = (slice(h, h + idim),) * 3
        z = lambda: np.zeros((kdim,) * 3, np.float32)
        self.trivial = z()
        self.identity = z()
        self.identity[h, h, h] = 1
        self.horizont, self.vertical, self.depth = z(), z(), z()
        for i in range(kdim):  # :123-151
            self.horizont[h, h, i] = i + 1
            self.vertical[h, i, h] = i + 1
            self.depth[i, h, h] = i + 1
        self.all1 = np.ones((kdim,) * 3, np.float32)
        # asymmetric kernels (kdim+1, kdim, kdim-1)  :153-183
        shp = (kdim + 1, kdim, kdim - 1)
        self.asymm_cross = np.zeros(shp, np.float32)
        self.asymm_one = np.zeros(shp, np.float32)
        self.asymm_identity = np.zeros(shp, np.float32)
        cz, cy, cx = shp[0] // 2, shp[1] // 2, shp[2] // 2
        self.asymm_identity[cz, cy, cx] = 1
        for zz in range(shp[0]):
            for yy in range(shp[1]):
                for xx in range(shp[2]):
                    if zz == cz and yy == cy:
                        self.asymm_cross[zz, yy, xx] = xx + 1
                        self.asymm_one[zz, yy, xx] = 1
                    if xx == cx and yy == cy:
                        self.asymm_cross[zz, yy, xx] = zz + 101
                        self.asymm_one[zz, yy, xx] = 1
                    if xx == cx and zz == cz:
                        self.asymm_cross[zz, yy, xx] = yy + 11
                        self.asymm_one[zz, yy, xx] = 1
        self.asymm_offsets = tuple(s // 2 for s in shp)
        ap = tuple(idim + 2 * o for o in self.asymm_offsets)
        self.asymm_padded_image = np.zeros(ap, np.float32)
        self.asymm_interior = tuple(slice(o, o + idim) for o in self.asymm_offsets)
        self.asymm_padded_image[self.asymm_interior] = self.image
        self.asymm_padded_one = np.zeros(ap, np.float32)
        self.asymm_padded_one[ap[0] // 2, ap[1] // 2, ap[2] // 2] = 1


# interior sums of the zero-padded 'same' spatial convolution of the 8^3 ramp with the 3^3
# kernels (SURVEY.md 8c golden (1); the reference computes them in its fixture constructor,
# tests/test_fixtures.hpp:258-283, and compares sums within 1e-5 %).
GOLDEN_SUMS = {"identity": 130816.0, "horizont": 719040.0, "vertical": 715904.0,
               "depth": 690816.0, "all1": 2720564.0}


def spatial_convolve_same(image, kernel):
    """float64 zero-padded 'same' true convolution (tests/test_algorithms.hpp:10-58)."""
    from scipy.signal import convolve
    full = convolve(image.astype(np.float64), kernel.astype(np.float64), mode="full")
    off = [k // 2 for k in kernel.shape]
    # reference indexes image[z - k/2 + kz] with flipped kernel -> centre (k-1) - k/2
    off = [(k - 1) - k // 2 for k in kernel.shape]
    sl = tuple(slice(o, o + s) for o, s in zip(off, image.shape))
    return full[sl]


def synthetic_views(shape, n_views, k1=21, k2=25):
    """bench/synthetic_data.hpp:59-96 with the kernel edge as a parameter."""
    views, w, ka, kb = [], [], [], []
    for i in range(n_views):
        views.append(np.full(shape, 16.0 + 4.0 * i, np.float32))
        w.append(np.ones(shape, np.float32))
        a = np.zeros((k1,) * 3, np.float32)
        a[k1 // 2, k1 // 2, k1 // 2] = i + 1
        b = np.zeros((k2,) * 3, np.float32)
        b[k2 // 2, k2 // 2, k2 // 2] = i + 2
        ka.append(a)
        kb.append(b)
    return views, ka, kb, w


def gaussian_psf(shape, sigma):
    ax = [np.arange(s, dtype=np.float64) - s // 2 for s in shape]
    g = np.exp(-0.5 * (ax[0][:, None, None] / sigma[0]) ** 2
               - 0.5 * (ax[1][None, :, None] / sigma[1]) ** 2
               - 0.5 * (ax[2][None, None, :] / sigma[2]) ** 2)
    return (g / g.sum()).astype(np.float32)


def realistic_views(shape, n_views, kshape=(7, 7, 7), seed=42, n_blobs=12):
    """SURVEY.md 8d generator (ii): blobs + background, views = truth (*) PSF_v (cyclic)."""
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    truth = np.full(shape, 10.0, np.float64)
    for _ in range(n_blobs):
        c = [rng.uniform(0.1 * s, 0.9 * s) for s in shape]
        sg = rng.uniform(1.0, 3.0)
        amp = rng.uniform(50, 500)
        truth += amp * np.exp(-((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) / (2 * sg * sg))
    views, k1s, k2s, ws = [], [], [], []
    for v in range(n_views):
        sig = [1.0, 1.0, 1.0]
        sig[v % 3] = 2.0
        psf = gaussian_psf(kshape, sig)
        big = np.zeros(shape, np.float64)
        for idx in np.ndindex(*kshape):
            t = tuple((i - k // 2) % s for i, k, s in zip(idx, kshape, shape))
            big[t] = psf[idx]
        view = np.fft.irfftn(np.fft.rfftn(truth) * np.fft.rfftn(big), s=shape, axes=(0, 1, 2))
        views.append(view.astype(np.float32))
        k1s.append(psf)
        k2s.append(np.ascontiguousarray(psf[::-1, ::-1, ::-1]))
        ws.append(np.full(shape, 1.0 / n_views, np.float32))
    psi0 = np.full(shape, np.float32(views[0].mean()), np.float32)
    return truth.astype(np.float32), views, k1s, k2s, ws, psi0


def structured_views(shape, n_views, kshape, seed=7, terms=4):
    """Cheap non-trivial stacks for FULL-SIZE parity cases (512^3 and up, where realistic_views'
    float64 meshgrids are too heavy): a few separable Gaussian blobs on a background of 10 plus a
    little noise, so that every voxel differs and every spectral bin is populated.  One shared
    weights array (1 / n_views).  Returns views, kernels1, kernels2, weights, psi0."""
    rng = np.random.default_rng(seed)
    ax = [np.arange(s, dtype=np.float32) for s in shape]
    views, k1s, k2s = [], [], []
    for v in range(n_views):
        acc = rng.random(shape, dtype=np.float32)
        acc *= np.float32(2.0)
        acc += np.float32(10.0)
        for _ in range(terms):
            f = [np.exp(-0.5 * ((a - np.float32(rng.uniform(0.1 * s, 0.9 * s))) /
                                np.float32(rng.uniform(2.0, max(2.5, 0.12 * s)))) ** 2).astype(np.float32)
                 for a, s in zip(ax, shape)]
            plane = (np.float32(rng.uniform(50, 400)) * f[1])[:, None] * f[2][None, :]
            acc += f[0][:, None, None] * plane[None, :, :]
        views.append(acc)
        sig = [1.5, 1.5, 1.5]
        sig[v % 3] = 3.0
        psf = gaussian_psf(tuple(min(k, s) for k, s in zip(kshape, shape)), sig)
        k1s.append(psf)
        k2s.append(np.ascontiguousarray(psf[::-1, ::-1, ::-1]))
    w = np.full(shape, 1.0 / n_views, np.float32)
    psi0 = np.full(shape, np.float32(views[0].mean()), np.float32)
    return views, k1s, k2s, [w] * n_views, psi0


def expected_good_extent(binding, n, last_axis):
    """The padded extent the ABI call picks for one axis (mvn_abi.cpp good_extent): the cheapest
    2^a 3^b 5^c 7^d length in [n, 1.3 n + 8], fixed-kernel lengths priced 1, others 1.45."""
    def smooth(c):
        for p in (2, 3, 5, 7):
            while c % p == 0:
                c //= p
        return c == 1

    def fixed(c):
        info = binding.plan_describe((8, 8, c) if last_axis else (c, 8, 8))
        return bool(info["fx_rows"] if last_axis else info["fx_ax0"])

    cand = [c for c in range(n, n + n * 3 // 10 + 9) if smooth(c)]
    return min(cand, key=lambda c: (c * (1.0 if fixed(c) else 1.45), c))
