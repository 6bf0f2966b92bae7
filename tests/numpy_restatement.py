"""Independent float64 numpy restatement of the reference CPU RL path, used to cross-check the C
oracle.  Follows src/multiviewnative.cpp:101-240,
inc/cpu_convolve.h:217-291, inc/cpu_kernels.h:19-90, inc/padd_utils.h:11-40.
"""
import numpy as np


def wrapped_insert(kernel, shape):
    out = np.zeros(shape, np.float64)
    for idx in np.ndindex(*kernel.shape):
        t = []
        for i, k, s in zip(idx, kernel.shape, shape):
            j = i - k // 2
            t.append(j + s if j < 0 else j)
        out[tuple(t)] = kernel[idx]
    return out


def cyclic_convolve(image, kernel):
    shape = image.shape
    spec = np.fft.rfftn(wrapped_insert(kernel, shape))
    return np.fft.irfftn(np.fft.rfftn(image.astype(np.float64)) * spec, s=shape, axes=(0, 1, 2))


def update(psi, integral, weight, lam, min_value):
    last = psi
    value = last * integral
    pos = value > 0
    with np.errstate(invalid="ignore"):
        if lam > 0:
            reg = (1.0 / lam) * (np.sqrt(1.0 + 2.0 * lam * np.where(pos, value, 0.0)) - 1.0)
        else:
            reg = value
    value = np.where(pos, reg, min_value)
    bad = ~np.isfinite(value)
    nxt = np.where(bad, min_value, np.maximum(value, min_value))
    return weight * (nxt - last) + last


def deconvolve(psi, views, k1s, k2s, weights, lam, min_value, iterations, simultaneous=False):
    psi = psi.astype(np.float64).copy()
    shape = psi.shape
    s1 = [np.fft.rfftn(wrapped_insert(k, shape)) for k in k1s]
    s2 = [np.fft.rfftn(wrapped_insert(k, shape)) for k in k2s]
    for _ in range(iterations):
        base = psi.copy()
        acc = np.zeros_like(psi)
        for v in range(len(views)):
            src = base if simultaneous else psi
            blurred = np.fft.irfftn(np.fft.rfftn(src) * s1[v], s=shape, axes=(0, 1, 2))
            with np.errstate(divide="ignore", invalid="ignore"):
                q = views[v].astype(np.float64) * (1.0 / blurred)
            integral = np.fft.irfftn(np.fft.rfftn(q) * s2[v], s=shape, axes=(0, 1, 2))
            new = update(src, integral, weights[v].astype(np.float64), lam, min_value)
            if simultaneous:
                acc += new - src
            else:
                psi = new
        if simultaneous:
            psi = base + acc
    return psi
