"""ctypes binding of the CPU parity oracle (TEST INFRASTRUCTURE ONLY).

May be imported from ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- never from the product package ``libmultiviewnative_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmvn_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "mvn_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            build()
        _lib = C.CDLL(_SO)
        _setup(_lib)
    return _lib


def _setup(l):
    from libmultiviewnative_amd.abi import Workspace, c_float_p, c_int_p
    l.oracle_rfft3_forward.argtypes = [c_float_p, C.c_int, C.c_int, C.c_int, C.c_int]
    l.oracle_rfft3_backward.argtypes = [c_float_p, C.c_int, C.c_int, C.c_int, C.c_int]
    l.oracle_padded_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    l.oracle_padded_floats.restype = C.c_size_t
    l.oracle_wrapped_insert.argtypes = [c_float_p, c_int_p, c_float_p, c_int_p]
    l.oracle_compute_quotient.argtypes = [c_float_p, c_float_p, C.c_size_t]
    l.oracle_final_values.argtypes = [c_float_p, c_float_p, c_float_p, C.c_size_t, C.c_float]
    l.oracle_regularized_final_values.argtypes = [c_float_p, c_float_p, c_float_p, C.c_size_t,
                                                  C.c_double, C.c_float]
    l.oracle_legacy_tikhonov_final_values.argtypes = [c_float_p, c_float_p, c_float_p, C.c_size_t,
                                                      C.c_float, C.c_float]
    l.oracle_legacy_tikhonov_final_values.restype = None
    l.oracle_update_delta.argtypes = [c_float_p, c_float_p, c_float_p, c_float_p, C.c_int,
                                      C.c_size_t, C.c_double, C.c_float]
    l.inplace_cpu_convolution.argtypes = [c_float_p, c_int_p, c_float_p, c_int_p, C.c_int]
    l.inplace_cpu_deconvolve.argtypes = [c_float_p, Workspace, C.c_int]
    l.oracle_deconvolve_simultaneous.argtypes = [c_float_p, Workspace, C.c_int]
    l.oracle_deconvolve_simultaneous_step.argtypes = [c_float_p, Workspace, C.c_int, C.c_int,
                                                      c_float_p, C.c_int]
    l.oracle_set_quotient_guard.argtypes = [C.c_int]
    l.oracle_set_quotient_guard.restype = None
    l.oracle_last_timing.argtypes = [C.POINTER(C.c_double)]
    l.oracle_last_timing.restype = None
    l.oracle_fft_backend.argtypes = []
    l.oracle_fft_backend.restype = C.c_char_p
    l.oracle_threads.argtypes = [C.c_int]
    l.oracle_threads.restype = C.c_int
    l.oracle_spatial_convolve.argtypes = [c_float_p, c_int_p, c_float_p, c_int_p, c_float_p]
    for name in ("oracle_rfft3_forward", "oracle_rfft3_backward", "oracle_wrapped_insert",
                 "oracle_compute_quotient", "oracle_final_values",
                 "oracle_regularized_final_values", "oracle_update_delta",
                 "inplace_cpu_convolution", "inplace_cpu_deconvolve",
                 "oracle_deconvolve_simultaneous", "oracle_deconvolve_simultaneous_step",
                 "oracle_spatial_convolve"):
        getattr(l, name).restype = None


def _fp(a):
    from libmultiviewnative_amd.abi import fptr
    return fptr(a)


def _ip(a):
    from libmultiviewnative_amd.abi import iptr
    return iptr(a)


def _dims(a):
    return np.array(a.shape, dtype=np.int32)


# ---- numpy-level helpers -------------------------------------------------------------------

def rfft3_forward(x, nthreads=1):
    """x: real [d0,d1,d2] -> complex64 [d0,d1,d2//2+1], un-normalised (FFTW r2c layout)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    d0, d1, d2 = x.shape
    buf = np.zeros((d0, d1, 2 * (d2 // 2 + 1)), dtype=np.float32)
    buf[:, :, :d2] = x
    lib().oracle_rfft3_forward(_fp(buf), d0, d1, d2, nthreads)
    return buf.view(np.complex64)


def rfft3_backward(spec, d2, nthreads=1):
    """spec: complex64 [d0,d1,d2//2+1] -> real [d0,d1,d2], un-normalised."""
    spec = np.ascontiguousarray(spec, dtype=np.complex64).copy()
    d0, d1, nc = spec.shape
    assert nc == d2 // 2 + 1
    buf = spec.view(np.float32)
    lib().oracle_rfft3_backward(_fp(buf), d0, d1, d2, nthreads)
    return np.ascontiguousarray(buf[:, :, :d2])


def wrapped_insert(kernel, target_shape):
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    target = np.zeros(target_shape, dtype=np.float32)
    kd, td = _dims(kernel), np.array(target_shape, dtype=np.int32)
    lib().oracle_wrapped_insert(_fp(kernel), _ip(kd), _fp(target), _ip(td))
    return target


def cpu_convolution(image, kernel, nthreads=1):
    im = np.ascontiguousarray(image, dtype=np.float32).copy()
    k = np.ascontiguousarray(kernel, dtype=np.float32)
    idims, kdims = _dims(im), _dims(k)
    lib().inplace_cpu_convolution(_fp(im), _ip(idims), _fp(k), _ip(kdims), nthreads)
    return im


def compute_quotient(view, blurred):
    out = np.ascontiguousarray(blurred, dtype=np.float32).copy()
    view = np.ascontiguousarray(view, dtype=np.float32)
    lib().oracle_compute_quotient(_fp(view), _fp(out), out.size)
    return out


def final_values(psi, integral, weight, min_value, lambda_=0.0):
    psi = np.ascontiguousarray(psi, dtype=np.float32).copy()
    integral = np.ascontiguousarray(integral, dtype=np.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    if lambda_ > 0:
        lib().oracle_regularized_final_values(_fp(psi), _fp(integral), _fp(weight), psi.size,
                                              lambda_, min_value)
    else:
        lib().oracle_final_values(_fp(psi), _fp(integral), _fp(weight), psi.size, min_value)
    return psi


def legacy_tikhonov_final_values(image, integral, weight, min_value, lambda_):
    image = np.ascontiguousarray(image, dtype=np.float32).copy()
    integral = np.ascontiguousarray(integral, dtype=np.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    lib().oracle_legacy_tikhonov_final_values(_fp(image), _fp(integral), _fp(weight), image.size,
                                              min_value, lambda_)
    return image


def iterate_fft(image, kernel, min_value=1e-4, lambda_=None, nthreads=1):
    """The legacy one-step entry points restated from src/multiviewnative.cu:395-506 (plain,
    lambda_ None) and :508-600 (tikhonov): psi_0 = view = image, kernel2 = 0.1 in every tap,
    weights = 1, both convolutions cyclic."""
    image = np.ascontiguousarray(image, dtype=np.float32)
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    blurred = cpu_convolution(image, kernel, nthreads)
    quotient = compute_quotient(image, blurred)
    integral = cpu_convolution(quotient, np.full_like(kernel, .1), nthreads)
    ones = np.ones_like(image)
    if lambda_ is None:
        return final_values(image, integral, ones, min_value, 0.0)
    return legacy_tikhonov_final_values(image, integral, ones, min_value, lambda_)


def cpu_deconvolve(psi, holder, nthreads=1):
    """inplace_cpu_deconvolve twin (sequential view sweep); returns the new psi."""
    out = np.ascontiguousarray(psi, dtype=np.float32).copy()
    lib().inplace_cpu_deconvolve(_fp(out), holder.ws, nthreads)
    return out


def set_quotient_guard(on):
    lib().oracle_set_quotient_guard(1 if on else 0)


def last_timing():
    """(psf_setup_seconds, iteration_loop_seconds) of the last cpu_deconvolve call."""
    t = (C.c_double * 2)()
    lib().oracle_last_timing(t)
    return t[0], t[1]


def fft_backend():
    """'fftw' when libfftw3f.so.3 could be dlopen'ed on this host (the reference's FFT library),
    else 'port' (the oracle's own transform)."""
    return lib().oracle_fft_backend().decode()


def threads(nthreads):
    return lib().oracle_threads(nthreads)


def cpu_deconvolve_simultaneous(psi, holder, nthreads=1):
    out = np.ascontiguousarray(psi, dtype=np.float32).copy()
    lib().oracle_deconvolve_simultaneous(_fp(out), holder.ws, nthreads)
    return out


def simultaneous_step(psi, holder, v_begin, v_end, nthreads=1):
    """Partial correction sum of views [v_begin, v_end) computed from psi (not applied)."""
    psi = np.ascontiguousarray(psi, dtype=np.float32)
    delta = np.empty_like(psi)
    lib().oracle_deconvolve_simultaneous_step(_fp(psi), holder.ws, v_begin, v_end, _fp(delta),
                                              nthreads)
    return delta


def spatial_convolve(image, kernel):
    image = np.ascontiguousarray(image, dtype=np.float32)
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    out = np.empty_like(image)
    idims, kdims = _dims(image), _dims(kernel)
    lib().oracle_spatial_convolve(_fp(image), _ip(idims), _fp(kernel), _ip(kdims), _fp(out))
    return out
