"""ctypes mirror of the reference C-ABI structs (``inc/multiviewnative.h:15-35``).

Lays the structs out exactly as JNA does for Fiji (x86-64 SysV, default alignment):
``view_data`` = 8 pointers (64 B); ``workspace`` = ``view_data*`` @0, ``unsigned short``
@8, ``double`` @16, ``float`` @24, ``int`` @28 (32 B), passed BY VALUE to the
deconvolve entry points (``inc/multiviewnative.h:50,66``).
"""
import ctypes as C

import numpy as np

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


class ViewData(C.Structure):
    _fields_ = [
        ("image_", c_float_p),
        ("kernel1_", c_float_p),
        ("kernel2_", c_float_p),
        ("weights_", c_float_p),
        ("image_dims_", c_int_p),
        ("kernel1_dims_", c_int_p),
        ("kernel2_dims_", c_int_p),
        ("weights_dims_", c_int_p),
    ]


class Workspace(C.Structure):
    _fields_ = [
        ("data_", C.POINTER(ViewData)),
        ("num_views_", C.c_ushort),
        ("lambda_", C.c_double),
        ("minValue_", C.c_float),
        ("num_iterations_", C.c_int),
    ]


assert C.sizeof(ViewData) == 64
assert C.sizeof(Workspace) == 32
assert Workspace.lambda_.offset == 16 and Workspace.minValue_.offset == 24
assert Workspace.num_iterations_.offset == 28


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def iptr(a):
    return a.ctypes.data_as(c_int_p)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class WorkspaceHolder:
    """Owns the numpy arrays a ``workspace`` points to (the ABI never copies or frees them)."""

    def __init__(self, views, kernels1, kernels2, weights, lambda_=0.006, min_value=1e-4,
                 iterations=1):
        self.views = [as_f32(v) for v in views]
        self.kernels1 = [as_f32(k) for k in kernels1]
        self.kernels2 = [as_f32(k) for k in kernels2]
        self.weights = [as_f32(w) for w in weights]
        n = len(self.views)
        assert len(self.kernels1) == n and len(self.kernels2) == n and len(self.weights) == n
        self._dims = []
        self.data = (ViewData * n)()
        for v in range(n):
            dims = [np.array(a.shape, dtype=np.int32) for a in
                    (self.views[v], self.kernels1[v], self.kernels2[v], self.weights[v])]
            self._dims.append(dims)
            d = self.data[v]
            d.image_, d.kernel1_, d.kernel2_, d.weights_ = (
                fptr(self.views[v]), fptr(self.kernels1[v]), fptr(self.kernels2[v]),
                fptr(self.weights[v]))
            d.image_dims_, d.kernel1_dims_, d.kernel2_dims_, d.weights_dims_ = (
                iptr(dims[0]), iptr(dims[1]), iptr(dims[2]), iptr(dims[3]))
        self.ws = Workspace()
        self.ws.data_ = C.cast(self.data, C.POINTER(ViewData))
        self.ws.num_views_ = n
        self.ws.lambda_ = float(lambda_)
        self.ws.minValue_ = float(min_value)
        self.ws.num_iterations_ = int(iterations)

    def with_iterations(self, iterations):
        self.ws.num_iterations_ = int(iterations)
        return self
