"""ctypes binding of the product shared library ``lib/libmultiviewnative.so``.

This is the only way Python touches the hot path: every call goes through the C-ABI declared
in ``include/multiviewnative.h`` / ``include/mvn_engine_api.h``.  There is no fallback -- if the
HIP library has not been built, :func:`lib` raises.

``Binding(path)`` can also wrap the test-only host emulation (``lib/libmvn_emu.so``); only
``tests/`` does that, to validate plans and index math on a box without a GPU.
"""
import ctypes as C
import os

import numpy as np

from .abi import Workspace, c_float_p, c_int_p, fptr, iptr

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
# MVN_PRODUCT_SO: timing-probe builds of the same HIP library (`make probe`); never a fallback
PRODUCT_SO = os.environ.get("MVN_PRODUCT_SO") or os.path.join(LIB_DIR, "libmultiviewnative.so")
EMU_SO = os.environ.get("MVN_EMU_SO") or os.path.join(LIB_DIR, "libmvn_emu.so")  # override: sanitizer build

REFERENCE_ABI_SYMBOLS = [
    "inplace_gpu_deconvolve", "inplace_gpu_convolution", "convolution3DfftCUDAInPlace",
    "convolution3DfftCUDAInPlace_core", "compute_quotient", "compute_final_values",
    "iterate_fft_plain", "iterate_fft_tikhonov",
    "selectDeviceWithHighestComputeCapability", "getCUDAcomputeCapabilityMinorVersion",
    "getCUDAcomputeCapabilityMajorVersion", "getNumDevicesCUDA", "getNameDeviceCUDA",
    "getMemDeviceCUDA",
]
ENGINE_ABI_SYMBOLS = [
    "mvn_last_error", "mvn_backend_name", "mvn_set_pad_mode", "mvn_get_pad_mode", "mvn_release_cached_engines", "mvn_deconvolve_submit", "mvn_deconvolve_wait", "mvn_psf_cache_counters", "mvn_split_launch_count", "mvn_mid_fused_launch_count", "mvn_multi_device_calls", "mvn_group_create", "mvn_group_destroy", "mvn_group_load", "mvn_group_iterate", "mvn_group_get_psi", "mvn_plan_store_add", "mvn_plan_store_has_key",
    "mvn_plan_store_size", "mvn_plan_store_empty", "mvn_plan_store_clear", "mvn_plan_describe",
    "mvn_fft3_r2c", "mvn_fft3_c2r", "mvn_fft3_time", "mvn_fft3_profile", "mvn_fft3_many_r2c", "mvn_fft3_many_time", "mvn_engine_create", "mvn_engine_destroy",
    "mvn_engine_set_view", "mvn_engine_set_psi", "mvn_engine_get_psi", "mvn_engine_iterate",
    "mvn_engine_compute_delta", "mvn_engine_apply_delta", "mvn_engine_delta_ptr",
    "mvn_engine_delta_chunks", "mvn_engine_delta_chunk_range", "mvn_engine_compute_delta_head",
    "mvn_engine_compute_delta_chunk", "mvn_engine_apply_delta_chunk",
    "mvn_engine_bind_delta", "mvn_engine_set_halo_hook", "mvn_engine_set_halo_planes", "mvn_engine_would_be_direct", "mvn_engine_poison_ptr", "mvn_engine_bind_poison", "mvn_engine_poison_get", "mvn_engine_poison_merge", "mvn_engine_copy_planes", "mvn_engine_psi_ptr", "mvn_engine_stream", "mvn_engine_sync", "mvn_engine_time_iterate",
    "mvn_engine_profile", "mvn_engine_profile_read", "mvn_kernel_kind_count",
    "mvn_kernel_kind_name", "mvn_engine_B",
    "mvn_slab_create", "mvn_slab_destroy", "mvn_slab_set_view", "mvn_slab_set_psi", "mvn_slab_get_psi",
    "mvn_slab_buffer_sizes", "mvn_slab_buffers", "mvn_slab_bind_buffers", "mvn_slab_begin",
    "mvn_slab_pack", "mvn_slab_mid", "mvn_slab_unpack", "mvn_slab_sync", "mvn_slab_stream",
]


class MvnError(RuntimeError):
    pass


def _dims(shape):
    return (C.c_int * 3)(*[int(s) for s in shape])


class _Missing:
    """A symbol an OLDER build of the library does not have (same-box A/B runs of a previous round's build through
    MVN_PRODUCT_SO): prototypes can be set, a call fails loudly."""

    def __init__(self, name):
        self.name = name

    def __call__(self, *a):
        raise MvnError("this build of the library has no %s" % self.name)


class _Tolerant:
    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)

    def __getattr__(self, name):
        try:
            return getattr(self._cdll, name)
        except AttributeError:
            m = _Missing(name)
            object.__setattr__(self, name, m)
            return m


class Binding:
    def __init__(self, path):
        if not os.path.exists(path):
            raise MvnError(
                "native library %s is missing -- build it with `python -c 'import "
                "__graft_entry__ as g; g.build()'` (there is no CPU fallback)" % path)
        self.path = path
        self.l = C.CDLL(path)
        if os.environ.get("MVN_PRODUCT_SO") and os.path.abspath(path) == os.path.abspath(os.environ["MVN_PRODUCT_SO"]):
            self.l = _Tolerant(self.l)  # an A/B build named explicitly may be an older one; the product itself is strict
        l = self.l
        i3 = C.POINTER(C.c_int)
        l.mvn_last_error.restype = C.c_char_p
        l.mvn_backend_name.restype = C.c_char_p
        l.mvn_split_launch_count.restype = C.c_long
        l.mvn_split_launch_count.argtypes = []
        l.mvn_mid_fused_launch_count.restype = C.c_long
        l.mvn_mid_fused_launch_count.argtypes = []
        l.mvn_multi_device_calls.restype = C.c_long
        l.mvn_multi_device_calls.argtypes = []
        l.mvn_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p)]
        l.mvn_group_destroy.argtypes = [C.c_void_p]
        l.mvn_group_load.argtypes = [C.c_void_p, C.POINTER(C.c_float), Workspace]
        l.mvn_group_iterate.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_float, C.POINTER(C.c_float)]
        l.mvn_group_get_psi.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.mvn_kernel_kind_name.restype = C.c_char_p
        l.mvn_kernel_kind_name.argtypes = [C.c_int]
        l.mvn_set_pad_mode.argtypes = [C.c_char_p]
        l.mvn_get_pad_mode.restype = C.c_char_p
        l.mvn_get_pad_mode.argtypes = []
        l.inplace_gpu_deconvolve.argtypes = [c_float_p, Workspace, C.c_int]
        l.inplace_gpu_deconvolve.restype = None
        l.mvn_deconvolve_submit.argtypes = [c_float_p, Workspace, C.c_int, C.POINTER(C.c_longlong)]
        l.mvn_deconvolve_wait.argtypes = [C.c_longlong]
        for n in ("inplace_gpu_convolution", "convolution3DfftCUDAInPlace"):
            getattr(l, n).argtypes = [c_float_p, c_int_p, c_float_p, c_int_p, C.c_int]
            getattr(l, n).restype = None
        l.convolution3DfftCUDAInPlace_core.argtypes = [C.c_void_p, c_int_p, C.c_void_p, c_int_p, C.c_int]
        l.convolution3DfftCUDAInPlace_core.restype = None
        l.compute_quotient.argtypes = [c_float_p, c_float_p, C.c_size_t, C.c_int]
        l.compute_quotient.restype = None
        l.compute_final_values.argtypes = [c_float_p, c_float_p, c_float_p, C.c_size_t, C.c_float,
                                           C.c_double, C.c_int]
        l.compute_final_values.restype = None
        l.iterate_fft_plain.argtypes = [c_float_p, c_float_p, c_float_p, c_int_p, c_int_p, C.c_int]
        l.iterate_fft_plain.restype = None
        l.iterate_fft_tikhonov.argtypes = [c_float_p, c_float_p, c_float_p, c_int_p, c_int_p,
                                           C.c_size_t, C.c_float, C.c_double, C.c_int]
        l.iterate_fft_tikhonov.restype = None
        l.getNameDeviceCUDA.argtypes = [C.c_int, C.c_char_p]
        l.getNameDeviceCUDA.restype = None
        l.getMemDeviceCUDA.argtypes = [C.c_int]
        l.getMemDeviceCUDA.restype = C.c_longlong
        l.mvn_plan_store_add.argtypes = [C.c_int, i3]
        l.mvn_plan_store_has_key.argtypes = [C.c_int, i3]
        l.mvn_plan_describe.argtypes = [C.c_int, i3, i3]
        l.mvn_fft3_r2c.argtypes = [C.c_int, i3, c_float_p, c_float_p]
        l.mvn_fft3_c2r.argtypes = [C.c_int, i3, c_float_p, c_float_p]
        l.mvn_fft3_time.argtypes = [C.c_int, i3, C.c_int, C.c_int, C.POINTER(C.c_float)]
        l.mvn_fft3_profile.argtypes = [C.c_int, i3, C.c_int, C.c_int, C.POINTER(C.c_float),
                                       C.POINTER(C.c_double)]
        l.mvn_fft3_many_r2c.argtypes = [C.c_int, i3, C.c_int, c_float_p, c_float_p]
        l.mvn_fft3_many_time.argtypes = [C.c_int, i3, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        l.mvn_engine_create.argtypes = [C.c_int, i3, C.c_int, C.POINTER(C.c_void_p)]
        l.mvn_engine_destroy.argtypes = [C.c_void_p]
        l.mvn_engine_set_view.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p, c_float_p, i3,
                                          c_float_p, i3]
        l.mvn_engine_set_psi.argtypes = [C.c_void_p, c_float_p]
        l.mvn_engine_get_psi.argtypes = [C.c_void_p, c_float_p]
        l.mvn_engine_iterate.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_float]
        l.mvn_engine_compute_delta.argtypes = [C.c_void_p, C.c_double, C.c_float]
        l.mvn_engine_apply_delta.argtypes = [C.c_void_p]
        l.mvn_engine_delta_chunks.argtypes = [C.c_void_p, C.c_int]
        l.mvn_engine_delta_chunk_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t),
                                                   C.POINTER(C.c_size_t)]
        l.mvn_engine_compute_delta_head.argtypes = [C.c_void_p, C.c_double, C.c_float]
        l.mvn_engine_compute_delta_chunk.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.mvn_engine_apply_delta_chunk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        l.mvn_engine_delta_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.mvn_engine_psi_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.mvn_engine_bind_delta.argtypes = [C.c_void_p, C.c_void_p]
        l.mvn_engine_set_halo_hook.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        l.mvn_engine_copy_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        l.mvn_engine_set_halo_planes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.mvn_engine_would_be_direct.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        l.mvn_engine_poison_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        l.mvn_engine_bind_poison.argtypes = [C.c_void_p, C.c_void_p]
        l.mvn_engine_poison_get.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
        l.mvn_engine_poison_merge.argtypes = [C.c_void_p, C.c_uint]
        l.mvn_engine_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        l.mvn_engine_sync.argtypes = [C.c_void_p]
        l.mvn_engine_time_iterate.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_float,
                                              C.POINTER(C.c_float)]
        l.mvn_engine_profile.argtypes = [C.c_void_p, C.c_int]
        l.mvn_engine_profile_read.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double),
                                              C.POINTER(C.c_long)]
        vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
        l.mvn_slab_create.argtypes = [C.c_int, i3, C.c_int, C.c_int, C.c_int, vpp]
        l.mvn_slab_destroy.argtypes = [vp]
        l.mvn_slab_set_view.argtypes = [vp, C.c_int, c_float_p, c_float_p, c_float_p, i3, c_float_p, i3]
        l.mvn_slab_set_psi.argtypes = [vp, c_float_p]
        l.mvn_slab_get_psi.argtypes = [vp, c_float_p]
        l.mvn_slab_buffer_sizes.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        l.mvn_slab_buffers.argtypes = [vp, vpp, vpp, vpp, vpp]
        l.mvn_slab_bind_buffers.argtypes = [vp, vp, vp, vp, vp]
        l.mvn_slab_begin.argtypes = [vp]
        l.mvn_slab_pack.argtypes = [vp, C.c_int, C.c_int]
        l.mvn_slab_mid.argtypes = [vp, C.c_int, C.c_int]
        l.mvn_slab_unpack.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_float, C.c_int]
        l.mvn_slab_sync.argtypes = [vp]
        l.mvn_slab_stream.argtypes = [vp, vpp]
        l.mvn_engine_B.argtypes = [C.c_void_p]
        l.mvn_engine_B.restype = C.c_size_t

    # ---- helpers -------------------------------------------------------------------------
    def check(self, rc):
        if rc < 0:
            raise MvnError(self.l.mvn_last_error().decode())
        return rc

    def backend_name(self):
        return self.l.mvn_backend_name().decode()

    def set_pad_mode(self, mode):
        """'zero' | 'zero_exact' | 'none' | None (back to MVN_PAD_MODE / the default)."""
        self.check(self.l.mvn_set_pad_mode(mode.encode() if mode else None))

    def get_pad_mode(self):
        """The mode selected with set_pad_mode, None when the environment / default decides."""
        return self.l.mvn_get_pad_mode().decode() or None

    # ---- reference ABI, numpy in / numpy out ----------------------------------------------
    def gpu_deconvolve(self, psi, holder, device=0, pad_mode="none"):
        """inplace_gpu_deconvolve on a copy of psi.  `pad_mode` defaults to the CPU path's cyclic
        policy, the one the oracle implements (the library's own default is 'zero', the reference
        GPU entry's); pass pad_mode=False to leave the process-wide setting alone.  The setting
        found on entry is restored on exit (the switch is process-wide: not for concurrent callers)."""
        out = np.ascontiguousarray(psi, dtype=np.float32).copy()
        before = self.get_pad_mode()
        if pad_mode is not False:
            self.set_pad_mode(pad_mode)
        try:
            self.l.inplace_gpu_deconvolve(fptr(out), holder.ws, device)
        finally:
            if pad_mode is not False:
                self.set_pad_mode(before)
        return out

    def deconvolve_submit(self, psi, holder, device=0):
        """mvn_deconvolve_submit: starts inplace_gpu_deconvolve on `psi` (C-contiguous float32, updated in
        place by the time deconvolve_wait returns) and returns the ticket.  `psi` and `holder` must be
        kept alive and untouched until then."""
        if not (psi.flags["C_CONTIGUOUS"] and psi.dtype == np.float32):
            raise ValueError("psi must be a C-contiguous float32 array")
        t = C.c_longlong(0)
        self.check(self.l.mvn_deconvolve_submit(fptr(psi), holder.ws, device, C.byref(t)))
        return t.value

    def deconvolve_wait(self, ticket):
        self.check(self.l.mvn_deconvolve_wait(ticket))

    def gpu_deconvolve_inplace(self, psi, holder, device=0):
        """The ABI call exactly as a host program makes it: `psi` (C-contiguous float32) is updated
        in place, the process-wide padding policy applies.  Returns the seconds spent inside the
        call (what the reference's bench times, bench/bench_gpu_deconvolve_synthetic.cu:190-203)."""
        import time
        if not (psi.flags["C_CONTIGUOUS"] and psi.dtype == np.float32):
            raise ValueError("psi must be a C-contiguous float32 array")
        t = time.perf_counter()
        self.l.inplace_gpu_deconvolve(fptr(psi), holder.ws, device)
        return time.perf_counter() - t

    def gpu_convolution(self, image, kernel, device=0, legacy=False):
        im = np.ascontiguousarray(image, dtype=np.float32).copy()
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        idims = np.array(im.shape, np.int32)
        kdims = np.array(k.shape, np.int32)
        f = self.l.convolution3DfftCUDAInPlace if legacy else self.l.inplace_gpu_convolution
        f(fptr(im), iptr(idims), fptr(k), iptr(kdims), device)
        return im

    def compute_quotient(self, view, blurred, device=0):
        view = np.ascontiguousarray(view, dtype=np.float32)
        out = np.ascontiguousarray(blurred, dtype=np.float32).copy()
        self.l.compute_quotient(fptr(view), fptr(out), out.size, device)
        return out

    def compute_final_values(self, psi, integral, weight, min_value, lambda_, device=0):
        psi = np.ascontiguousarray(psi, dtype=np.float32).copy()
        integral = np.ascontiguousarray(integral, dtype=np.float32)
        weight = np.ascontiguousarray(weight, dtype=np.float32)
        self.l.compute_final_values(fptr(psi), fptr(integral), fptr(weight), psi.size, min_value,
                                    lambda_, device)
        return psi

    def psf_cache_counters(self):
        out = (C.c_long * 2)()
        self.check(self.l.mvn_psf_cache_counters(out))
        return int(out[0]), int(out[1])

    def iterate_fft(self, image, kernel, min_value=1e-4, lambda_=None, device=0):
        """iterate_fft_plain (lambda_ None; the reference fixes minValue = 1e-4 there) or
        iterate_fft_tikhonov: one legacy RL step on a single stack."""
        im = np.ascontiguousarray(image, dtype=np.float32)
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        out = np.full_like(im, np.nan)
        idims = np.array(im.shape, np.int32)
        kdims = np.array(k.shape, np.int32)
        if lambda_ is None:
            self.l.iterate_fft_plain(fptr(im), fptr(k), fptr(out), iptr(idims), iptr(kdims), device)
        else:
            self.l.iterate_fft_tikhonov(fptr(im), fptr(k), fptr(out), iptr(idims), iptr(kdims),
                                        im.size, min_value, lambda_, device)
        return out

    # ---- transforms ------------------------------------------------------------------------
    def rfft3(self, x, device=0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        d0, d1, d2 = x.shape
        spec = np.empty((d0, d1, d2 // 2 + 1), dtype=np.complex64)
        self.check(self.l.mvn_fft3_r2c(device, _dims(x.shape), fptr(x), fptr(spec.view(np.float32))))
        return spec

    def irfft3(self, spec, d2, device=0):
        spec = np.ascontiguousarray(spec, dtype=np.complex64)
        d0, d1, nc = spec.shape
        assert nc == d2 // 2 + 1
        out = np.empty((d0, d1, d2), dtype=np.float32)
        self.check(self.l.mvn_fft3_c2r(device, _dims((d0, d1, d2)), fptr(spec.view(np.float32)), fptr(out)))
        return out

    def fft3_time(self, shape, direction=0, reps=10, device=0):
        ms = C.c_float(0)
        self.check(self.l.mvn_fft3_time(device, _dims(shape), direction, reps, C.byref(ms)))
        return ms.value

    def rfft3_many(self, stacks, device=0):
        """Forward r2c transform of [batch][d0][d1][d2] stacks through one plan."""
        x = np.ascontiguousarray(stacks, dtype=np.float32)
        batch, d0, d1, d2 = x.shape
        out = np.empty((batch, d0, d1, d2 // 2 + 1), np.complex64)
        self.check(self.l.mvn_fft3_many_r2c(device, _dims((d0, d1, d2)), batch, fptr(x),
                                            out.ctypes.data_as(c_float_p)))
        return out

    def fft3_many_time(self, shape, batch, direction=0, reps=5, device=0):
        ms = C.c_float(0)
        self.check(self.l.mvn_fft3_many_time(device, _dims(shape), batch, direction, reps, C.byref(ms)))
        return ms.value

    def fft3_profile(self, shape, direction=0, reps=10, device=0):
        ms = C.c_float(0)
        n = self.l.mvn_kernel_kind_count()
        per = (C.c_double * n)()
        self.check(self.l.mvn_fft3_profile(device, _dims(shape), direction, reps, C.byref(ms), per))
        return ms.value, {self.l.mvn_kernel_kind_name(k).decode(): per[k] for k in range(n) if per[k] > 0}

    def plan_describe(self, shape, device=0):
        out = (C.c_int * 12)()
        self.check(self.l.mvn_plan_describe(device, _dims(shape), out))
        keys = ["h", "C", "RP", "even", "rows_T", "ax1_T", "ax0_T", "n_stages", "fx_rows", "fx_ax1",
                "fx_ax0", "reserved"]
        return dict(zip(keys, list(out)))

    def engine(self, shape, num_views, device=0):
        return EngineHandle(self, shape, num_views, device)

    def slab_engine(self, shape, nranks, rank, num_views, device=0):
        return SlabHandle(self, shape, nranks, rank, num_views, device)

    def group(self, devices, shape, halo_planes, num_views):
        return GroupHandle(self, devices, shape, halo_planes, num_views)


class GroupHandle:
    """One volume as dim0 slabs on several devices of this process (``mvn_group_*``; what MVN_DEVICES runs inside
    ``inplace_gpu_deconvolve``), stacks resident between ``load`` and ``get_psi``."""

    def __init__(self, binding, devices, shape, halo_planes, num_views):
        self.b = binding
        self.shape = tuple(int(s) for s in shape)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        binding.check(binding.l.mvn_group_create(devs, len(devices), _dims(shape), int(halo_planes), int(num_views),
                                                 C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.b.l.mvn_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, psi, holder):
        """psi and the stacks of a ``WorkspaceHolder`` (extents == the group's)"""
        psi = np.ascontiguousarray(psi, dtype=np.float32)
        assert psi.shape == self.shape
        self.b.check(self.b.l.mvn_group_load(self.h, fptr(psi), holder.ws))

    def iterate(self, iterations, lam, min_value):
        """blocking; returns the wall time of the sweeps in ms"""
        ms = C.c_float(0)
        self.b.check(self.b.l.mvn_group_iterate(self.h, int(iterations), float(lam), float(min_value), C.byref(ms)))
        return ms.value

    def get_psi(self):
        out = np.empty(self.shape, np.float32)
        self.b.check(self.b.l.mvn_group_get_psi(self.h, fptr(out)))
        return out


class EngineHandle:
    """Resident RL engine (``mvn_engine_*``)."""

    def __init__(self, binding, shape, num_views, device=0):
        self.b = binding
        self.shape = tuple(int(s) for s in shape)
        self.num_views = num_views
        self.device = device
        h = C.c_void_p()
        binding.check(binding.l.mvn_engine_create(device, _dims(shape), num_views, C.byref(h)))
        self.h = h
        self._keep = []

    def close(self):
        if self.h:
            self.b.l.mvn_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_view(self, v, image, weights, kernel1, kernel2):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (image, weights, kernel1, kernel2)]
        assert a[0].shape == self.shape and a[1].shape == self.shape
        self.b.check(self.b.l.mvn_engine_set_view(self.h, v, fptr(a[0]), fptr(a[1]), fptr(a[2]),
                                                  _dims(a[2].shape), fptr(a[3]), _dims(a[3].shape)))

    def set_psi(self, psi):
        p = np.ascontiguousarray(psi, dtype=np.float32)
        assert p.shape == self.shape
        self.b.check(self.b.l.mvn_engine_set_psi(self.h, fptr(p)))

    def get_psi(self):
        out = np.empty(self.shape, np.float32)
        self.b.check(self.b.l.mvn_engine_get_psi(self.h, fptr(out)))
        return out

    def iterate(self, iterations, lambda_, min_value, sync=True):
        self.b.check(self.b.l.mvn_engine_iterate(self.h, iterations, lambda_, min_value))
        if sync:
            self.sync()

    def time_iterate(self, iterations, lambda_, min_value):
        ms = C.c_float(0)
        self.b.check(self.b.l.mvn_engine_time_iterate(self.h, iterations, lambda_, min_value, C.byref(ms)))
        return ms.value

    def compute_delta(self, lambda_, min_value):
        self.b.check(self.b.l.mvn_engine_compute_delta(self.h, lambda_, min_value))

    def apply_delta(self):
        self.b.check(self.b.l.mvn_engine_apply_delta(self.h))

    # the same step in pieces (all-reduce under compute): see include/mvn_engine_api.h
    def delta_chunks(self, wanted):
        return self.b.check(self.b.l.mvn_engine_delta_chunks(self.h, int(wanted)))

    def delta_chunk_range(self, c, n):
        a, cnt = C.c_size_t(), C.c_size_t()
        self.b.check(self.b.l.mvn_engine_delta_chunk_range(self.h, c, n, C.byref(a), C.byref(cnt)))
        return a.value, cnt.value

    def compute_delta_head(self, lambda_, min_value):
        self.b.check(self.b.l.mvn_engine_compute_delta_head(self.h, lambda_, min_value))

    def compute_delta_chunk(self, c, n):
        self.b.check(self.b.l.mvn_engine_compute_delta_chunk(self.h, c, n))

    def apply_delta_chunk(self, c, n, feed_next):
        self.b.check(self.b.l.mvn_engine_apply_delta_chunk(self.h, c, n, 1 if feed_next else 0))

    def _ptr(self, fn):
        p, n = C.c_void_p(), C.c_size_t()
        self.b.check(fn(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def delta_ptr(self):
        return self._ptr(self.b.l.mvn_engine_delta_ptr)

    def bind_delta(self, dev_ptr):
        self.b.check(self.b.l.mvn_engine_bind_delta(self.h, C.c_void_p(dev_ptr)))

    HALO_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int)

    def set_halo_hook(self, fn, drain=True, post=False):
        """fn(spectrum_ptr, view, conv) before every dim0 leg (mvn_engine_set_halo_hook); None switches it off.
        drain=False: fn is called without waiting for the engine's stream and must order its work on it.
        post=True: fn is called again behind the leg with conv + 2, where the slabs merge their poison words."""
        if fn is None:
            self._halo_cb = None
            self.b.check(self.b.l.mvn_engine_set_halo_hook(self.h, None, None, 1))
            return
        self._halo_cb = self.HALO_FN(lambda user, spectrum, view, conv: fn(spectrum, view, conv))
        self.b.check(self.b.l.mvn_engine_set_halo_hook(self.h, C.cast(self._halo_cb, C.c_void_p), None,
                                                       (1 if drain else 0) | (2 if post else 0)))

    def would_be_direct(self, kernel_shape):
        rc = self.b.l.mvn_engine_would_be_direct(self.h, _dims(kernel_shape))
        if rc < 0:
            self.b.check(rc)
        return rc == 1

    def set_halo_planes(self, planes, split=False):
        self.b.check(self.b.l.mvn_engine_set_halo_planes(self.h, int(planes), 1 if split else 0))

    def bind_poison(self, dev_ptr):
        self.b.check(self.b.l.mvn_engine_bind_poison(self.h, C.c_void_p(dev_ptr)))

    def poison_get(self):
        v = C.c_uint(0)
        self.b.check(self.b.l.mvn_engine_poison_get(self.h, C.byref(v)))
        return v.value

    def poison_merge(self, value):
        self.b.check(self.b.l.mvn_engine_poison_merge(self.h, C.c_uint(int(value))))

    def copy_planes(self, spectrum, plane0, nplanes, buffer_ptr, to_buffer, host_buffer=False, wait=True):
        self.b.check(self.b.l.mvn_engine_copy_planes(self.h, C.c_void_p(spectrum), plane0, nplanes,
                                                     C.c_void_p(buffer_ptr),
                                                     (1 if to_buffer else 0) | (2 if host_buffer else 0) |
                                                     (0 if wait else 4)))

    def psi_ptr(self):
        return self._ptr(self.b.l.mvn_engine_psi_ptr)

    def stream(self):
        p = C.c_void_p()
        self.b.check(self.b.l.mvn_engine_stream(self.h, C.byref(p)))
        return p.value

    def sync(self):
        self.b.check(self.b.l.mvn_engine_sync(self.h))

    def profile(self, enable):
        """enable: False/0 off, True/1 every launch, n > 1 the launches of every n-th (view, iteration)."""
        self.b.check(self.b.l.mvn_engine_profile(self.h, int(enable)))

    def profile_read(self):
        out = {}
        for k in range(self.b.l.mvn_kernel_kind_count()):
            ms, n = C.c_double(), C.c_long()
            self.b.check(self.b.l.mvn_engine_profile_read(self.h, k, C.byref(ms), C.byref(n)))
            out[self.b.l.mvn_kernel_kind_name(k).decode()] = (ms.value, n.value)
        return out

    def B(self):
        return self.b.l.mvn_engine_B(self.h)


_product = None


def lib():
    """The product library.  Raises if it has not been built (no fallback)."""
    global _product
    if _product is None:
        _product = Binding(PRODUCT_SO)
    return _product


class SlabHandle:
    """Slab-decomposed RL engine (``mvn_slab_*``): this rank's planes of a volume of `shape`."""

    def __init__(self, binding, shape, nranks, rank, num_views, device=0):
        self.b = binding
        self.shape = tuple(int(s) for s in shape)
        self.nranks, self.rank, self.num_views = nranks, rank, num_views
        self.slab_shape = (self.shape[0] // nranks, self.shape[1], self.shape[2])
        h = C.c_void_p()
        binding.check(binding.l.mvn_slab_create(device, _dims(self.shape), nranks, rank, num_views,
                                                C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.b.l.mvn_slab_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _slab(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.shape != self.slab_shape:
            raise ValueError("expected this rank's slab %r, got %r" % (self.slab_shape, a.shape))
        return a

    def set_view(self, v, image_slab, weights_slab, kernel1, kernel2):
        im, w = self._slab(image_slab), self._slab(weights_slab)
        k1 = np.ascontiguousarray(kernel1, dtype=np.float32)
        k2 = np.ascontiguousarray(kernel2, dtype=np.float32)
        self.b.check(self.b.l.mvn_slab_set_view(self.h, v, fptr(im), fptr(w), fptr(k1), _dims(k1.shape),
                                                fptr(k2), _dims(k2.shape)))

    def set_psi(self, psi_slab):
        self.b.check(self.b.l.mvn_slab_set_psi(self.h, fptr(self._slab(psi_slab))))
        self.b.check(self.b.l.mvn_slab_begin(self.h))

    def get_psi(self):
        out = np.empty(self.slab_shape, np.float32)
        self.b.check(self.b.l.mvn_slab_get_psi(self.h, fptr(out)))
        return out

    def buffer_sizes(self):
        a, b = C.c_size_t(0), C.c_size_t(0)
        self.b.check(self.b.l.mvn_slab_buffer_sizes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def buffers(self):
        p = [C.c_void_p() for _ in range(4)]
        self.b.check(self.b.l.mvn_slab_buffers(self.h, *[C.byref(x) for x in p]))
        return [x.value for x in p]

    def bind_buffers(self, a_main, b_main, a_nyq, b_nyq):
        self.b.check(self.b.l.mvn_slab_bind_buffers(self.h, a_main, b_main, a_nyq, b_nyq))

    def pack(self, v, conv):
        self.b.check(self.b.l.mvn_slab_pack(self.h, v, conv))

    def mid(self, v, conv):
        self.b.check(self.b.l.mvn_slab_mid(self.h, v, conv))

    def unpack(self, v, conv, lambda_, min_value, feed_next):
        self.b.check(self.b.l.mvn_slab_unpack(self.h, v, conv, lambda_, min_value, 1 if feed_next else 0))

    def sync(self):
        self.b.check(self.b.l.mvn_slab_sync(self.h))
