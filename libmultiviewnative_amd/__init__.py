"""MI355X-native multi-view Richardson-Lucy deconvolution behind the multiviewnative.h C-ABI.

The product is ``csrc/`` (HIP kernels + the C-ABI shared library ``libmultiviewnative.so``);
this Python package only loads that library through ctypes for tests, benches and the
multi-GPU launcher.  There is no CPU fallback: if the library is missing, imports of
``libmultiviewnative_amd.native`` fail loudly.
"""
__all__ = ["abi"]
