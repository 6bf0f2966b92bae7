import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The product uses the direct dim0 leg (mvn_dim0_direct.hpp) at every volume size.  The suites keep it to volumes
# that offer >= 98304 work items (512 x 512 planes and up, 256^3 in pieces) unless a test says otherwise, so that
# the many small-shape RL cases go on validating the fused FFT dim0 pass and its fixed-length kernels; the
# direct leg at small sizes has its own tests (test_direct_dim0_leg_*, the child-process cases, tools/fuzz_shapes.py).
os.environ.setdefault("MVN_DIM0_DIRECT_MIN_ITEMS", "98304")
os.environ.setdefault("MVN_DIM0_DIRECT_MIN_PLANE", "98304")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "asan: runs the host emulation of the kernels under AddressSanitizer + UBSan "
                                       "(CPU only; builds lib/libmvn_emu_asan.so on first use)")
