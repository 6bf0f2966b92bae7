"""The kernels' own index arithmetic under AddressSanitizer + UBSan: the host emulation
(lib/libmvn_emu_asan.so, `make emu-asan`) executes the same workgroup bodies as the HIP build, so
an out-of-range tile / table / LDS index of a newly added fixed length shows up HERE, on the CPU,
instead of as a GPU fault (round 2: the clamp of unused last-stage items read out of range for
96 / 160 / 288).  GPU sanitizers are not available on the pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libmultiviewnative_amd", "csrc")
ASAN_SO = os.path.join(ROOT, "libmultiviewnative_amd", "lib", "libmvn_emu_asan.so")


def _runtime(name):
    p = subprocess.check_output(["gcc", "-print-file-name=" + name]).decode().strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.asan
def test_fixed_and_wave_row_kernels_under_asan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not (asan and ubsan):
        pytest.skip("no libasan / libubsan next to gcc")
    subprocess.check_call(["make", "-C", CSRC, "emu-asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, MVN_EMU_SO=ASAN_SO,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    # every compile-time specialised length (power-of-two, mixed-radix, long rows, split-window), the
    # wave-row passes, the direct dim0 leg (all tap counts, packed and split Nyquist) and the fused middle pass with the
    # line-layout last-axis passes (its LDS line buffers are a host vector here); -p no:cacheprovider: the child must not fight the parent over .pytest_cache
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_emu_engine.py"), "-q", "-x",
                        "-p", "no:cacheprovider", "-k", "mixed_radix or wave_row or long_rows or fixed or direct_dim0 or fused_middle_pass_in_the_simultaneous "
                              "or default_padding_policy_reaches or (slabs_run_the_fused and not None)"],  # (one of the two slab cases: 50 s each here)
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in r.stderr, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
