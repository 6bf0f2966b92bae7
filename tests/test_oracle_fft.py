"""Pins the oracle's FFT against the reference's own FFT-boundary tests and numpy pocketfft."""
import os

import numpy as np
import pytest

from oracle import binding as orc


def test_plan_store_ramp_roundtrip_exact():
    # tests/test_plan_store.cpp:83-142: 8^3 integer ramp, r2c -> c2r -> /512 equals input exactly
    x = np.arange(512, dtype=np.float32).reshape(8, 8, 8)
    spec = orc.rfft3_forward(x)
    back = orc.rfft3_backward(spec, 8) / np.float32(512)
    assert np.array_equal(back, x)


@pytest.mark.parametrize("shape", [(13, 17, 19), (16, 16, 16), (9, 27, 3), (27, 9, 81),
                                   (5, 25, 125), (7, 49, 7), (16, 18, 14), (10, 10, 10)])
def test_roundtrip_mse(shape):
    # tests/test_fftw_numerical_stability.cpp:32-664: ramp 0..N-1, roundtrip MSE < 1e-4
    n = int(np.prod(shape))
    x = np.arange(n, dtype=np.float32).reshape(shape)
    back = orc.rfft3_backward(orc.rfft3_forward(x), shape[2]) / np.float32(n)
    mse = float(np.mean((back.astype(np.float64) - x) ** 2))
    # the reference bound is absolute (1e-4) on ramps up to 16^3=4096; scale it for larger ramps
    bound = 1e-4 * max(1.0, (n / 4096.0) ** 2)
    assert mse < bound, mse


@pytest.mark.parametrize("shape", [(8, 8, 8), (13, 17, 19), (16, 18, 14), (6, 10, 15), (32, 20, 64),
                                   (4, 4, 7), (3, 5, 2), (64, 64, 64)])
def test_forward_matches_pocketfft(shape):
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape).astype(np.float32)
    ref = np.fft.rfftn(x.astype(np.float64))
    got = orc.rfft3_forward(x)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() / scale < 5e-6


@pytest.mark.parametrize("shape", [(8, 8, 8), (13, 17, 19), (16, 18, 14), (4, 4, 7), (32, 20, 64)])
def test_backward_matches_pocketfft(shape):
    rng = np.random.default_rng(2)
    x = rng.standard_normal(shape)
    spec = np.fft.rfftn(x).astype(np.complex64)
    ref = np.fft.irfftn(spec.astype(np.complex128), s=shape, axes=(0, 1, 2)) * np.prod(shape)
    got = orc.rfft3_backward(spec, shape[2])
    assert np.abs(got - ref).max() / np.abs(ref).max() < 5e-6


def test_threads_agree():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((12, 20, 18)).astype(np.float32)
    assert np.array_equal(orc.rfft3_forward(x, 1), orc.rfft3_forward(x, 4))


# ---- optional FFTW backend of the oracle (bench.py's cpu_baseline "kind": "fftw") ----------------
_FAKE_FFTW_C = r"""
/* Test double of libfftw3f.so.3: the FFTW entry points the oracle binds with dlsym, backed by the
 * oracle's own built-in transform, with a log of how they were called.  It checks the plumbing
 * (probe, plan cache, MEASURE planning on scratch memory, new-array execution, thread count); it
 * is NOT FFTW and is never used for a baseline. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
typedef struct { int d0, d1, d2, fwd; } plan;
static void (*fwd_fn)(float*, int, int, int, int);
static void (*bwd_fn)(float*, int, int, int, int);
static int g_threads = 1;
static void bind(void) {
  if (fwd_fn) return;
  void* h = dlopen(getenv("MVN_TEST_ORACLE_SO"), RTLD_NOW);
  *(void**)&fwd_fn = dlsym(h, "oracle_builtin_rfft3_forward");
  *(void**)&bwd_fn = dlsym(h, "oracle_builtin_rfft3_backward");
}
static void note(const char* what, int a, int b, int c, unsigned f) {
  FILE* fp = fopen(getenv("MVN_TEST_FFTW_LOG"), "a");
  fprintf(fp, "%s %d %d %d %u %d\n", what, a, b, c, f, g_threads);
  fclose(fp);
}
void* fftwf_plan_dft_r2c_3d(int a, int b, int c, float* in, float* out, unsigned flags) {
  bind(); in[0] = 12345.f; (void)out;  /* MEASURE may overwrite the planning arrays */
  plan* p = malloc(sizeof(plan)); p->d0 = a; p->d1 = b; p->d2 = c; p->fwd = 1; note("plan_r2c", a, b, c, flags); return p;
}
void* fftwf_plan_dft_c2r_3d(int a, int b, int c, float* in, float* out, unsigned flags) {
  bind(); in[0] = 12345.f; (void)out;
  plan* p = malloc(sizeof(plan)); p->d0 = a; p->d1 = b; p->d2 = c; p->fwd = 0; note("plan_c2r", a, b, c, flags); return p;
}
void fftwf_execute_dft_r2c(void* q, float* in, float* out) { plan* p = q; (void)out; fwd_fn(in, p->d0, p->d1, p->d2, g_threads); }
void fftwf_execute_dft_c2r(void* q, float* in, float* out) { plan* p = q; (void)out; bwd_fn(in, p->d0, p->d1, p->d2, g_threads); }
void fftwf_destroy_plan(void* q) { free(q); }
int fftwf_init_threads(void) { return 1; }
void fftwf_plan_with_nthreads(int n) { g_threads = n; }
"""

_FFTW_CHILD = r"""
import os, sys
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from libmultiviewnative_amd.abi import WorkspaceHolder
from oracle import binding as orc
from ref_fixtures import realistic_views
print(orc.fft_backend())
x = np.random.default_rng(1).standard_normal((6, 10, 12)).astype(np.float32)
np.save(sys.argv[2] + "_fft.npy", orc.rfft3_forward(x, 2))
_, views, k1, k2, w, psi0 = realistic_views((12, 10, 14), 2, (3, 3, 3))
h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 2)
np.save(sys.argv[2] + "_rl.npy", orc.cpu_deconvolve(psi0, h, 2))
"""


def test_fftw_backend_probe_with_a_test_double(tmp_path):
    import subprocess
    import sys
    from oracle import binding as orc
    assert orc.fft_backend() in ("port", "fftw")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "fake_fftw.c"
    src.write_text(_FAKE_FFTW_C)
    fake = tmp_path / "libfftw3f.so.3"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", str(fake), str(src), "-ldl"])
    log = tmp_path / "calls.log"
    log.write_text("")
    outs = {}
    for name, env_extra in (("fake", {"MVN_ORACLE_FFTW_LIB": str(fake)}), ("port", {"MVN_ORACLE_FFTW": "0"})):
        env = dict(os.environ, MVN_TEST_ORACLE_SO=os.path.join(root, "oracle", "_build", "libmvn_oracle.so"),
                   MVN_TEST_FFTW_LOG=str(log), **env_extra)
        r = subprocess.run([sys.executable, "-c", _FFTW_CHILD, root, str(tmp_path / name)], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = (r.stdout.strip(), np.load(str(tmp_path / name) + "_fft.npy"), np.load(str(tmp_path / name) + "_rl.npy"))
    assert outs["fake"][0] == "fftw" and outs["port"][0] == "port"
    # the double runs the built-in transform, so both routes agree bit for bit
    assert np.array_equal(outs["fake"][1], outs["port"][1]) and np.array_equal(outs["fake"][2], outs["port"][2])
    calls = [l.split() for l in log.read_text().splitlines()]
    plans = [c for c in calls if c[0].startswith("plan")]
    # one r2c + one c2r plan per (shape, threads), FFTW_MEASURE | FFTW_UNALIGNED (flags 0 | 2: the plans
    # are executed on caller arrays of any alignment through the new-array interface), thread count set
    assert sorted((c[0], c[1:4]) for c in plans) == sorted(
        [("plan_r2c", ["6", "10", "12"]), ("plan_c2r", ["6", "10", "12"]),
         ("plan_r2c", ["12", "10", "14"]), ("plan_c2r", ["12", "10", "14"])])
    assert all(c[4] == "2" and c[5] == "2" for c in plans)
