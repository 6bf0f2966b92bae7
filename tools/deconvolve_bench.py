"""RL deconvolution benchmark in the reference's own record format: the counterpart of
bench/bench_gpu_deconvolve_synthetic.cu (one `inplace_gpu_deconvolve`-type call on synthetic
stacks, :150-215) driven over the size ladder of python/sweep_gpu.py:144-160 /
python/generate_dims.py:4-48, one line per size with the columns of bench/logging.hpp:37-60

    n_devices alg_type dev_name n_repeats total_time_ms dims_x dims_y dims_z type_width_byte comment

    python tools/deconvolve_bench.py [-s 6] [-e 10] [-n 6] [-r 10] [--psf 15] [--resident]

Two figures per size: `gpu_deconvolve_abi` = the ABI call with host buffers, as the reference's
bench times it (uploads, PSF preparation, loop, download; cyclic policy `none` so that the
transform size is the stack size, like the reference's `all_on_device` run with its own padding
switched to as_is); `--resident` adds `gpu_deconvolve_resident` = the loop alone on the resident
engine.  The comment column carries mode, views and the derived iterations/s.
"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

from fft_bench import size_ladder  # noqa: E402
from libmultiviewnative_amd import native  # noqa: E402
from libmultiviewnative_amd.abi import WorkspaceHolder  # noqa: E402


def gaussian(edge, sigma):
    ax = np.arange(edge, dtype=np.float64) - edge // 2
    g = np.exp(-0.5 * (ax[:, None, None] / sigma[0]) ** 2 - 0.5 * (ax[None, :, None] / sigma[1]) ** 2
               - 0.5 * (ax[None, None, :] / sigma[2]) ** 2)
    return (g / g.sum()).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-s", "--start", type=int, default=6, help="first size: 2^s cubed")
    ap.add_argument("-e", "--end", type=int, default=10, help="one past the last exponent")
    ap.add_argument("-n", "--views", type=int, default=6)
    ap.add_argument("-r", "--repeats", type=int, default=10, help="RL iterations per call (the reference's -r)")
    ap.add_argument("--psf", type=int, default=15)
    ap.add_argument("--resident", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--max-gb", type=float, default=64.0, help="skip sizes whose stacks exceed this on the host")
    args = ap.parse_args()
    lib = native.lib()
    buf = ctypes.create_string_buffer(256)
    lib.l.getNameDeviceCUDA(args.device, buf)
    name = (buf.value.decode() or "gpu").replace(" ", "_")
    print("n_devices alg_type dev_name n_repeats total_time_ms stack_dims_x stack_dims_y stack_dims_z "
          "type_width_byte comment", flush=True)
    V = args.views
    for shape in size_ladder(args.start, args.end):
        vol = 4.0 * shape[0] * shape[1] * shape[2]
        if vol * (2 * V + 2) > args.max_gb * (1 << 30):
            continue
        rng = np.random.default_rng(17)
        views, k1s, k2s = [], [], []
        for v in range(V):
            views.append(rng.random(shape, dtype=np.float32) * 50 + 10)
            sig = [2.0, 2.0, 2.0]
            sig[v % 3] = 3.0
            k = gaussian(min(args.psf, *shape), sig)
            k1s.append(k)
            k2s.append(np.ascontiguousarray(k[::-1, ::-1, ::-1]))
        w = np.full(shape, 1.0 / V, np.float32)
        h = WorkspaceHolder(views, k1s, k2s, [w] * V, 0.006, 1e-3, args.repeats)  # lambda, minValue of the reference bench
        lib.set_pad_mode("none")
        secs = []
        for _ in range(3):
            psi = np.full(shape, np.float32(35.0), np.float32)
            secs.append(lib.gpu_deconvolve_inplace(psi, h, args.device))
        lib.set_pad_mode(None)
        warm = min(secs[1:])
        print(1, "gpu_deconvolve_abi", name, args.repeats, "%.3f" % (warm * 1e3), shape[0], shape[1], shape[2], 4,
              "all_on_device,first_call_ms=%.1f,nstacks=%d,it_per_s=%.2f" % (secs[0] * 1e3, V * 4, args.repeats / warm),
              flush=True)
        lib.check(lib.l.mvn_release_cached_engines())
        if args.resident:
            eng = lib.engine(shape, V, device=args.device)
            for v in range(V):
                eng.set_view(v, views[v], w, k1s[v], k2s[v])
            eng.set_psi(np.full(shape, np.float32(35.0), np.float32))
            eng.iterate(2, 0.006, 1e-3)
            t = time.perf_counter()
            eng.iterate(args.repeats, 0.006, 1e-3, sync=True)
            dt = time.perf_counter() - t
            eng.close()
            print(1, "gpu_deconvolve_resident", name, args.repeats, "%.3f" % (dt * 1e3), shape[0], shape[1], shape[2], 4,
                  "resident,NA,nstacks=%d,it_per_s=%.2f" % (V * 4, args.repeats / dt), flush=True)


if __name__ == "__main__":
    main()
