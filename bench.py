#!/usr/bin/env python3
"""Headline benchmark: multi-view Richardson-Lucy iterations/s on 512^3 float32, 6 views.

    python bench.py --gpus N --steps K --warmup W

A "step" is one RL iteration = one sweep over all views resident on a GPU (BASELINE.json
configs[2]: 512^3 f32, 6 views, 31^3 PSFs), with every stack, both PSF spectra per view and psi
already resident in HBM when the timed region starts.

N = 1   the reference's sequential (Gauss-Seidel) view sweep (src/gpu_deconvolve_methods.cuh:
        487-535) through the C-ABI engine of lib/libmultiviewnative.so.
N > 1   one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI); every rank
        holds 6 views of a 6N-view data set and a replica of psi; per iteration each rank
        computes its correction sum from the same psi and ONE all-reduce of the 512^3 float32
        delta combines them (simultaneous update).  Per-GPU work is fixed -> weak
        scaling; `value` counts 6-view sweeps per second over all ranks.

Rank 0 prints ONE JSON line (the driver's contract) with two extra objects:
  roofline      dominant kernel: algorithmic bytes per launch / average launch duration (HIP
                events on the engine's stream, taken inside the timed region) vs 8 TB/s HBM
  cpu_baseline  the oracle (CPU restatement of inplace_cpu_deconvolve) timed on this host on a
                bounded sample (N = 1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def gaussian_psf(edge, sigma):
    import numpy as np
    ax = np.arange(edge, dtype=np.float64) - edge // 2
    g = np.exp(-0.5 * (ax[:, None, None] / sigma[0]) ** 2 - 0.5 * (ax[None, :, None] / sigma[1]) ** 2
               - 0.5 * (ax[None, None, :] / sigma[2]) ** 2)
    return (g / g.sum()).astype(np.float32)


def kernel_bytes(kind, d0, d1, d2):
    """Algorithmic HBM bytes of ONE launch of each kernel kind."""
    vol = 4.0 * d0 * d1 * d2            # dense real volume == main half-spectrum array
    nyq = 8.0 * d0 * d1 if d2 % 2 == 0 else 0.0
    B = vol + nyq                       # the reference's in-place r2c footprint
    return {
        "rows_r2c": vol + B,            # read real volume, write half-spectrum (+ Nyquist plane)
        "rows_c2r": B + 2.5 * vol,      # read spectrum, write volume; + view (divide) or psi+weights (update): mean 1.5
        # c2r + pointwise + r2c in one pass: read spectrum, write spectrum, + view (divide) or
        # psi + weights in, psi out (update)
        "rows_fused_div": 2 * B + vol,
        "rows_fused_upd": 2 * B + 3 * vol,
        "axis1_fwd": 2 * vol,
        "axis1_inv": 2 * vol,
        "axis0_fused": 3 * vol,         # read data, read PSF spectrum, write data
        "axis0_fwd": 2 * vol,
        "axis0_inv": 2 * vol,
        "nyquist": 2 * nyq,
    }.get(kind, 0.0), B


def cpu_baseline(shape, psf_edge, n_views):
    """Time the CPU oracle on a bounded sample: 1 of the views, 1 iteration, all host cores; the
    per-iteration figure for `n_views` views follows by scaling (the loop is linear in views)."""
    import numpy as np
    from libmultiviewnative_amd.abi import WorkspaceHolder
    from oracle import binding as orc
    rng = np.random.default_rng(1)
    view = rng.random(shape, dtype=np.float32) * 50 + 10
    psf = gaussian_psf(psf_edge, (3.0, 2.0, 2.0))
    h = WorkspaceHolder([view], [psf], [np.ascontiguousarray(psf[::-1, ::-1, ::-1])],
                        [np.ones(shape, np.float32)], 0.006, 1e-4, 1)
    psi0 = np.full(shape, np.float32(view.mean()), np.float32)
    cores = orc.threads(-1)
    orc.cpu_deconvolve(psi0, h, -1)
    setup_s, loop_s = orc.last_timing()
    return {
        "value": 1.0 / (loop_s * n_views),
        "unit": "RL iterations/s (%d views)" % n_views,
        "cores": cores,
        "kind": "port",
        "sample": "%dx%dx%d f32, 1 of %d views, 1 iteration (loop %.2f s, PSF setup %.2f s excluded), "
                  "scaled x1/%d" % (shape[0], shape[1], shape[2], n_views, loop_s, setup_s, n_views),
    }


def small_parity(lib):
    """GPU vs CPU oracle on a small realistic case, reported beside the timing."""
    import numpy as np
    from libmultiviewnative_amd.abi import WorkspaceHolder
    from oracle import binding as orc
    from ref_fixtures import realistic_views
    shape = (64, 64, 64)
    _, views, k1, k2, w, psi0 = realistic_views(shape, 3, (9, 9, 9))
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 5)
    got = lib.gpu_deconvolve(psi0, h, 0).astype(np.float64)
    ref = orc.cpu_deconvolve(psi0, h, -1).astype(np.float64)
    d = got - ref
    return {"case": "64^3, 3 views, 9^3 PSF, 5 it vs CPU oracle",
            "max_rel": float(np.abs(d).max() / np.abs(ref).max()),
            "rms_rel": float(np.sqrt((d * d).mean()) / np.sqrt((ref * ref).mean()))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--views-per-gpu", type=int, default=6)
    ap.add_argument("--psf", type=int, default=31)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 4],
                    help="BASELINE.json configs[i]: 1 = 256^3 x 1 view x 15^3 PSF, 2 = 512^3 x 6 views x 31^3 "
                         "(default, the headline), 4 = 320x1920x1920 x 6 views x 31^3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel events in the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1,
                    help="rehearsal on a one-GPU box: put every rank on this device (needs --backend gloo)")
    ap.add_argument("--simultaneous", action="store_true",
                    help="run the simultaneous-update (Jacobi) loop even on one rank (what every rank of an N>1 run computes)")
    ap.add_argument("--no-side", action="store_true",
                    help="skip ms_per_fft and the small parity case (keeps rocprof --stats averages clean)")
    args = ap.parse_args()

    if args.config == 1:
        args.size, args.views_per_gpu, args.psf = [256, 256, 256], 1, 15
    elif args.config == 4:
        args.size, args.views_per_gpu, args.psf = [320, 1920, 1920], 6, 31
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    dist = None
    torch = None
    if world > 1:
        # torch first: the product library then binds to the HIP runtime torch already loaded
        import torch
        import torch.distributed as dist
        if args.all_ranks_on_device >= 0:
            local_rank = args.all_ranks_on_device
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    import numpy as np
    from libmultiviewnative_amd import native
    from libmultiviewnative_amd.sharded import SimultaneousDriver

    lib = native.lib()  # raises if the HIP library is missing: there is no fallback
    shape = tuple(args.size)
    V = args.views_per_gpu
    lam, minv = 0.006, 1e-4  # Fiji defaults (tests/tiff_fixtures.hpp:449-450)

    eng = lib.engine(shape, V, device=local_rank)
    rng = np.random.default_rng(1000 + rank)
    weights = np.full(shape, 1.0 / (V * world), np.float32)
    mean = 35.0  # the views are uniform in [10, 60); the same start value on every rank keeps the replicas identical
    for v in range(V):
        view = rng.random(shape, dtype=np.float32) * 50 + 10
        sig = [2.0, 2.0, 2.0]
        sig[v % 3] = 4.0
        psf = gaussian_psf(args.psf, sig)
        eng.set_view(v, view, weights, psf, np.ascontiguousarray(psf[::-1, ::-1, ::-1]))
        del view
    eng.set_psi(np.full(shape, np.float32(mean), np.float32))

    driver = None
    if world > 1:
        nfl = eng.psi_ptr()[1]
        delta = torch.zeros(nfl, dtype=torch.float32, device="cuda:%d" % local_rank)
        eng.bind_delta(delta.data_ptr())
        driver = SimultaneousDriver(eng, delta, dist,
                                    after_collective=lambda: torch.cuda.current_stream().synchronize())
    elif args.simultaneous:
        driver = SimultaneousDriver(eng, None, None)

    def run(steps):
        if driver is None:
            eng.iterate(steps, lam, minv, sync=True)
        else:
            driver.run(steps, lam, minv)

    def fence():
        eng.sync()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    if world > 1:
        # communicator set-up (lazy in RCCL) must never land in the timed region, even with --warmup 0
        dist.all_reduce(delta, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        delta.zero_()
    run(args.warmup)
    if not args.no_profile:
        # events around the launches of every 5th (view, iteration) of the timed region: enough
        # samples of every kernel, < 1 % perturbation (all launches: 3 %)
        eng.profile(5)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read() if not args.no_profile else {}
    eng.profile(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda:%d" % local_rank)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    psi = eng.get_psi()
    psi_ok = bool(np.isfinite(psi).all() and (psi > 0).all())

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        d0, d1, d2 = shape
        # dominant kernel by total time inside the timed region
        roofline = None
        if prof:
            kind = max(prof, key=lambda k: prof[k][0])
            tot_ms, n = prof[kind]
            kb, B = kernel_bytes(kind, d0, d1, d2)
            avg_ms = tot_ms / max(n, 1)
            achieved = kb / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": kind, "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": None, "avg_launch_ms": round(avg_ms, 4), "launches": n,
                        "bytes_per_launch": kb,
                        "per_kernel": {k: {"avg_ms": round(v[0] / v[1], 4), "launches": v[1],
                                           "GBps": round(kernel_bytes(k, d0, d1, d2)[0] / (v[0] / v[1] * 1e-3) / 1e9, 1)}
                                       for k, v in prof.items() if v[1]}}
            traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(traffic_file) and shape == (512, 512, 512):  # PMC passes were taken at 512^3
                try:
                    roofline["traffic"] = json.load(open(traffic_file)).get(kind)
                except Exception:
                    pass
        _, B = kernel_bytes("nyquist", d0, d1, d2)
        whole = 25.0 * B * V / (ms_per_step * 1e-3) / 1e9  # 25*B per (view, iteration)
        out = {
            "metric": "RL iterations/sec on 512^3 f32, 6 views",
            "value": round(value, 4),
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%dx%dx%d f32, %d views/GPU, %d^3 PSFs, lambda=0.006" % (d0, d1, d2, V, args.psf),
                       "views_total": V * world,
                       "update_mode": "sequential (reference order)" if driver is None else "simultaneous, single rank" if world == 1 else
                       "simultaneous + 1 all-reduce/iteration (backend %s)" % args.backend,
                       "unit_of_value": "%d-view sweeps per second over all ranks" % V},
            "whole_step_GBps_25B_model": round(whole, 1),
            "whole_step_frac_of_8TBps": round(whole / HBM_PEAK_GBS, 4),
            "psi_finite_positive": psi_ok,
            "roofline": roofline,
        }
    eng.close()

    if rank == 0 and world == 1 and not args.no_side:
        try:
            out["ms_per_fft"] = round(lib.fft3_time(shape, 0, 10, device=local_rank), 4)
            out["parity"] = small_parity(lib)
        except Exception as e:  # the headline number must survive a failing side measurement
            out["side_measurement_error"] = str(e)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(shape, args.psf, V)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
