#!/usr/bin/env python3
"""Headline benchmark: multi-view Richardson-Lucy iterations/s on 512^3 float32, 6 views.

    python bench.py --gpus N --steps K --warmup W [--config 1|2|3|4]

A "step" is one RL iteration = one update of psi from ALL views of ONE problem (BASELINE.json
configs[2] by default: 512^3 f32, 6 views, 31^3 PSFs), with every stack, both PSF spectra per
view and psi resident in HBM when the timed region starts.  `value` = iterations/s of the whole
problem at every N -> strong scaling.

N = 1   the reference's sequential (Gauss-Seidel) view sweep (src/gpu_deconvolve_methods.cuh:
        487-535) through the C-ABI engine of lib/libmultiviewnative.so -- the headline number.
N > 1   one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The views of
        the SAME problem are partitioned over the ranks (6 views: 3+3, 2+2+1+1, and at N = 8 six
        ranks with one view each plus two that only hold a replica of psi); every rank computes
        the correction of its views from the same psi and ONE all-reduce of the psi-sized delta
        per iteration combines them (simultaneous / Jacobi update), issued in
        dim0 chunks under the compute (libmultiviewnative_amd/sharded.py).
        `python bench.py --gpus N` starts its N ranks itself (children are spawned before the
        parent touches the GPU); under `python -m torch.distributed.run ... bench.py --gpus N`
        it is one of the ranks.

--config 1  256^3, 1 view, 15^3 PSF            (BASELINE.json configs[1])
--config 2  512^3, 6 views, 31^3 PSFs          (configs[2], default, the metric's configuration)
--config 3  512^3, 8 views (1 per GPU at N=8)  (configs[3])
--config 4  320x1920x1920, 6 views, 31^3 PSFs  (configs[4]; 2+2+1+1 at N=4)

Rank 0 prints ONE JSON line (the driver's contract) with extra objects:
  roofline      dominant kernel: algorithmic bytes per launch / average launch duration (HIP
                events on the engine's stream, taken inside the timed region) vs 8 TB/s HBM
  cpu_baseline  the oracle (CPU restatement of inplace_cpu_deconvolve) timed on this host on a
                bounded sample: ONE view update of the timed problem, started from the psi the
                timed region left (rank 0 only)
  parity        that same view update through the product's reference entry point
                (inplace_gpu_deconvolve, host buffers) compared with the oracle's result
  abi_end_to_end  one inplace_gpu_deconvolve call with host buffers (what the reference's
                bench/bench_gpu_deconvolve_synthetic.cu:190-203 times), PCIe included; never `value`
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
LAMBDA, MIN_VALUE = 0.006, 1e-4  # Fiji defaults (tests/tiff_fixtures.hpp:449-450)

CONFIGS = {
    1: dict(size=[256, 256, 256], views=1, psf=15),
    2: dict(size=[512, 512, 512], views=6, psf=31),
    3: dict(size=[512, 512, 512], views=8, psf=31),
    4: dict(size=[320, 1920, 1920], views=6, psf=31),
}


def gaussian_psf(edge, sigma):
    import numpy as np
    ax = np.arange(edge, dtype=np.float64) - edge // 2
    g = np.exp(-0.5 * (ax[:, None, None] / sigma[0]) ** 2 - 0.5 * (ax[None, :, None] / sigma[1]) ** 2
               - 0.5 * (ax[None, None, :] / sigma[2]) ** 2)
    return (g / g.sum()).astype(np.float32)


def make_view(shape, v, psf_edge):
    """View v of the synthetic problem: the same on every rank and in every leg of the bench."""
    import numpy as np
    rng = np.random.default_rng(1000 + v)
    view = rng.random(shape, dtype=np.float32) * 50 + 10
    sig = [2.0, 2.0, 2.0]
    sig[v % 3] = 4.0
    edge = min(psf_edge, *shape)
    psf = gaussian_psf(edge, sig)
    return view, psf, np.ascontiguousarray(psf[::-1, ::-1, ::-1])


def start_value():
    return 35.0  # the views are uniform in [10, 60)


def kernel_bytes(kind, d0, d1, d2, psf_planes=31):
    """Algorithmic HBM bytes of ONE launch of each kernel kind."""
    vol = 4.0 * d0 * d1 * d2            # dense real volume == main half-spectrum array
    nyq = 8.0 * d0 * d1 if d2 % 2 == 0 else 0.0
    B = vol + nyq                       # the reference's in-place r2c footprint
    return {
        "rows_r2c": vol + B,            # read real volume, write half-spectrum (+ Nyquist plane)
        "rows_c2r": B + 3 * vol,        # read spectrum, psi, weights; write delta (un-chunked DELTA pass)
        # c2r + pointwise + r2c in one pass: read spectrum, write spectrum, + view (divide) or
        # psi + weights in, psi out (update)
        "rows_fused_div": 2 * B + vol,
        "rows_fused_upd": 2 * B + 3 * vol,
        "axis1_fwd": 2 * vol,
        "axis1_inv": 2 * vol,
        "axis0_fused": 3 * vol,         # read data, read PSF spectrum, write data
        # direct dim0 leg: read data, read the PSF's planes after the dim1/dim2 transforms, write data
        "axis0_direct": (2.0 + float(psf_planes) / d0) * vol,
        # dim1 forward + direct dim0 leg + dim1 inverse in ONE pass over the line layout (csrc/mvn_mid_fused.hpp):
        # read the half-spectrum, read the PSF's planes, write the half-spectrum
        "mid_fused": (2.0 + float(psf_planes) / d0) * vol,
        "axis0_fwd": 2 * vol,
        "axis0_inv": 2 * vol,
        "nyquist": 2 * nyq,
    }.get(kind, 0.0), B


class stdout_to_stderr:
    """Route the process's stdout (file descriptor 1, C libraries included) to stderr for a while:
    RCCL prints a version banner on stdout when its first communicator comes up, and rank 0's
    stdout must carry the one JSON line only."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as children of this
    process, which itself never touches the GPU (no HIP / torch.cuda call before or after), let
    rank 0's JSON line through on stdout and leave with the children's status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)]
    cmd += sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def one_view_check(lib, shape, psf_edge, n_views, psi_t, device, want_cpu):
    """ONE view update (view 0 of the timed problem, one iteration) started from the psi the timed
    region left: through the product's reference entry point on host buffers, and -- timed -- through
    the CPU oracle.  Gives the parity figures and the bounded CPU baseline sample in one go."""
    import numpy as np
    from libmultiviewnative_amd.abi import WorkspaceHolder
    view, k1, k2 = make_view(shape, 0, psf_edge)
    w = np.full(shape, 1.0 / n_views, np.float32)
    h = WorkspaceHolder([view], [k1], [k2], [w], LAMBDA, MIN_VALUE, 1)
    lib.set_pad_mode("none")  # the CPU path's policy (cyclic on image_dims_): the parity target
    got = lib.gpu_deconvolve(psi_t, h, device)
    lib.set_pad_mode(None)
    lib.check(lib.l.mvn_release_cached_engines())
    out = {"gpu_checksum": float(got.astype(np.float64).sum())}
    if not want_cpu:
        return out, None
    from oracle import binding as orc
    cores = orc.threads(-1)
    ref = orc.cpu_deconvolve(psi_t, h, -1)
    setup_s, loop_s = orc.last_timing()
    d = got.astype(np.float64) - ref.astype(np.float64)
    out.update({
        "case": "view 0 of the timed problem, 1 iteration from the timed psi: inplace_gpu_deconvolve (host "
                "buffers, cyclic policy) vs CPU oracle",
        "oracle_checksum": float(ref.astype(np.float64).sum()),
        "max_rel": float(np.abs(d).max() / np.abs(ref).max()),
        "rms_rel": float(np.sqrt((d * d).mean()) / np.sqrt((ref.astype(np.float64) ** 2).mean())),
        "tolerance": {"max_rel": 1e-4, "rms_rel": 1e-5},
    })
    out["ok"] = bool(out["max_rel"] <= 1e-4 and out["rms_rel"] <= 1e-5)
    cpu = {
        "value": 1.0 / (loop_s * n_views),
        "unit": "RL iterations/s (%d views)" % n_views,
        "cores": cores,
        "kind": orc.fft_backend(),  # "port" (the oracle's own FFT) or "fftw" (dlopen'ed libfftw3f, the reference's)
        "sample": "%dx%dx%d f32, view 0 of %d, 1 iteration from the timed psi (loop %.2f s, PSF setup %.2f s "
                  "excluded), scaled x1/%d" % (shape[0], shape[1], shape[2], n_views, loop_s, setup_s, n_views),
    }
    if cpu["kind"] == "port":
        cpu["note"] = "libfftw3f.so.3 not found on this host: the port's own mixed-radix FFT is not FFTW"
    return out, cpu


def abi_end_to_end(lib, shape, psf_edge, n_views, iterations, device, pad_mode):
    """inplace_gpu_deconvolve with host buffers, as Fiji calls it and as the reference's
    bench_gpu_deconvolve_synthetic times it: uploads, PSF preparation, loop, download."""
    import numpy as np
    from libmultiviewnative_amd.abi import WorkspaceHolder
    views, k1s, k2s = [], [], []
    for v in range(n_views):
        view, k1, k2 = make_view(shape, v, psf_edge)
        views.append(view)
        k1s.append(k1)
        k2s.append(k2)
    w = np.full(shape, 1.0 / n_views, np.float32)
    h = WorkspaceHolder(views, k1s, k2s, [w] * n_views, LAMBDA, MIN_VALUE, iterations)
    lib.set_pad_mode(pad_mode)
    secs = []
    ok = True
    for _ in range(3):
        out = np.full(shape, np.float32(start_value()), np.float32)
        secs.append(lib.gpu_deconvolve_inplace(out, h, device))  # the C call alone, psi updated in place
        ok = ok and bool(np.isfinite(out).all() and float(out[0, 0, 0]) != start_value())
    lib.set_pad_mode(None)
    lib.check(lib.l.mvn_release_cached_engines())
    warm = min(secs[1:])
    return {"pad_mode": pad_mode, "iterations": iterations, "seconds_first_call": round(secs[0], 4),
            "seconds_warm_call": round(warm, 4), "iterations_per_s_warm": round(iterations / warm, 3),
            "result_finite_and_changed": ok}


def exact_halo_mode(lib, devices, shape, psf_edge, n_views, steps, warmup, check_device):
    """The same problem as ONE volume cut into dim0 slabs over `devices` of THIS process (mvn_group_*: what
    MVN_DEVICES runs inside inplace_gpu_deconvolve; libmultiviewnative_amd/csrc/mvn_multi.cpp): the reference's
    sequential sweep, halos pulled from the neighbours before every dim0 leg.  Stacks resident; timed like the
    headline (wall time of `steps` sweeps between two all-slab syncs).  Parity: the result must equal the
    one-device engine's sequential sweep BIT FOR BIT (that sweep is what the parity tests hold against the oracle)."""
    import numpy as np
    from libmultiviewnative_amd.abi import WorkspaceHolder
    views, k1s, k2s = [], [], []
    for v in range(n_views):
        view, k1, k2 = make_view(shape, v, psf_edge)
        views.append(view)
        k1s.append(k1)
        k2s.append(k2)
    w = np.full(shape, 1.0 / n_views, np.float32)
    h = WorkspaceHolder(views, k1s, k2s, [w] * n_views, LAMBDA, MIN_VALUE, steps)
    psi0 = np.full(shape, np.float32(start_value()), np.float32)
    g = lib.group(devices, shape, max(1, min(psf_edge, shape[0]) // 2), n_views)
    try:
        g.load(psi0, h)
        if warmup > 0:
            g.iterate(warmup, LAMBDA, MIN_VALUE)
        ms = g.iterate(steps, LAMBDA, MIN_VALUE)
        got = g.get_psi()
    finally:
        g.close()
    out = {"update_mode": "sequential (reference order, Gauss-Seidel) on dim0 slabs: halo exchange of 2 x %d planes per "
                          "convolution and neighbour pair, peer copies under the interior planes' dim0 leg"
                          % max(1, min(psf_edge, shape[0]) // 2),
           "devices": list(devices), "value": round(steps / (ms * 1e-3), 4), "unit": "iterations/s",
           "ms_per_step": round(ms / steps, 4), "steps": steps, "warmup": warmup,
           "psi_finite_positive": bool(np.isfinite(got).all() and (got > 0).all())}
    # (the slabs run the layout and the middle - fused on planes of 512 x 512 - the whole volume has on one device: the
    # same arithmetic, bit for bit)
    e = lib.engine(shape, n_views, device=check_device)
    try:
        for v in range(n_views):
            e.set_view(v, views[v], w, k1s[v], k2s[v])
        e.set_psi(psi0)
        # (in the same two calls as the group: the last view update of a call ends in the plain c2r pass, every
        # other one in the fused c2r + r2c pass - the same values up to the rounding of two different kernels)
        if warmup > 0:
            e.iterate(warmup, LAMBDA, MIN_VALUE, sync=True)
        e.iterate(steps, LAMBDA, MIN_VALUE, sync=True)
        one = e.get_psi()
    finally:
        e.close()
    out["parity"] = {"case": "psi after %d sweeps vs the one-device engine's sequential sweep" % (warmup + steps),
                     "bit_equal": bool(np.array_equal(got, one)),
                     "max_rel": float(np.abs(got.astype(np.float64) - one).max() / np.abs(one).max())}
    return out


def measured_traffic(kind, shape):
    """HBM bytes per launch of the dominant kernel from the PMC counters, collected NOW: two rocprofv3 passes
    (FETCH_SIZE; WRITE_SIZE) of child processes that run tools/pmc_probe.py on this GPU, corrected as
    MI355X_MICROARCH.md's HBM section prescribes (counters in KiB, read side x 2 on gfx950) by
    tools/pmc_summarize.py.  None when rocprofv3 is not there or a pass fails - the static record is used then."""
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None or tuple(shape) != (512, 512, 512):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summarize
    tmp = tempfile.mkdtemp(prefix="mvn_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for sub, counters in (("rd", ["FETCH_SIZE"]), ("wr", ["WRITE_SIZE", "GRBM_GUI_ACTIVE"])):
            cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + [
                "-d", os.path.join(tmp, sub), "-o", "p", "--output-format", "csv", "--",
                sys.executable, os.path.join(ROOT, "tools", "pmc_probe.py")]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150)
            if r.returncode != 0:
                return None
        rd, rn = pmc_summarize.read(os.path.join(tmp, "rd"), "FETCH_SIZE")
        wr, wn = pmc_summarize.read(os.path.join(tmp, "wr"), "WRITE_SIZE")
        for name in rd:
            short = name.replace("void ", "").split("(")[0]
            if kind in pmc_summarize.KINDS.get(short, []) and rn[name] and wn.get(name):
                return (2.0 * rd[name] / rn[name] + wr[name] / wn[name]) * 1024.0
        return None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[i] (see the module docstring); default 2, the headline")
    ap.add_argument("--size", type=int, nargs=3, default=None, help="override the volume (test rehearsals)")
    ap.add_argument("--views", type=int, default=None, help="override the TOTAL number of views")
    ap.add_argument("--psf", type=int, default=None, help="override the PSF edge")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel events in the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1,
                    help="rehearsal on a one-GPU box: put every rank on this device (needs --backend gloo)")
    ap.add_argument("--simultaneous", action="store_true",
                    help="run the simultaneous-update (Jacobi) loop even on one rank (what the ranks of an N>1 run compute)")
    ap.add_argument("--chunks", type=int, default=4, help="dim0 chunks of the overlapped all-reduce (N > 1)")
    ap.add_argument("--host-sync", action="store_true",
                    help="N > 1: order collectives against the engine with host synchronisation instead of stream events")
    ap.add_argument("--no-side", action="store_true",
                    help="skip ms_per_fft, the one-view parity check and the ABI end-to-end call "
                         "(keeps rocprof --stats averages clean)")
    ap.add_argument("--no-abi", action="store_true", help="skip the ABI end-to-end call")
    ap.add_argument("--force-dist", action="store_true",
                    help="one rank that still goes through torch.distributed and the chunked all-reduce "
                         "(rehearses the N > 1 code path, stream ordering included, on a one-GPU box)")
    ap.add_argument("--check-parity", action="store_true",
                    help="N > 1: rank 0 still runs the one-view parity check / CPU sample (default: N = 1 only)")
    ap.add_argument("--dump-psi", default=None, help="rank 0 saves the final psi here (.npy)")
    ap.add_argument("--exact-halo-child", default=None, metavar="DEVICES",
                    help="internal: run only the exact_halo_mode measurement on these devices (comma separated) and "
                         "print its JSON block (the N > 1 lines start it as a child process with a time limit)")
    args = ap.parse_args()
    if args.exact_halo_child:
        cfg = CONFIGS[args.config]
        shape = tuple(args.size or cfg["size"])
        V = args.views if args.views is not None else cfg["views"]
        with stdout_to_stderr():
            from libmultiviewnative_amd import native
            devices = [int(d) for d in args.exact_halo_child.split(",")]
            block = exact_halo_mode(native.lib(), devices, shape, args.psf or cfg["psf"], V, args.steps, args.warmup,
                                    devices[0])
        print(json.dumps(block), flush=True)
        return

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(self_launch(args))  # the parent never touches the GPU
    with stdout_to_stderr():  # nothing but the final JSON line may reach stdout
        out = run_rank(args, world_env)
    if out is not None:
        print(json.dumps(out), flush=True)


def run_rank(args, world_env):

    cfg = CONFIGS[args.config]
    shape = tuple(args.size or cfg["size"])
    V = args.views if args.views is not None else cfg["views"]
    psf_edge = args.psf or cfg["psf"]
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    args.gpus = world
    if args.force_dist and world_env is None:
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(free_port()))
    use_dist = world > 1 or args.force_dist

    dist = None
    torch = None
    if use_dist:
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
    from libmultiviewnative_amd.sharded import SimultaneousDriver, view_partition

    lib = native.lib()  # raises if the HIP library is missing: there is no fallback
    mine = view_partition(V, world, rank)
    eng = lib.engine(shape, len(mine), device=local_rank)
    weights = np.full(shape, 1.0 / V, np.float32)
    for i, v in enumerate(mine):
        view, k1, k2 = make_view(shape, v, psf_edge)
        eng.set_view(i, view, weights, k1, k2)
        del view
    del weights
    eng.set_psi(np.full(shape, np.float32(start_value()), np.float32))

    driver = None
    delta = None
    if use_dist:
        nfl = eng.psi_ptr()[1]
        dev = torch.device("cuda", local_rank)
        delta = torch.zeros(nfl, dtype=torch.float32, device=dev)
        eng.bind_delta(delta.data_ptr())
        ext = None if args.host_sync else torch.cuda.ExternalStream(eng.stream(), device=dev)
        driver = SimultaneousDriver(eng, delta, dist, chunks=args.chunks, stream=ext, force_collective=args.force_dist)
    elif args.simultaneous:
        driver = SimultaneousDriver(eng, None, None)

    def run(steps):
        if driver is None:
            eng.iterate(steps, LAMBDA, MIN_VALUE, sync=True)
        else:
            driver.run(steps, LAMBDA, MIN_VALUE)

    def fence():
        eng.sync()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    if use_dist:
        # communicator set-up (lazy in RCCL) must never land in the timed region, even with --warmup 0
        dist.all_reduce(delta, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        delta.zero_()
        torch.cuda.synchronize()
    if args.warmup > 0:
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

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda:%d" % local_rank)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    psi = None
    if rank == 0:
        psi = eng.get_psi()
        psi_ok = bool(np.isfinite(psi).all() and (psi > 0).all())
        if args.dump_psi:
            np.save(args.dump_psi, psi)
        ms_per_step = elapsed / args.steps * 1e3
        value = args.steps / elapsed  # iterations/s of the WHOLE problem, at every N
        d0, d1, d2 = shape
        vol = 4.0 * d0 * d1 * d2
        # dominant kernel by total time inside the timed region
        roofline = None
        KB = lambda k: kernel_bytes(k, d0, d1, d2, min(psf_edge, d0))
        prof = {k: v for k, v in prof.items() if v[1] and KB(k)[0] > 0}
        if prof:
            # the kernel with the largest total time; two kernels within 5 % of each other (the direct dim0 leg and the
            # fused update pass are, at 512^3) would swap places from run to run: the one FURTHER from its roofline
            # is reported then, the other one is named in `runner_up`
            top = max(v[0] for v in prof.values())
            close = [k for k in prof if prof[k][0] >= 0.95 * top]
            kind = min(close, key=lambda k: KB(k)[0] / (prof[k][0] / max(prof[k][1], 1)))
            tot_ms, n = prof[kind]
            kb, B = KB(kind)
            avg_ms = tot_ms / max(n, 1)
            achieved = kb / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": kind, "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": None, "avg_launch_ms": round(avg_ms, 4), "launches": n,
                        "bytes_per_launch": kb,
                        "runner_up": [k for k in close if k != kind],
                        "per_kernel": {k: {"avg_ms": round(v[0] / v[1], 4), "launches": v[1],
                                           "GBps": round(KB(k)[0] / (v[0] / v[1] * 1e-3) / 1e9, 1)}
                                       for k, v in prof.items()}}
            if kind == "mid_fused":
                # the fused middle pass is priced against HBM like every pass (the contract's bound), but it is NOT
                # memory-bound: it does the arithmetic of three passes on one read and one write of the volume
                K = min(psf_edge, d0)
                flops = d0 * (d2 // 2) * (d1 * 8.0 * K + 2 * 5.0 * d1 * np.log2(d1))
                roofline["note"] = ("three passes' arithmetic on one read + one write of the half-spectrum: bound by "
                                    "vector instruction issue and LDS latency at two waves per SIMD (the filter's 126 "
                                    "registers per bin), not by HBM - DESIGN.md section 4, profiles/r04_mid_fused.md")
                roofline["vector_f32"] = {"flop_per_launch": flops, "achieved_TFLOPs": round(flops / (avg_ms * 1e-3) / 1e12, 1),
                                          "peak_TFLOPs_packed_f32": 157.3,
                                          "frac": round(flops / (avg_ms * 1e-3) / 1e12 / 157.3, 4)}
            traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(traffic_file) and shape == (512, 512, 512):  # PMC passes were taken at 512^3
                try:
                    roofline["traffic"] = json.load(open(traffic_file)).get(kind)
                    roofline["traffic_source"] = ("static: profiles/pmc_traffic.json (PMC passes of tools/profile_round.sh "
                                                  "over tools/pmc_probe.py, builder-run)")
                except Exception:
                    pass
        _, B = kernel_bytes("nyquist", d0, d1, d2)
        whole = 25.0 * B * V / (ms_per_step * 1e-3) / 1e9  # 25*B per (view, iteration)
        parts = [len(view_partition(V, world, r)) for r in range(world)]
        if driver is None:
            mode = "sequential (reference order, Gauss-Seidel)"
        elif not use_dist:
            mode = "simultaneous (Jacobi), single rank"
        else:
            mode = "simultaneous (Jacobi): views sharded over ranks + 1 all-reduce of the psi-sized delta per " \
                   "iteration in %d overlapped chunks (backend %s, %s ordering)" % (
                       driver.n, args.backend, "host-sync" if args.host_sync else "stream-event")
        out = {
            "metric": "RL iterations/sec on %s f32, %d views" % (
                "512^3" if shape == (512, 512, 512) else "%dx%dx%d" % shape, V),
            "value": round(value, 4),
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: %dx%dx%d f32, %d views, %d^3 PSFs, lambda=%g, "
                                   "resident in HBM" % (args.config, d0, d1, d2, V, psf_edge, LAMBDA),
                       "views_total": V, "views_per_rank": parts,
                       "update_mode": mode,
                       "unit_of_value": "updates of psi from all %d views per second (one problem, all ranks)" % V},
            "whole_step_GBps_25B_model": round(whole, 1),
            "whole_step_frac_of_8TBps": round(whole / HBM_PEAK_GBS, 4),
            "psi_finite_positive": psi_ok,
            "roofline": roofline,
        }
        if driver is None:
            # what the 8-pass pipeline actually moves per (view, iteration), DESIGN.md section 4: 22 volumes
            # with the fused FFT dim0 pass, 20 + 2 K / d0 with the direct dim0 leg (K PSF planes)
            direct = "axis0_direct" in prof
            fused_mid = "mid_fused" in prof
            vols = 20.0 + 2.0 * min(psf_edge, d0) / d0 if direct else 22.0
            if fused_mid:  # per convolution: last-axis pass out, ONE middle pass (2 volumes + taps), last-axis pass in
                vols = 12.0 + 2.0 * min(psf_edge, d0) / d0
            actual = vols * vol * V / (ms_per_step * 1e-3) / 1e9
            out["whole_step_volumes_moved_per_view_iteration"] = round(vols, 3)
            out["whole_step_GBps_actual"] = round(actual, 1)
            out["whole_step_actual_frac_of_8TBps"] = round(actual / HBM_PEAK_GBS, 4)
            out["dim0_leg"] = "direct (%d PSF planes)" % min(psf_edge, d0) if (direct or fused_mid) else "fused FFT pass"
            out["middle_passes"] = ("one (dim1 forward + direct dim0 leg + dim1 inverse fused, line layout)" if fused_mid
                                    else "three")
        # The N > 1 lines time the simultaneous (Jacobi) loop, this line's `value` the reference-order
        # sequential sweep: a scaling series must be read against the SAME loop on one rank, so the
        # N = 1 line carries that rate too (same engine, same stacks, same step count).
        if world == 1 and driver is None and not args.no_side:
            jac = SimultaneousDriver(eng, None, None)
            jac.run(max(1, args.warmup), LAMBDA, MIN_VALUE)
            eng.sync()
            t1 = time.perf_counter()
            jac.run(args.steps, LAMBDA, MIN_VALUE)
            eng.sync()
            dt = time.perf_counter() - t1
            out["scaling_baseline"] = {
                "update_mode": "simultaneous (Jacobi), single rank: the loop the N > 1 lines run, on one GPU",
                "value": round(args.steps / dt, 4), "unit": "iterations/s",
                "ms_per_step": round(dt / args.steps * 1e3, 4),
                "note": "speed-up of an N > 1 line = its value / this value; `value` above is the sequential "
                        "(Gauss-Seidel, reference-order) sweep and stays the headline"}
        elif use_dist:
            out["scaling_relative_to"] = ("the N = 1 line's scaling_baseline.value (the same Jacobi loop on one "
                                          "rank), not its headline value (sequential sweep)")
    if use_dist:
        eng.bind_delta(None)
    eng.close()

    if rank == 0 and not args.no_side and (world == 1 or args.check_parity):
        try:
            if world == 1 and out.get("roofline") and not args.no_abi:
                live = measured_traffic(out["roofline"]["kernel"], shape)  # (the engine is closed: the GPU is free)
                if live:
                    out["roofline"]["traffic"] = live
                    out["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                                         "over tools/pmc_probe.py on this GPU (read side x 2 on gfx950)")
            if world == 1:
                out["ms_per_fft"] = round(lib.fft3_time(shape, 0, 10, device=local_rank), 4)
            parity, cpu = one_view_check(lib, shape, psf_edge, V, psi, local_rank, not args.no_cpu_baseline)
            out["parity"] = parity
            if cpu:
                out["cpu_baseline"] = cpu
            if world == 1 and not args.no_abi and args.config != 4:
                its = 10
                out["abi_end_to_end"] = {
                    "what": "one inplace_gpu_deconvolve call, %dx%dx%d x %d views x %d iterations, host buffers "
                            "in and out (PCIe included)" % (shape[0], shape[1], shape[2], V, its),
                    "timed": "the C call alone (psi updated in place), best of two warm calls",
                    "cyclic_policy": abi_end_to_end(lib, shape, psf_edge, V, its, local_rank, "none"),
                }
                if args.config == 2:
                    # the library's default: the reference GPU entry's zero_padd on FFT-friendly
                    # padded extents (512 + 31 - 1 = 542 -> 560), what a Fiji block gets
                    out["abi_end_to_end"]["reference_gpu_policy"] = abi_end_to_end(
                        lib, shape, psf_edge, V, its, local_rank, "zero")
                    # the same call cut into two dim0 slabs (MVN_DEVICES; both on this device: the rehearsal a
                    # one-GPU box allows - on a node the entries name different GPUs)
                    os.environ["MVN_DEVICES"] = "%d,%d" % (local_rank, local_rank)
                    try:
                        out["abi_end_to_end"]["cyclic_policy_MVN_DEVICES_two_slabs_one_gpu"] = abi_end_to_end(
                            lib, shape, psf_edge, V, its, local_rank, "none")
                    finally:
                        os.environ.pop("MVN_DEVICES", None)
            if world == 1 and not args.no_abi:
                # ONE slab that is its own neighbour (cyclic self-exchange): what the slab mode costs on a device
                # before any link is involved - own planes + 2 x h halo planes, the exchange, the split leg
                out["exact_halo_mode"] = exact_halo_mode(lib, [local_rank], shape, psf_edge, V, args.steps,
                                                         args.warmup, local_rank)
                # two slabs sharing the one device: a rehearsal of the concurrency (threads, events, copies under
                # compute) - its TIME is that of two full-device launch chains competing for one GPU
                out["exact_halo_mode"]["two_slabs_one_gpu_rehearsal"] = exact_halo_mode(
                    lib, [local_rank, local_rank], shape, psf_edge, V, args.steps, args.warmup, local_rank)
        except Exception as e:  # the headline number must survive a failing side measurement
            out["side_measurement_error"] = "%s: %s" % (type(e).__name__, e)
    # (MVN_BENCH_EXACT_DEVICES=0,0: rehearsal of this step on a one-GPU box, where every rank sits on one device)
    exact_devices = os.environ.get("MVN_BENCH_EXACT_DEVICES") or (
        ",".join(str(d) for d in range(world)) if args.all_ranks_on_device < 0 else "")
    if use_dist and world > 1 and not args.no_side and exact_devices:
        # The N > 1 lines' `value` is the north-star mode (views sharded, one all-reduce per iteration: Jacobi).
        # The mode that keeps the REFERENCE's update order on N GPUs - dim0 slabs with a halo exchange, driven from
        # ONE process (rank 0) - rides along.  The other ranks wait on the HOST (a key in the rendezvous store, not
        # a collective: a device-side barrier would spin on the very GPUs being measured).
        import datetime
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            # in a CHILD process with a time limit: this path has never run on more than one device (the builder's
            # pool leases one GPU), and neither an error nor a hang in it may cost the line its headline
            try:
                cmd = [sys.executable, os.path.abspath(__file__), "--exact-halo-child",
                       exact_devices, "--config", str(args.config), "--steps", str(args.steps),
                       "--warmup", str(args.warmup)]
                if args.size:
                    cmd += ["--size"] + [str(x) for x in args.size]
                if args.views is not None:
                    cmd += ["--views", str(args.views)]
                if args.psf:
                    cmd += ["--psf", str(args.psf)]
                env = {k: v for k, v in os.environ.items()
                       if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MVN_DEVICES")}
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode == 0 and lines:
                    out["exact_halo_mode"] = json.loads(lines[-1])
                else:
                    out["exact_halo_mode"] = {"error": "child rc %d: %s" % (r.returncode, r.stderr[-600:])}
            except Exception as e:  # the headline number must survive
                out["exact_halo_mode"] = {"error": "%s: %s" % (type(e).__name__, e)}
            store.set("mvn_exact_halo_done", "1")
        else:
            store.wait(["mvn_exact_halo_done"], datetime.timedelta(minutes=30))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
