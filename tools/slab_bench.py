"""Slab-decomposed SEQUENTIAL sweep on N GPUs: strong scaling of one
V-view stack, exact single-GPU arithmetic, four all-to-all exchanges per (view, iteration).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/slab_bench.py [--size 512 512 512] [--views 6] [--steps 5] [--check]

`--backend nccl` (default) exchanges the bound device buffers in place over RCCL.
`--backend gloo --all-ranks-on-device 0` is the one-GPU rehearsal: every rank on the same card,
exchange staged through host tensors (gloo has no device all-to-all) -- it validates the
multi-rank device path, its timings mean nothing.
`--check` also runs the resident single-GPU engine on rank 0 and compares (small sizes).
Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gaussian_psf(edge, sigma):
    import numpy as np
    ax = [np.arange(edge) - edge // 2 for _ in range(3)]
    g = [np.exp(-0.5 * (a / s) ** 2) for a, s in zip(ax, sigma)]
    k = g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]
    return (k / k.sum()).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--views", type=int, default=6)
    ap.add_argument("--psf", type=int, default=31)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch  # before the product library: both then share torch's HIP runtime
    import torch.distributed as dist
    if args.all_ranks_on_device >= 0:
        local_rank = args.all_ranks_on_device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(args.backend)

    import numpy as np
    from libmultiviewnative_amd import native
    from libmultiviewnative_amd.sharded import SlabDriver

    lib = native.lib()
    shape = tuple(args.size)
    V = args.views
    lam, minv = 0.006, 1e-4
    z0, z1 = rank * shape[0] // world, (rank + 1) * shape[0] // world
    eng = lib.slab_engine(shape, world, rank, V, device=local_rank)
    slab = (z1 - z0,) + shape[1:]
    weights = np.full(slab, 1.0 / V, np.float32)
    psfs = []
    for v in range(V):
        # every rank draws the whole view's random stream lazily, plane by plane, so that the
        # stack does not depend on the rank count
        rng = np.random.default_rng(5000 + v)
        planes = []
        for z in range(shape[0]):
            p = rng.random(shape[1:], dtype=np.float32) * 50 + 10
            if z0 <= z < z1:
                planes.append(p)
        sig = [2.0, 2.0, 2.0]
        sig[v % 3] = 4.0
        psf = gaussian_psf(args.psf, sig)
        psfs.append(psf)
        eng.set_view(v, np.stack(planes), weights, psf, np.ascontiguousarray(psf[::-1, ::-1, ::-1]))
    eng.set_psi(np.full(slab, np.float32(35.0), np.float32))

    nm, nn = eng.buffer_sizes()
    a_main = torch.zeros(nm, dtype=torch.float32, device=dev)
    b_main = torch.zeros(nm, dtype=torch.float32, device=dev)
    a_nyq = torch.zeros(nn, dtype=torch.float32, device=dev) if nn else None
    b_nyq = torch.zeros(nn, dtype=torch.float32, device=dev) if nn else None
    eng.bind_buffers(a_main.data_ptr(), b_main.data_ptr(), a_nyq.data_ptr() if nn else None,
                     b_nyq.data_ptr() if nn else None)

    if args.backend == "nccl":
        driver = SlabDriver(eng, a_main, b_main, a_nyq, b_nyq, dist,
                            after_collective=lambda: torch.cuda.current_stream().synchronize())
    else:
        class HostStaged(SlabDriver):
            def _exchange(self, src_main, dst_main, src_nyq, dst_nyq):
                self.engine.sync()
                for src, dst in ((src_main, dst_main), (src_nyq, dst_nyq)):
                    if src is None:
                        continue
                    h_src = src.cpu()
                    h_dst = torch.empty_like(h_src)
                    dist.all_to_all_single(h_dst, h_src)
                    dst.copy_(h_dst)
                torch.cuda.synchronize()
        driver = HostStaged(eng, a_main, b_main, a_nyq, b_nyq, dist)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    # communicator set-up outside the timed region
    if args.backend == "nccl":
        dist.all_to_all_single(b_main, a_main)
        torch.cuda.synchronize()
    driver.run(args.warmup, V, lam, minv)
    fence()
    t0 = time.perf_counter()
    driver.run(args.steps, V, lam, minv)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    psi = eng.get_psi()
    ok = bool(np.isfinite(psi).all() and (psi > 0).all())
    check = None
    if args.check:
        gathered = [None] * world
        dist.all_gather_object(gathered, psi)
        if rank == 0:
            full = np.concatenate(gathered, axis=0)
            one = lib.engine(shape, V, device=local_rank)
            wfull = np.full(shape, 1.0 / V, np.float32)
            for v in range(V):
                rng = np.random.default_rng(5000 + v)
                view = np.stack([rng.random(shape[1:], dtype=np.float32) * 50 + 10 for _ in range(shape[0])])
                one.set_view(v, view, wfull, psfs[v], np.ascontiguousarray(psfs[v][::-1, ::-1, ::-1]))
            one.set_psi(np.full(shape, np.float32(35.0), np.float32))
            one.iterate(args.warmup + args.steps, lam, minv, sync=True)
            ref = one.get_psi()
            one.close()
            check = {"max_rel_vs_single_gpu_engine": float(np.abs(full - ref).max() / np.abs(ref).max())}
    if rank == 0:
        B = 4 * shape[0] * shape[1] * 2 * (shape[2] // 2 + 1)
        print(json.dumps({
            "metric": "RL iterations/sec, slab-decomposed sequential sweep",
            "value": round(args.steps / elapsed, 4), "unit": "iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%dx%d f32, %d views total, %d^3 PSFs, lambda=0.006" % (shape + (V, args.psf)),
                       "update_mode": "sequential (reference order), planes split over ranks, "
                                      "4 all-to-all per (view, iteration), backend %s" % args.backend},
            "exchange_bytes_per_rank_per_iteration": int(4 * V * (world - 1) / world * B / world),
            "psi_finite_positive": ok, "check": check}), flush=True)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
