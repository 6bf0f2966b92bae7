"""The halo-exchange slab mode (libmultiviewnative_amd/sharded.py: HaloSlabDriver) on N ranks: ONE volume cut into
dim0 slabs, swept in the reference's view order.  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/halo_bench.py --steps 10 --warmup 2            # 512^3 x 6 views x 31^3 PSFs over RCCL
    python tools/halo_bench.py --ranks 2 --backend gloo --all-ranks-on-device 0 --size 64 64 64 ...   # rehearsal

Rank 0 prints one JSON line: iterations/s of the whole problem (max over ranks of the timed region), and can save the
assembled psi (--dump-psi) for a comparison with the sequential oracle.  Un-measured on several GPUs so far: the
builder's box has one; the exchange is synchronous (no overlap with the interior planes yet).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--views", type=int, default=6)
    ap.add_argument("--psf", type=int, default=31)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1)
    ap.add_argument("--ranks", type=int, default=0, help="spawn this many ranks from here (no torchrun)")
    ap.add_argument("--dump-psi", default=None)
    ap.add_argument("--force-dist", action="store_true",
                    help="one rank that still exchanges through torch.distributed with itself (isend / irecv of device "
                         "tensors over RCCL, poison word all-reduced): rehearses the multi-rank path on a one-GPU box")
    args = ap.parse_args()

    if args.ranks > 1 and "RANK" not in os.environ:
        import bench
        port = bench.free_port()
        procs = []
        for r in range(args.ranks):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.ranks), LOCAL_RANK=str(r),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable] + sys.argv, env=env))
        rc = [p.wait() for p in procs]
        sys.exit(max(rc))

    os.environ.setdefault("MVN_DIM0_DIRECT_MIN_ITEMS", "0")  # the mode is built on the direct dim0 leg
    os.environ["MVN_DIM0_DIRECT_MIN_ITEMS"] = "0"
    if args.force_dist and "RANK" not in os.environ:
        import bench
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(bench.free_port()))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.all_ranks_on_device >= 0:
        local_rank = args.all_ranks_on_device
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    import numpy as np
    import bench
    from libmultiviewnative_amd import native
    from libmultiviewnative_amd.sharded import HaloSlabDriver

    lib = native.lib()  # raises if the HIP library is missing: there is no fallback
    shape, V = tuple(args.size), args.views
    tdev = torch.device("cuda", local_rank) if (world == 1 or args.backend == "nccl") else torch.device("cpu")
    drv = HaloSlabDriver(lib, shape, V, min(args.psf, shape[0]), dist=dist if (world > 1 or args.force_dist) else None,
                         rank=rank, world=world, device=local_rank, torch_device=tdev, force_collective=args.force_dist)
    sl = slice(drv.z0, drv.z0 + drv.nz)
    weights = np.full((drv.nz,) + shape[1:], 1.0 / V, np.float32)
    for v in range(V):
        view, k1, k2 = bench.make_view(shape, v, args.psf)
        drv.set_view(v, np.ascontiguousarray(view[sl]), weights, k1, k2)
        del view
    drv.set_psi(np.full((drv.nz,) + shape[1:], np.float32(bench.start_value()), np.float32))

    def fence():
        drv.eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        drv.run(args.warmup, bench.LAMBDA, bench.MIN_VALUE)
    fence()
    t0 = time.perf_counter()
    drv.run(args.steps, bench.LAMBDA, bench.MIN_VALUE)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mine = torch.from_numpy(drv.get_psi()).to(tdev)
    parts = [torch.empty_like(mine) for _ in range(world)] if world > 1 else [mine]
    if world > 1:
        dist.all_gather(parts, mine)
    if rank == 0:
        psi = np.concatenate([p.cpu().numpy() for p in parts], axis=0)
        if args.dump_psi:
            np.save(args.dump_psi, psi)
        print(json.dumps({
            "metric": "RL iterations/sec on %dx%dx%d f32, %d views" % (shape + (V,)),
            "value": round((args.steps) / elapsed, 4), "unit": "iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%dx%d f32, %d views, %d^3 PSFs, resident in HBM" % (shape + (V, args.psf)),
                       "update_mode": "sequential (reference order) on dim0 slabs of %d planes + %d halo planes, "
                                      "halo exchange before every dim0 leg (backend %s)" % (drv.nz, drv.h, args.backend)},
            "psi_finite_positive": bool(np.isfinite(psi).all() and (psi > 0).all())}), flush=True)
    drv.close()
    if world > 1 or args.force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
