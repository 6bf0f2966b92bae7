"""world_size-2 gloo runs of the multi-GPU drivers on CPU: the product's partition + loop
(libmultiviewnative_amd/sharded.py) around the PRODUCT engine on its host-emulation backend must
reproduce the oracle's single-process simultaneous mode (view sharding) and sequential sweep
(slab decomposition).  On the GPU the same drivers wrap the HIP engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from libmultiviewnative_amd.abi import WorkspaceHolder
from libmultiviewnative_amd.sharded import SimultaneousDriver, view_partition

HERE = os.path.dirname(os.path.abspath(__file__))


def test_view_partition():
    assert view_partition(6, 1, 0) == [0, 1, 2, 3, 4, 5]
    assert [view_partition(6, 4, r) for r in range(4)] == [[0, 1], [2, 3], [4], [5]]
    assert [view_partition(8, 8, r) for r in range(8)] == [[r] for r in range(8)]
    assert [view_partition(2, 4, r) for r in range(4)] == [[0], [1], [], []]
    allv = sum((view_partition(13, 5, r) for r in range(5)), [])
    assert allv == list(range(13))
    with pytest.raises(ValueError):
        view_partition(4, 2, 2)


# (shape, views, world): 3 views split 2 + 1 (generic kernels, odd d2 = padded rows), a
# fixed-kernel shape (whole-tile chunks), and 1 view on 2 ranks (rank 1 has none and contributes zeros)
SIM_CASES = {"split21": ((12, 10, 14), 3, 2), "odd": ((8, 6, 9), 3, 2), "fixed": ((64, 64, 32), 2, 2),
             "idle_rank": ((12, 10, 14), 1, 2)}
SIM_ITS = 3


def _worker(rank, world, port, out_dir, case):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from libmultiviewnative_amd import native
    from ref_fixtures import realistic_views
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape, V, _ = SIM_CASES[case]
    _, views, k1, k2, w, psi0 = realistic_views(shape, V, (3, 3, 3))
    mine = view_partition(V, world, rank)
    # the PRODUCT engine (compute_delta_head / _chunk / apply_delta_chunk, bind_delta) on the
    # host-emulation backend, driven by the product's SimultaneousDriver
    emu = native.Binding(native.EMU_SO)
    eng = emu.engine(shape, len(mine))
    for i, v in enumerate(mine):
        eng.set_view(i, views[v], w[v], k1[v], k2[v])
    eng.set_psi(psi0)
    delta = torch.zeros(eng.psi_ptr()[1], dtype=torch.float32)
    eng.bind_delta(delta.data_ptr())
    drv = SimultaneousDriver(eng, delta, dist, chunks=3)
    assert drv.n >= 2
    drv.run(SIM_ITS, 0.006, 1e-4)
    np.save(os.path.join(out_dir, "psi_rank%d.npy" % rank), eng.get_psi())
    eng.bind_delta(None)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", sorted(SIM_CASES))
def test_two_rank_gloo_matches_single_process(tmp_path, case):
    from oracle import binding as orc
    from ref_fixtures import realistic_views
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    shape, V, world = SIM_CASES[case]
    mp.spawn(_worker, args=(world, port, str(tmp_path), case), nprocs=world, join=True)
    a = np.load(tmp_path / "psi_rank0.npy")
    b = np.load(tmp_path / "psi_rank1.npy")
    assert np.array_equal(a, b)  # replicas stay identical
    _, views, k1, k2, w, psi0 = realistic_views(shape, V, (3, 3, 3))
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, SIM_ITS)
    ref = orc.cpu_deconvolve_simultaneous(psi0, h, 1)
    assert np.abs(a - ref).max() <= 1e-5 * np.abs(ref).max()


def test_chunked_step_equals_whole_step():
    """compute_delta_head + chunks + apply_delta_chunk(feed_next) must give bit for bit what
    compute_delta + apply_delta give (same kernels on row / plane ranges)."""
    from libmultiviewnative_amd import native
    from ref_fixtures import realistic_views
    emu = native.Binding(native.EMU_SO)
    for shape in ((12, 10, 14), (64, 32, 32), (6, 5, 9)):
        _, views, k1, k2, w, psi0 = realistic_views(shape, 2, (3, 3, 3), seed=5)
        outs = []
        for chunked in (False, True):
            e = emu.engine(shape, 2)
            for v in range(2):
                e.set_view(v, views[v], w[v], k1[v], k2[v])
            e.set_psi(psi0)
            if chunked:
                n = e.delta_chunks(4)
                assert n >= 2
                for it in range(3):
                    e.compute_delta_head(0.006, 1e-4)
                    for c in range(n):
                        e.compute_delta_chunk(c, n)
                    for c in reversed(range(n)):  # any order within a round
                        e.apply_delta_chunk(c, n, it < 2)
            else:
                for it in range(3):
                    e.compute_delta(0.006, 1e-4)
                    e.apply_delta()
            e.sync()
            outs.append(e.get_psi())
            e.close()
        assert np.array_equal(outs[0], outs[1])


# ---- slab-decomposed sequential sweep -------------------------------------
SLAB_CASES = {"even": ((8, 12, 16), 3), "odd": ((6, 10, 9), 2), "pow2": ((64, 64, 32), 2)}


def _slab_worker(rank, world, port, out_dir, case):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from libmultiviewnative_amd import native
    from libmultiviewnative_amd.sharded import SlabDriver
    from ref_fixtures import realistic_views
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape, V = SLAB_CASES[case]
    _, views, k1, k2, w, psi0 = realistic_views(shape, V, (3, 3, 3), seed=21)
    emu = native.Binding(native.EMU_SO)  # the product's engine code on the host-emulation backend
    eng = emu.slab_engine(shape, world, rank, V)
    z0, z1 = rank * shape[0] // world, (rank + 1) * shape[0] // world
    for v in range(V):
        eng.set_view(v, views[v][z0:z1], w[v][z0:z1], k1[v], k2[v])
    eng.set_psi(psi0[z0:z1])
    nm, nn = eng.buffer_sizes()
    a_main, b_main = torch.zeros(nm), torch.zeros(nm)
    a_nyq = torch.zeros(nn) if nn else None
    b_nyq = torch.zeros(nn) if nn else None
    eng.bind_buffers(a_main.data_ptr(), b_main.data_ptr(), a_nyq.data_ptr() if nn else None,
                     b_nyq.data_ptr() if nn else None)
    SlabDriver(eng, a_main, b_main, a_nyq, b_nyq, dist).run(2, V, 0.006, 1e-4)
    np.save(os.path.join(out_dir, "slab_rank%d.npy" % rank), eng.get_psi())
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", sorted(SLAB_CASES))
def test_two_rank_slab_sweep_matches_sequential(tmp_path, case):
    """Two gloo ranks, each with half the planes, must reproduce the single-process sequential
    sweep: to rounding (1e-6) against the same engine on one rank, within float tolerance against the
    CPU oracle of the reference's loop."""
    from libmultiviewnative_amd import native
    from oracle import binding as orc
    from ref_fixtures import realistic_views
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_slab_worker, args=(2, port, str(tmp_path), case), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / ("slab_rank%d.npy" % r)) for r in range(2)], axis=0)
    shape, V = SLAB_CASES[case]
    _, views, k1, k2, w, psi0 = realistic_views(shape, V, (3, 3, 3), seed=21)
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 2)
    ref = orc.cpu_deconvolve(psi0, h, 2)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
    emu = native.Binding(native.EMU_SO)
    e = emu.engine(shape, V)
    for v in range(V):
        e.set_view(v, views[v], w[v], k1[v], k2[v])
    e.set_psi(psi0)
    e.iterate(2, 0.006, 1e-4)
    e.sync()
    one = e.get_psi()
    e.close()
    assert np.abs(got - one).max() <= 1e-6 * np.abs(one).max()


# ---- dim0 slabs with halo exchange (HaloSlabDriver): the reference's update order on several ranks ---------------
HALO_CASES = {"two_ranks": ((24, 16, 32), 2, 2, (7, 3, 5)), "four_ranks_even_depth": ((32, 16, 32), 2, 4, (4, 5, 3)),
              "three_ranks_deep_psf": ((48, 32, 16), 1, 3, (15, 3, 3)),
              # one Inf voxel of psi in the middle of rank 0's slab, beyond the reach of its neighbours' halos, and rank 2
              # is not even a neighbour: the reference's FFT convolution floods the WHOLE volume
              # (inc/cpu_convolve.h:256-268), so the ranks have to tell each other (poison word, MAX-reduced behind
              # every dim0 leg; without that rank 2 gets an ordinary update from view 0)
              "four_ranks_inf_voxel": ((64, 16, 32), 2, 4, (5, 3, 3))}
HALO_INF_AT = (7, 5, 5)
HALO_ITS = 3


def _halo_worker(rank, world, port, out_dir, case):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MVN_DIM0_DIRECT_MIN_ITEMS"] = "0"  # the direct dim0 leg at every size: what the mode is built on
    from libmultiviewnative_amd import native
    from libmultiviewnative_amd.sharded import HaloSlabDriver
    from ref_fixtures import realistic_views
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape, V, _, ks = HALO_CASES[case]
    _, views, k1, k2, w, psi0 = realistic_views(shape, V, ks)
    if "inf_voxel" in case:
        psi0[HALO_INF_AT] = np.inf
    emu = native.Binding(native.EMU_SO)
    drv = HaloSlabDriver(emu, shape, V, ks[0], dist=dist, rank=rank, world=world)
    sl = slice(drv.z0, drv.z0 + drv.nz)
    for v in range(V):
        drv.set_view(v, views[v][sl], w[v][sl], k1[v], k2[v])
    drv.set_psi(psi0[sl])
    drv.run(HALO_ITS, 0.006, 1e-4)
    np.save(os.path.join(out_dir, "halo_rank%d.npy" % rank), drv.get_psi())
    drv.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", sorted(HALO_CASES))
def test_halo_slab_sweep_matches_the_sequential_oracle(tmp_path, case):
    from oracle import binding as orc
    from ref_fixtures import realistic_views
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    shape, V, world, ks = HALO_CASES[case]
    mp.spawn(_halo_worker, args=(world, port, str(tmp_path), case), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("halo_rank%d.npy" % r)) for r in range(world)], axis=0)
    _, views, k1, k2, w, psi0 = realistic_views(shape, V, ks)
    if "inf_voxel" in case:
        psi0[HALO_INF_AT] = np.inf
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, HALO_ITS)
    ref = orc.cpu_deconvolve(psi0, h, 2)  # the reference's own (sequential) order: no Jacobi deviation
    assert got.shape == ref.shape
    if "inf_voxel" in case:  # every update is the exact minValue blend (the voxel itself stays NaN): bit-equal
        assert np.isnan(ref).sum() == 1 and np.array_equal(got, ref, equal_nan=True)
        return
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
