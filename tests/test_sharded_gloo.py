"""world_size-2 gloo run of the view-sharded driver on CPU: the product's partition + loop
(libmultiviewnative_amd/sharded.py) around an oracle-backed engine must reproduce the oracle's
single-process simultaneous mode.  (On the GPU the same driver wraps the HIP engine.)"""
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


class OracleShardEngine:
    """Engine-like test double: the oracle computes this rank's partial correction."""

    def __init__(self, psi, holder, my_views):
        from oracle import binding as orc
        self.orc = orc
        self.psi = psi.copy()
        self.holder = holder
        self.my = my_views
        self.delta = torch.zeros(psi.shape, dtype=torch.float32)

    def compute_delta(self, lam, minv):
        d = np.zeros_like(self.psi)
        if self.my:
            d = self.orc.simultaneous_step(self.psi, self.holder, self.my[0], self.my[-1] + 1, 1)
        self.delta.copy_(torch.from_numpy(d))

    def apply_delta(self):
        self.psi = self.psi + self.delta.numpy()

    def sync(self):
        pass


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from ref_fixtures import realistic_views
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape = (12, 10, 14)
    _, views, k1, k2, w, psi0 = realistic_views(shape, 3, (3, 3, 3))
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 1)
    eng = OracleShardEngine(psi0, h, view_partition(3, world, rank))
    SimultaneousDriver(eng, eng.delta, dist).run(3, 0.006, 1e-4)
    np.save(os.path.join(out_dir, "psi_rank%d.npy" % rank), eng.psi)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    from oracle import binding as orc
    from ref_fixtures import realistic_views
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "psi_rank0.npy")
    b = np.load(tmp_path / "psi_rank1.npy")
    assert np.array_equal(a, b)  # replicas stay identical
    shape = (12, 10, 14)
    _, views, k1, k2, w, psi0 = realistic_views(shape, 3, (3, 3, 3))
    h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 3)
    ref = orc.cpu_deconvolve_simultaneous(psi0, h, 1)
    assert np.abs(a - ref).max() <= 2e-6 * np.abs(ref).max()


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
