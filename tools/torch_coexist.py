"""Checks that torch (imported first) and the product library share one HIP runtime, and that a
world_size-1 RCCL all-reduce works on a tensor the engine writes to."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.distributed as dist
import numpy as np
torch.cuda.set_device(0)
x = torch.ones(4, device="cuda:0")
from libmultiviewnative_amd import native
from libmultiviewnative_amd.sharded import SimultaneousDriver
from libmultiviewnative_amd.abi import WorkspaceHolder
from ref_fixtures import realistic_views
from oracle import binding as orc
lib = native.lib()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
shape = (24, 20, 28)
_, views, k1, k2, w, psi0 = realistic_views(shape, 2, (5, 5, 5))
eng = lib.engine(shape, 2)
for v in range(2):
    eng.set_view(v, views[v], w[v], k1[v], k2[v])
eng.set_psi(psi0)
delta = torch.zeros(eng.psi_ptr()[1], dtype=torch.float32, device="cuda:0")
eng.bind_delta(delta.data_ptr())
class OneRank:  # force the collective even at world_size 1
    ReduceOp = dist.ReduceOp
    def get_world_size(self): return 2
    def all_reduce(self, t, op=None): dist.all_reduce(t, op=op)
SimultaneousDriver(eng, delta, OneRank(), after_collective=lambda: torch.cuda.current_stream().synchronize()).run(3, 0.006, 1e-4)
got = eng.get_psi()
h = WorkspaceHolder(views, k1, k2, w, 0.006, 1e-4, 3)
ref = orc.cpu_deconvolve_simultaneous(psi0, h, 2)
print("torch+engine+rccl ok, max rel err", float(np.abs(got - ref).max() / np.abs(ref).max()), "delta nonzero", bool(delta.abs().sum().item() > 0))
libs = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l]
print(sorted(set(libs)))
dist.destroy_process_group()
