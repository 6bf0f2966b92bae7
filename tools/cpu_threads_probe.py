"""How many threads the CPU baseline should use on this host: cgroup quota, affinity, and the
oracle's 256^3 single view update at several thread counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import binding as orc
from libmultiviewnative_amd.abi import WorkspaceHolder
import bench
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.strerror)
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "oracle default threads", orc.threads(-1))
shape = (256, 256, 256)
view, k1, k2 = bench.make_view(shape, 0, 15)
h = WorkspaceHolder([view], [k1], [k2], [np.ones(shape, np.float32)], bench.LAMBDA, bench.MIN_VALUE, 1)
psi0 = np.full(shape, np.float32(35.0), np.float32)
for nt in (1, 4, 8, 16, 32, 64, 128, 256, -1):
    if nt > os.cpu_count():
        continue
    orc.cpu_deconvolve(psi0, h, nt)
    setup_s, loop_s = orc.last_timing()
    print("threads %4d: loop %.3f s, PSF setup %.3f s" % (orc.threads(nt), loop_s, setup_s), flush=True)
