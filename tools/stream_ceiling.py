"""Practical HBM streaming ceilings on this GPU for the read/write mixes of the RL passes (torch
elementwise kernels on 512^3 float32 volumes): what "memory-bound" can mean per pass."""
import torch, time
n = 512 ** 3
dev = "cuda:0"
a, b, c, d, e = (torch.rand(n, device=dev) for _ in range(5))
def bench(name, fn, nbytes, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps): fn()
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / reps
    print("%-34s %.4f ms  %.0f GB/s" % (name, ms, nbytes / ms / 1e6), flush=True)
v = 4.0 * n
bench("copy 1R+1W (out-of-place)", lambda: c.copy_(a), 2 * v)
bench("scale in place 1R+1W", lambda: a.mul_(1.0001), 2 * v)
bench("mul 2R+1W (c = a*b)", lambda: torch.mul(a, b, out=c), 3 * v)
bench("mul in place 2R+1W (a *= b)", lambda: a.mul_(b), 3 * v)
bench("addcmul 3R+1W (d = a + b*c)", lambda: torch.addcmul(a, b, c, out=d), 4 * v)
bench("read only (sum) 1R", lambda: a.sum(), v)
bench("write only (fill) 1W", lambda: c.fill_(1.0), v)
