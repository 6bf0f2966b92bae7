"""One iteration of the sharded step as a timeline (profiles/r03_overlap_timeline.md) from the rocpd
database of `rocprofv3 --kernel-trace -- python3 bench.py --gpus 1 --force-dist ...`:
    python tools/timeline.py <dir-or-db> [iteration-from-the-end=2] > profiles/r03_overlap_timeline.md
Lists every kernel (and memory copy) between two consecutive launches of the first kernel of the
step's head, with its queue, start and end relative to the first one."""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dbs = [path] if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    c = sqlite3.connect(dbs[0])
    rows = list(c.execute("select name, queue_id, stream_id, start, end, grid_x, workgroup_x from kernels order by start"))
    short = lambda n: n.replace("void ", "").split("(")[0]
    # an iteration of the sharded step ends with the apply_delta kernels (k_axpy1 on chunk ranges)
    marks = [i for i, r in enumerate(rows) if short(r[0]).startswith("k_axpy1")]
    # group consecutive axpy launches (one per chunk) into iterations: a gap of > 20 kernels starts a new one
    iters, last = [], None
    for i in marks:
        if last is None or i - last > 40:
            iters.append([i, i])
        else:
            iters[-1][1] = i
        last = i
    if len(iters) < back + 1:
        sys.exit("not enough iterations in the trace (%d)" % len(iters))
    lo = iters[-back - 1][1] + 1   # first kernel after the previous iteration's last axpy ... (its trailing passes included below)
    hi = iters[-back][1]
    # walk back to include the kernels that follow the previous iteration's last axpy only
    t0 = rows[lo][3]
    print("# One iteration of the sharded (Jacobi) step on one rank: kernel timeline\n")
    print("`rocprofv3 --kernel-trace -- python3 bench.py --gpus 1 --force-dist --backend nccl --chunks 4 --steps 6 --warmup 2 "
          "--no-side` (512^3, 6 views on the one rank; the all-reduce of each chunk is issued through RCCL although "
          "the world has one rank).  Times in microseconds from the first kernel of the iteration; one row per kernel.\n")
    print("| # | kernel | queue | start | end | duration |")
    print("|---|---|---|---|---|---|")
    for n, r in enumerate(rows[lo:hi + 1]):
        print("| %d | `%s` | %s | %.1f | %.1f | %.1f |" % (n, short(r[0])[:70], r[1], (r[3] - t0) / 1e3, (r[4] - t0) / 1e3,
                                                           (r[4] - r[3]) / 1e3))
    print("\nIteration wall time in the trace: %.3f ms." % ((rows[hi][4] - t0) / 1e6))


if __name__ == "__main__":
    main()
