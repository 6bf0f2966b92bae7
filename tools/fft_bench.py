"""3-D FFT micro-benchmark: the counterpart of the reference's bench_gpu_nd_fft / bench_gpu_many_nd_fft
family (bench/bench_gpu_nd_fft.cu, bench/bench_gpu_many_nd_fft.cu:403-463) driven over the size
ladder of python/generate_dims.py:4-48 (16^3, 32x16x16, 32x32x16, ... style growth, one axis at a
time), printing the reference's record columns (bench/logging.hpp:9-60) plus the resident figures.

    python tools/fft_bench.py [-s 6] [-e 10] [--batch 8] [--reps 10]

`total_time_ms` is `reps` transforms with the stack resident in HBM (events on the launch
stream); the "comment" column carries ms per transform, the 6B algorithmic GB/s and,
for the batched sweep, ms per stack.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libmultiviewnative_amd import native  # noqa: E402


def size_ladder(begin, end, base=2, n_dims=3):
    """Shapes from base^begin cubed up to base^(end-1) cubed, growing one axis at a time (same
    sequence as the reference's produce_size_strings)."""
    cards = [base ** c for c in range(begin, end)]
    if not cards:
        return []
    cur = [cards[0]] * n_dims
    out = [tuple(cur)]
    idx = 1
    while idx < len(cards):
        if cards[idx - 1] in cur:
            cur[cur.index(cards[idx - 1])] = cards[idx]
            out.append(tuple(cur))
        else:
            idx += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-s", "--start", type=int, default=6)
    ap.add_argument("-e", "--end", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    lib = native.lib()
    import ctypes
    buf = ctypes.create_string_buffer(256)
    lib.l.getNameDeviceCUDA(args.device, buf)
    name = buf.value.decode() or "gpu"
    print("n_devices dev_type dev_name n_repeats total_time_ms stack_dims_x stack_dims_y stack_dims_z "
          "type_width_byte comment", flush=True)
    for shape in size_ladder(args.start, args.end):
        B = 4 * shape[0] * shape[1] * 2 * (shape[2] // 2 + 1)
        for direction, tag in ((0, "r2c"), (1, "c2r")):
            ms, _ = lib.fft3_profile(shape, direction, args.reps, args.device)
            print(1, "gpu_%s_resident" % tag, name.replace(" ", "_"), args.reps, "%.4f" % (ms * args.reps),
                  shape[0], shape[1], shape[2], 4,
                  "ms_per_fft=%.4f,GBps_6B=%.0f" % (ms, 6 * B / ms / 1e6), flush=True)
        if args.batch * 2 * B < 64 << 30:
            ms = lib.fft3_many_time(shape, args.batch, 0, max(1, args.reps // 2), args.device)
            print(1, "gpu_many_r2c_resident", name.replace(" ", "_"), max(1, args.reps // 2),
                  "%.4f" % (ms * max(1, args.reps // 2)), shape[0], shape[1], shape[2], 4,
                  "batch=%d,ms_per_stack=%.4f,GBps_6B=%.0f" % (args.batch, ms / args.batch,
                                                               6 * B * args.batch / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
