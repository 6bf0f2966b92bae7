"""profiles/r03_mem_counters.md from the rocprofv3 passes of tools/r03_mem_counters.sh:
    python tools/mem_counters_md.py gpurun_out/r03/mem > profiles/r03_mem_counters.md
Per-launch averages (summed over the device) of the memory-path counters of the hot kernels at
512^3, plus the derived figures the text argues with."""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rocpd_stats

KERNELS = [("kx_rows_r2c<256>", "rows_r2c", 2.0), ("kx_strided<512, 0>", "axis1_fwd", 2.0),
           ("kx_strided<512, 1>", "axis1_inv", 2.0), ("kx_strided<512, 2>", "axis0_fused", 3.0),
           ("kd_dim0<31, 4>", "axis0_direct", 2.06),
           ("kw_rows<2, 1>", "rows_fused_div", 3.0), ("kw_rows<2, 2>", "rows_fused_upd", 5.0),
           # configs[4]'s extents (1920 x 1920 planes)
           ("kx_rows_r2c<960>", "rows_r2c 960", 2.0), ("kx_strided_split<1920, 0>", "axis1_fwd 1920", 2.0),
           ("kx_strided_split<1920, 1>", "axis1_inv 1920", 2.0), ("kx_strided<1920, 0>", "axis1_fwd 1920 (8 col)", 2.0),
           ("kx_rows_c2r_r2c<960, 1>", "rows_fused_div 960", 3.0), ("kx_rows_c2r_r2c<960, 2>", "rows_fused_upd 960", 5.0),
           ("kd_dim0<15, 4>", "axis0_direct K=15", 2.06)]


def main():
    root = sys.argv[1]
    merged = {}
    for p in sorted(glob.glob(os.path.join(root, "p*"))):
        if not os.path.isdir(p):
            continue
        for k, d in rocpd_stats.counter_averages(p).items():
            merged.setdefault(rocpd_stats.short(k), {}).update({c: s / n for c, (s, n) in d.items() if n})
    cols = [(k, lbl, vols) for k, lbl, vols in KERNELS if k in merged]
    names = sorted({c for k, _, _ in cols for c in merged[k]})
    print("| counter (sum over the device, per launch) | " + " | ".join("%s `%s`" % (l, k) for k, l, _ in cols) + " |")
    print("|---|" + "---|" * len(cols))
    for c in names:
        print("| %s | " % c + " | ".join("%.4g" % merged[k].get(c, float("nan")) for k, _, _ in cols) + " |")
    print()
    print("| derived | " + " | ".join(l for _, l, _ in cols) + " |")
    print("|---|" + "---|" * len(cols))

    def row(label, fn, fmt="%.3g"):
        cells = []
        for k, _, vols in cols:
            try:
                cells.append(fmt % fn(merged[k], vols))
            except (KeyError, ZeroDivisionError):
                cells.append("-")
        print("| %s | " % label + " | ".join(cells) + " |")

    row("read requests from the CUs to the L2 (128 B each), millions", lambda m, v: m["TCP_TCC_READ_REQ_sum"] / 1e6)
    row("write requests (64 B each), millions", lambda m, v: m["TCP_TCC_WRITE_REQ_sum"] / 1e6)
    row("average read latency seen by the CU's L1 (cycles per request)", lambda m, v: m["TCP_TCC_READ_REQ_LATENCY_sum"] / m["TCP_TCC_READ_REQ_sum"], "%.0f")
    row("average write latency (cycles per request)", lambda m, v: m["TCP_TCC_WRITE_REQ_LATENCY_sum"] / m["TCP_TCC_WRITE_REQ_sum"], "%.0f")
    row("reads in flight, device-wide (latency sum / L2 cycles per channel)", lambda m, v: m["TCP_TCC_READ_REQ_LATENCY_sum"] / (m["TCC_CYCLE_sum"] / 128.0), "%.0f")
    row("L1 busy cycles stalled on pending misses (PENDING_STALL / GATE_EN1)", lambda m, v: m["TCP_PENDING_STALL_CYCLES_sum"] / m["TCP_GATE_EN1_sum"], "%.2f")
    row("L2 -> L1 return path stalled (TCR_TCP_STALL / GATE_EN1)", lambda m, v: m["TCP_TCR_TCP_STALL_CYCLES_sum"] / m["TCP_GATE_EN1_sum"], "%.3f")
    row("L2 hit rate", lambda m, v: m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), "%.2f")
    row("L2 tag stalls per request", lambda m, v: m["TCC_TAG_STALL_sum"] / m["TCC_REQ_sum"], "%.3f")
    row("HBM read-credit stall cycles per read request to memory", lambda m, v: m["TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"] / m["TCC_EA0_RDREQ_sum"], "%.3f")
    row("HBM write-credit stall cycles per write request to memory", lambda m, v: m["TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"] / m["TCC_EA0_WRREQ_sum"], "%.3f")
    row("reads outstanding at the memory side (RDREQ_LEVEL / L2 cycles per channel)", lambda m, v: m["TCC_EA0_RDREQ_LEVEL_sum"] / (m["TCC_CYCLE_sum"] / 128.0), "%.0f")
    row("address translations missing the CU's TLB, per million requests", lambda m, v: 1e6 * m["TCP_UTCL1_TRANSLATION_MISS_sum"] / m["TCP_UTCL1_REQUEST_sum"], "%.1f")


if __name__ == "__main__":
    main()
