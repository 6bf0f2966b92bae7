"""HBM traffic per launch of every kernel of tools/l2_plane_probe.hip from the rocprofv3 counter passes
(tools/r03_l2_probe_a2.sh).  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section): the read side is doubled."""
import collections
import csv
import glob
import os
import sys


def read(dirname):
    tot = collections.defaultdict(lambda: collections.Counter())
    cnt = collections.defaultdict(lambda: collections.Counter())
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("void ", "").split("(")[0]
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
    return {k: {c: tot[k][c] / cnt[k][c] for c in tot[k]} for k in tot}


def main():
    out = sys.argv[1]
    rd, wr, tcc = read(os.path.join(out, "a2_rd")), read(os.path.join(out, "a2_wr")), read(os.path.join(out, "a2_tcc"))
    vol = 512.0 ** 3 * 4
    print("| kernel | HBM read / launch (volumes) | HBM written / launch (volumes) | L2 hit rate |")
    print("|---|---|---|---|")
    for k in sorted(set(rd) | set(wr)):
        r = 2.0 * rd.get(k, {}).get("FETCH_SIZE", float("nan")) * 1024 / vol
        w = wr.get(k, {}).get("WRITE_SIZE", float("nan")) * 1024 / vol
        t = tcc.get(k, {})
        hit = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]) if "TCC_HIT_sum" in t and t["TCC_HIT_sum"] + t["TCC_MISS_sum"] > 0 else float("nan")
        print("| `%s` | %.2f | %.2f | %.2f |" % (k, r, w, hit))


if __name__ == "__main__":
    main()
