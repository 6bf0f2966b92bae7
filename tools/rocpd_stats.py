"""Summaries of rocprofv3 output (the rocpd SQLite database ROCm 7.2 writes by default).

    python tools/rocpd_stats.py stats   <dir-or-db> [out.csv]     per-kernel launch statistics
                                                                   (what --kernel-trace --stats prints)
    python tools/rocpd_stats.py counters <out.md> <title> <dir-or-db> [<dir-or-db> ...]
                                                                   per-kernel PMC averages of several
                                                                   --pmc passes, one markdown table

Used for the files under profiles/: the GPU box runs rocprofv3, this turns its database into the
small text files that are committed.
"""
import collections
import glob
import os
import sqlite3
import sys


def databases(path):
    if os.path.isfile(path):
        return [path]
    return sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))


def short(name):
    return name.replace("void ", "").split("(")[0]


def kernel_stats(path):
    rows = collections.OrderedDict()
    for db in databases(path):
        c = sqlite3.connect(db)
        q = ("select name, count(*), sum(end - start), min(end - start), max(end - start), "
             "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
             "max(workgroup_x), max(grid_x) from kernels group by name")
        for r in c.execute(q):
            rows[r[0]] = r[1:]
    return rows


def cmd_stats(path, out):
    rows = kernel_stats(path)
    total = float(sum(r[1] for r in rows.values())) or 1.0
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,VGPRs,AGPRs,SGPRs,LDS_bytes,Scratch_bytes,"
             "Workgroup,Grid"]
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        n, tot, mn, mx, vg, ag, sg, lds, scr, wg, grid = r
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d,%s,%s,%s,%s,%s,%s,%s' % (
            name, n, tot, tot / n, 100.0 * tot / total, mn, mx, vg, ag, sg, lds, scr, wg, grid))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


def counter_averages(path):
    """{kernel: {counter: (sum, launches)}} of one pass"""
    res = collections.defaultdict(dict)
    for db in databases(path):
        c = sqlite3.connect(db)
        cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        # one row per (dispatch, counter[, dimension instance]): sum the instances of a dispatch first
        q = ("select %s, counter_name, dispatch_id, sum(value) from counters_collection group by %s, "
             "counter_name, dispatch_id" % (name_col, name_col))
        per = collections.defaultdict(list)
        for kname, cname, _, val in c.execute(q):
            per[(kname, cname)].append(float(val))
        for (kname, cname), vals in per.items():
            res[kname][cname] = (sum(vals), len(vals))
    return res


def cmd_counters(out, title, paths):
    merged = collections.defaultdict(dict)
    for p in paths:
        for k, d in counter_averages(p).items():
            merged[k].update(d)
    names = []
    for d in merged.values():
        for cn in d:
            if cn not in names:
                names.append(cn)
    names.sort()
    lines = ["# " + title, "", "Averages per launch, summed over the device (rocprofv3 --pmc passes, "
             "summarised by tools/rocpd_stats.py).", ""]
    derived = []
    if "SQ_LDS_BANK_CONFLICT" in names and "SQ_LDS_IDX_ACTIVE" in names:
        derived.append(("LDS conflict share", lambda a: a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]
                        if a.get("SQ_LDS_IDX_ACTIVE") else float("nan")))
    if "SQ_WAIT_INST_LDS" in names and "SQ_WAVE_CYCLES" in names:
        derived.append(("LDS issue stall / wave-cycle", lambda a: a["SQ_WAIT_INST_LDS"] / a["SQ_WAVE_CYCLES"]
                        if a.get("SQ_WAVE_CYCLES") else float("nan")))
    head = ["kernel", "launches"] + names + [d[0] for d in derived]
    lines.append("| " + " | ".join(head) + " |")
    lines.append("|" + "---|" * len(head))
    for k in sorted(merged):
        d = merged[k]
        if not any(cn.startswith("SQ_") or cn.startswith("TCC") or cn.endswith("_SIZE") for cn in d):
            continue
        avg = {cn: s / n for cn, (s, n) in d.items() if n}
        launches = max(n for (_, n) in d.values())
        if launches < 2 and len(merged) > 12:
            continue
        cells = ["`%s`" % short(k), str(launches)] + ["%.0f" % avg[cn] if cn in avg else "-" for cn in names]
        for _, fn in derived:
            try:
                cells.append("%.3f" % fn(avg))
            except KeyError:
                cells.append("-")
        lines.append("| " + " | ".join(cells) + " |")
    text = "\n".join(lines) + "\n"
    open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "stats":
        cmd_stats(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    elif len(sys.argv) >= 5 and sys.argv[1] == "counters":
        cmd_counters(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        sys.exit(__doc__)
