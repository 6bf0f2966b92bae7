"""Instruction statistics of device kernels in an assembly listing (hipcc -S --cuda-device-only).

    python tools/asm_stats.py listing.s 'kw_rows<2, 1>' 'kx_strided<512, 2>' ...
prints, per kernel whose demangled name contains the pattern: instructions, VALU, packed VALU, f64,
LDS and global instructions, VGPRs, scratch, occupancy.
"""
import re
import subprocess
import sys


def kernels(path):
    name, body, out = None, [], {}
    for line in open(path, errors="replace"):
        m = re.match(r"^([_A-Za-z0-9]+):\s", line)
        if m and m.group(1).startswith("_Z") and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line)
            if line.startswith(".Lfunc_end"):
                out[name] = body
                name = None
    return out


def meta(path):
    res, cur = {}, None
    for line in open(path, errors="replace"):
        m = re.match(r"^([_A-Za-z0-9]+):\s", line)
        if m and m.group(1).startswith("_Z"):
            cur = m.group(1)
        m = re.match(r"^; (NumVgprs|ScratchSize|Occupancy): (\d+)", line)
        if m and cur:
            res.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    return res


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    ks, mt = kernels(path), meta(path)
    names = list(ks)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for mangled, d in zip(names, dem):
        if pats and not any(p in d for p in pats):
            continue
        ins = [l.split()[0] for l in ks[mangled] if re.match(r"^\s+[a-z]", l) and not l.strip().startswith(".")]
        c = lambda f: sum(1 for i in ins if f(i))
        m = mt.get(mangled, {})
        print("%-46s insts %5d valu %5d pk %4d f64 %4d lds %3d global %3d salu %4d | vgprs %s scratch %s occ %s" % (
            d.replace("void ", "").split("(")[0], len(ins), c(lambda i: i.startswith("v_")),
            c(lambda i: i.startswith("v_pk_")), c(lambda i: i.startswith("v_") and "f64" in i),
            c(lambda i: i.startswith("ds_")), c(lambda i: i.startswith("global_")),
            c(lambda i: i.startswith("s_")), m.get("NumVgprs"), m.get("ScratchSize"), m.get("Occupancy")))


if __name__ == "__main__":
    main()
