"""The C-ABI library loads on a box without a GPU and exports every symbol the headers declare
(no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from libmultiviewnative_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def product():
    if not os.path.exists(native.PRODUCT_SO):
        import __graft_entry__
        __graft_entry__.build()
    return C.CDLL(native.PRODUCT_SO)


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"#\s*define[^\n]*", "", txt)  # the MVN_API macro definitions themselves
    return re.findall(r"MVN_API[^;(]*?\b(\w+)\s*\(", txt)


def test_headers_and_binding_lists_agree():
    assert sorted(declared("multiviewnative.h")) == sorted(native.REFERENCE_ABI_SYMBOLS)
    assert sorted(declared("mvn_engine_api.h")) == sorted(native.ENGINE_ABI_SYMBOLS)


def test_every_declared_symbol_is_exported(product):
    for name in declared("multiviewnative.h") + declared("mvn_engine_api.h"):
        assert hasattr(product, name), name


def dynamic_symbols(path):
    """every DEFINED symbol of the dynamic table, whatever its type (T, W, V, u, B, D, ...)"""
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    return {l.split()[-1].split("@")[0]: l.split()[-2] for l in out.splitlines() if len(l.split()) >= 3}


def test_only_the_abi_is_exported():
    # no kernel stubs (V), no weak libstdc++ instantiations (W), no HIP fat-binary bookkeeping (B/D):
    # the linker version script csrc/mvn_exports.map keeps everything but the C-ABI local
    syms = dynamic_symbols(native.PRODUCT_SO)
    allowed = set(native.REFERENCE_ABI_SYMBOLS) | set(native.ENGINE_ABI_SYMBOLS) | {"MVN_1"}
    extra = {s: t for s, t in syms.items() if s not in allowed}
    assert not extra, extra
    assert all(syms[s] == "T" for s in allowed - {"MVN_1"}), syms


def test_version_script_lists_exactly_the_headers():
    txt = open(os.path.join(ROOT, "libmultiviewnative_amd", "csrc", "mvn_exports.map")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    listed = re.findall(r"^\s+(\w+);", txt[txt.index("global:"):txt.index("local:")], flags=re.M)
    assert sorted(listed) == sorted(declared("multiviewnative.h") + declared("mvn_engine_api.h"))


def test_no_cpu_entry_points_in_the_product(product):
    # the CPU path is test infrastructure (oracle/), never shipped: no fallback
    for name in ("inplace_cpu_deconvolve", "inplace_cpu_convolution"):
        assert not hasattr(product, name)


def test_product_does_not_link_the_oracle():
    out = subprocess.check_output(["ldd", native.PRODUCT_SO]).decode()
    assert "oracle" not in out and "mvn_emu" not in out
    assert "libamdhip64" in out


def test_capitalised_names_resolve():
    for n in ("libMultiViewNative.so", "libMultiviewNative.so"):
        assert os.path.realpath(os.path.join(native.LIB_DIR, n)) == os.path.realpath(native.PRODUCT_SO)


def test_struct_layout_matches_jna():
    from libmultiviewnative_amd.abi import ViewData, Workspace
    assert C.sizeof(ViewData) == 64 and C.sizeof(Workspace) == 32
    assert [f[0] for f in Workspace._fields_] == ["data_", "num_views_", "lambda_", "minValue_", "num_iterations_"]
    assert (Workspace.num_views_.offset, Workspace.lambda_.offset, Workspace.minValue_.offset,
            Workspace.num_iterations_.offset) == (8, 16, 24, 28)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(native.MvnError):
        native.Binding(str(tmp_path / "nope.so"))


def test_bench_keeps_foreign_output_off_stdout():
    # bench.py's contract is ONE JSON line on stdout; libraries that print there (RCCL's version
    # banner) are routed to stderr by bench.stdout_to_stderr, C-level writes included
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); import bench\n"
            "with bench.stdout_to_stderr():\n"
            "    print('python-level noise'); os.write(1, b'c-level noise\\n')\n"
            "print('{\"ok\": true}')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == '{"ok": true}', r.stdout
    assert "python-level noise" in r.stderr and "c-level noise" in r.stderr


def test_product_build_has_no_timing_experiment_switches():
    # switches that change what a kernel computes (some give WRONG results) or how much LDS it
    # asks for are timing experiments; they compile only under -DMVN_EXPERIMENTS, which the product target never sets
    csrc = os.path.join(ROOT, "libmultiviewnative_amd", "csrc")
    plan = subprocess.check_output(["make", "-n", "-B", "-C", csrc, "all"]).decode()
    assert "libmultiviewnative.so" in plan and "MVN_EXPERIMENTS" not in plan and "MVN_PROBE" not in plan
    blob = open(native.PRODUCT_SO, "rb").read()
    for trace in (b"MVN_WR_LDS_PAD_KB", b"MVN_PROBE_WRAP", b"MVN_EXP_", b"MVN_D0_LDS_PAD"):
        assert trace not in blob, trace
    # every experiment switch in the sources sits behind the one define
    for name in os.listdir(csrc):
        if not name.endswith((".hpp", ".hip", ".cpp")):
            continue
        for line in open(os.path.join(csrc, name)):
            if re.match(r"\s*#\s*if", line) and re.search(r"MVN_(EXP_|FX_NO_LDS_FUSED|WR_LDS_PAD_KB|D0_LDS_PAD_KB)", line):
                assert "MVN_EXPERIMENTS" in line, (name, line)
