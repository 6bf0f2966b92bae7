"""Loaders for the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np

from libmultiviewnative_amd.abi import WorkspaceHolder

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixture_a():
    return np.load(os.path.join(GOLDEN, "fixture_a.npz"))


def rl_small(lam):
    g = np.load(os.path.join(GOLDEN, "rl_small.npz"))
    nv = sum(1 for k in g.files if k.startswith("view"))
    views = [g["view%d" % v] for v in range(nv)]
    w = [g["weights%d" % v] for v in range(nv)]
    k1 = [g["kernel1_%d" % v] for v in range(nv)]
    k2 = [g["kernel2_%d" % v] for v in range(nv)]
    h = WorkspaceHolder(views, k1, k2, w, lam, float(g["min_value"]), int(g["iterations"]))
    tag = "lam%g" % lam
    return g["psi0"], h, g["expect_sequential_" + tag], g["expect_simultaneous_" + tag]
