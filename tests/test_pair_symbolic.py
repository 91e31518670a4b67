"""Pair mode of the symbolic analysis (k_kkt5: two 16-pivot stages behind one set of barriers) on the CPU.

tests/emul/pair_emul.cpp (test infrastructure, compiled here with g++) emulates the kernel's data flow at the level of matrices by
slot -- pair records through their gather table into cells, cells and extracted Schur columns into the pivot columns, the pair
step V1 / P2' / V2 / W1 with its blanking rules, the Schur update, the extraction of the next columns -- on the tables
Symbolic::build emits in pair mode, and compares the factor panels (w, V per 16-pivot stage: what k_chord and the sweeps read) with
a dense block elimination of the same matrix in position order."""
import ctypes as C
import os
import subprocess

import pytest

from qtos_amd import capi
from qtos_amd.config import PlannerConfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emul", "pair_emul.cpp")
OUT = os.path.join(HERE, "emul", "_build", "libpair_emul.so")


@pytest.fixture(scope="module")
def emul():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    csrc = os.path.join(os.path.dirname(HERE), "quadruped-trajectory-optimization-stack_amd", "csrc")
    deps = [SRC] + [os.path.join(csrc, f) for f in ("symbolic.hpp", "model.hpp")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", SRC, "-o", OUT])
    lib = C.CDLL(OUT)
    lib.qtos_pair_emul.argtypes = [C.POINTER(capi.QtosParams), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]

    def run(cfg, pair):
        p = capi.params_from_config(cfg)
        out, info = (C.c_double * 8)(), (C.c_int * 8)()
        rc = lib.qtos_pair_emul(C.byref(p), pair, out, info)
        keys = ("n_stages", "front", "n_records", "max_srec", "max_drec", "n_cells", "n_cont", "n_positions")
        return rc, list(out)[:4], dict(zip(keys, info))
    return run


CASES = {   # name: (config, stages and front of the standard analysis -- short stages where they cost none --, of the pair-mode analysis)
    "reference_compat": (lambda: PlannerConfig.reference_compat(reduce_swing=False), 71, 96, 72, 112),
    "reference_compat_full": (lambda: PlannerConfig.reference_compat(reduce_base=False, reduce_swing=False), 106, 96, 106, 112),
    "knots100_walk": (lambda: PlannerConfig.knots100(reduce_swing=False), 108, 112, 108, 128),
    "knots100_trot": (lambda: PlannerConfig.knots100(gait="trot", reduce_swing=False), 127, 112, 128, 128),
    # the default: reduced swings (the mid nodes' columns folded onto the footholds)
    "reference_compat_swing": (lambda: PlannerConfig.reference_compat(), 63, 96, 64, 112),
    "knots100_walk_swing": (lambda: PlannerConfig.knots100(), 100, 112, 100, 128),
    "knots100_trot_swing": (lambda: PlannerConfig.knots100(gait="trot"), 113, 96, 114, 112),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_pair_mode_tables_give_the_factor_panels_of_the_block_elimination(emul, name):
    make, ns, front, ns_pair, front_pair = CASES[name]
    cfg = make()
    rc, _, std = emul(cfg, 0)
    assert rc == 0 and (std["n_stages"], std["front"]) == (ns, front)
    rc, (err_v, err_w, unstored, worst_stage), info = emul(cfg, 1)
    assert rc == 0, rc
    # whole pairs, one record per pair, no continuation records; the front grows by at most one 16-slot group
    assert info["n_stages"] == ns_pair and info["n_stages"] % 2 == 0 and info["n_records"] * 2 == info["n_stages"]
    assert info["front"] == front_pair and info["n_cont"] == 0
    # panels of the emulated pair kernel against the dense elimination, relative to each stage's largest entry
    assert err_v <= 1e-9, (err_v, worst_stage)
    assert err_w <= 1e-9
    # every row of a reference panel that is not zero is a row the stage's mask stores
    assert unstored == 0.0
