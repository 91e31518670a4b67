"""P4 boundary parity: flag strings, heightfield file bytes, CSV format, the C ABI's
exported symbols and the planner's structural dimensions (host-only analysis, no GPU)."""
import ctypes as C
import hashlib
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

FIX = json.load(open(os.path.join(GOLDEN, "boundary.json")))


def test_cmd_args_strings_equal_reference():
    from qtos_amd import flags
    for case in FIX["cmd_args"]:
        assert flags.cmd_args(case["args"]) == case["string"]


def test_parse_flags_inverts_cmd_args():
    from qtos_amd import flags
    for case in FIX["cmd_args"]:
        back = flags.parse_flags(case["string"])
        for k, v in case["args"].items():
            if k in flags.FLAGS and v:
                assert np.allclose(np.ravel(back[k]), np.ravel(v), rtol=0, atol=0)
    start, goal, t0 = flags.problem_arrays(flags.parse_flags(FIX["cmd_args"][1]["string"]))
    assert len(start) == 24 and t0 == 3.756 and goal[0] == 0.9100042764299588
    assert start[18:21] == [0.135043, -0.422901, -0.014325]


def test_heightfield_files_equal_reference(tmp_path):
    from qtos_amd import heightfield
    for name, ref in FIX["heightfield"].items():
        tiles = []
        for t in ref["tiles"]:
            fname = {"feasibility": "feasibility_test", "feasibility_1": "feasibility_test_1"}.get(t, t)
            tiles.append(heightfield.read_tile(os.path.join(GOLDEN, "heightfields", fname + ".txt")))
        m = heightfield.build_map(tiles, ref["mesh_scale"])
        assert list(m.shape) == ref["map_shape"]
        assert abs(float(np.sum(m * np.arange(m.size).reshape(m.shape))) - ref["map_checksum"]) < 1e-9
        assert heightfield.cell_size(m) == ref["resolution"]
        tw = heightfield.towr_map(m)
        assert list(tw.shape) == ref["towr_shape"]
        path = tmp_path / (name + ".txt")
        heightfield.write_height_file(str(path), tw)
        text = open(path).read()
        assert len(text) == ref["n_chars"] and text[:160] == ref["head"]
        assert hashlib.sha256(text.encode()).hexdigest() == ref["sha256"]
        if ref["text"] is not None:
            assert text == ref["text"]
        assert np.array_equal(heightfield.read_height_file(str(path)), tw)


def test_csv_format(tmp_path):
    from qtos_amd import csvio
    rows = np.zeros((3, 37))
    rows[:, 0] = [0, 0.001, 0.002]
    rows[1, 1] = 6.9309e-07
    rows[2, 3] = 0.240002
    p = tmp_path / "t.csv"
    csvio.write_csv(str(p), rows)
    lines = open(p).read().split("\n")
    assert lines[1].startswith("0.001,6.9309e-07,0,0,") and lines[2].split(",")[3] == "0.240002"
    assert np.array_equal(csvio.read_csv(str(p)), rows)
    assert csvio.COLUMN_MAP["HR_force"] == slice(34, 37)


def test_native_csv_writer_prints_every_number_like_percent_g(tmp_path, hip_lib):
    """qtos_write_csv (csrc/csv_writer.hpp: the native writer behind csvio.write_csv -- a "%g" of its own with snprintf for the
    values near a rounding boundary) against Python's "%g", byte for byte: random values over 36 decades, the switch points of the
    format (1e-5 / 1e-4, 999999.5 / 1e6), ties and near-ties of the sixth digit, zeros of both signs, inf / nan, a golden plan's rows;
    one thread and several; the error codes."""
    from conftest import load_gv
    from qtos_amd import csvio
    rng = np.random.default_rng(7)
    big = rng.standard_normal((20000, 37)) * 10.0 ** rng.integers(-18, 18, (20000, 37))
    big[::7, 3] = np.round(big[::7, 3], 3)                     # short decimals (trailing zeros stripped)
    edge = np.zeros((8, 37))
    edge[0, :12] = [0.0, -0.0, 1.0, -1.0, 123456.5, 1e-4, 1e-5, 999999.5, 999999.4999, 9.999995e-5, 99999.95, 0.24]
    edge[1, :10] = [1e5, 1e6, 1e-5, 1e-4, np.inf, -np.inf, np.nan, 3.756, 1e29, 1e-29]
    edge[2, :8] = [0.1234565, 0.1234575, 2.5e-5, 1234565.0, 1234575.0, 0.5, 1.5e300, 4e-310]
    edge[3, :] = 100000.5 + np.arange(37)                      # exact ties of the sixth digit: round half to even
    edge[4, :] = (100000.5 + np.arange(37)) * 1e-9
    edge[5, :] = np.nextafter(100000.5 + np.arange(37), 0)     # one ulp under a tie
    edge[6, :] = np.nextafter(100000.5 + np.arange(37), 1e9)   # one ulp over
    edge[7, :] = 10.0 ** np.arange(-18, 19)                    # the powers of ten themselves
    gv = load_gv("gv1")
    golden = np.asarray(gv["rows"], float)
    for name, rows in (("big", big), ("edge", edge), ("golden", golden)):
        assert rows.shape[1] == 37
        ref = tmp_path / (name + "_py.csv")
        csvio.write_csv_python(str(ref), rows)
        want = open(ref, "rb").read()
        for nt in (1, 3, 0):
            out = tmp_path / ("%s_%d.csv" % (name, nt))
            csvio.write_csv(str(out), rows, n_threads=nt)
            got = open(out, "rb").read()
            if got != want:
                la, lb = got.decode().splitlines(), want.decode().splitlines()
                bad = [(i, [(u, v) for u, v in zip(x.split(","), y.split(",")) if u != v][:3]) for i, (x, y) in enumerate(zip(la, lb)) if x != y][:3]
                raise AssertionError((name, nt, len(la), len(lb), bad))
    csvio.write_csv(str(tmp_path / "empty.csv"), np.zeros((0, 37)))
    assert open(tmp_path / "empty.csv").read() == ""
    with pytest.raises(OSError):
        csvio.write_csv(str(tmp_path / "no_such_dir" / "x.csv"), edge)


def test_c_abi_exports_every_declared_symbol(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "qtos_planner.h")).read()
    names = sorted(set(re.findall(r"\b(qtos_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 14
    from qtos_amd import capi
    assert sorted(capi.EXPORTS) == names
    for n in names:
        assert hasattr(hip_lib, n), n


def test_planner_dimensions_match_reference_log(hip_lib, cfg):
    """Same NLP as logs/towr_log.out:40-52: 1040 variables (1005 free), 706 + 1024 constraints,
    bound split 112 / 816 / 96 -- computed by the product's own host code."""
    from qtos_amd import capi
    import dataclasses
    dims = json.load(open(os.path.join(GOLDEN, "nlp_dims.json")))
    # (the NLP's own counts do not depend on how the KKT system is formed; the unknowns of the system do)
    dr, actr = capi.analyze(dataclasses.replace(cfg, reduce_base=True, reduce_swing=False))
    assert (dr.n_vars, dr.n_cons, dr.n_free, dr.n_eq, dr.n_ineq) == (1040, 1730, dims["n_vars_free"], dims["n_eq"], dims["n_ineq"])
    assert (dr.n_unknowns, dr.n_stages, dr.front) == (1121, 71, 96) and actr.max() <= dr.front
    ds, acts = capi.analyze(cfg)      # the default: reduced base and reduced swings (8 unknowns per swing leave the system)
    assert cfg.reduce_base and cfg.reduce_swing and (ds.n_unknowns, ds.n_stages, ds.front) == (1121 - 8 * 16, 63, 96) and acts.max() <= ds.front
    assert (ds.n_vars, ds.n_cons, ds.n_free, ds.n_eq, ds.n_ineq) == (dr.n_vars, dr.n_cons, dr.n_free, dr.n_eq, dr.n_ineq)
    d, act = capi.analyze(dataclasses.replace(cfg, reduce_base=False, reduce_swing=False))
    assert (d.n_unknowns, d.n_stages, d.front) == (1685, 106, 96)      # (112 slots with stage boundaries at multiples of 16 unknowns: short stages)
    assert (d.n_vars, d.n_cons, d.n_free) == (1040, 1730, dims["n_vars_free"])
    assert (d.n_eq, d.n_ineq) == (dims["n_eq"], dims["n_ineq"])
    assert (d.n_ineq_lower, d.n_ineq_both, d.n_ineq_upper) == (112, 816, 96)
    assert (d.n_base_nodes, d.n_dyn_times, d.n_rom_times, d.n_rows_csv) == (51, 52, 64, 5001)
    assert d.n_unknowns == d.n_free + d.n_eq_work and d.n_stages == -(-d.n_unknowns // 16)
    assert act.max() <= d.front <= 128 and d.front % 16 == 0
    # SURVEY.md 8d byte formula evaluated on the actual stage sizes
    assert d.kkt_algorithmic_bytes == 8 * (16 * int(act.sum()) + 2 * d.n_unknowns)


def test_knots100_structure(hip_lib):
    from qtos_amd import capi
    from qtos_amd.config import PlannerConfig
    d, act = capi.analyze(PlannerConfig.knots100())
    assert d.n_base_nodes == 101 and d.n_dyn_times == 102 and d.n_vars == 1640
    assert act.max() <= d.front <= 128
    # round 6: on a reduced base the analysis builds two elimination orders and keeps the smaller front (QtosDims.order_rule): the
    # walk and the 200-knot transcription take the order with the late force nodes (rule 1: 96 slots instead of 112), the trot and
    # the reference's own transcription the one with the early coefficients alone (rule 2); every full-base system keeps the order
    # of rounds 1 - 5 (rule 0), which is no longer chosen on a reduced base (short trots: a KKT solve accurate to 1e-3 only)
    assert (d.order_rule, d.front, d.n_stages) == (1, 96, 100)
    dt, _ = capi.analyze(PlannerConfig.knots100(gait="trot"))
    assert (dt.order_rule, dt.front, dt.n_stages) == (2, 96, 113)
    dc, _ = capi.analyze(PlannerConfig.reference_compat())
    assert (dc.order_rule, dc.front, dc.n_stages) == (2, 96, 63)
    dct, _ = capi.analyze(PlannerConfig.reference_compat(gait="trot"))
    assert (dct.order_rule, dct.front, dct.n_stages) == (2, 96, 75)      # (rule 0: 80 slots, one KKT solve accurate to 1.2e-3)
    df, _ = capi.analyze(PlannerConfig.knots100(reduce_base=False))
    assert df.order_rule == 0
    dk, _ = capi.analyze(PlannerConfig.knots200())
    assert (dk.order_rule, dk.front) == (1, 96)


def test_knots200_structure(hip_lib):
    """BASELINE configs[4]: two walk cycles over 10 s keep the elimination front at the 128 slots of the
    100-knot problem (scaling the one-cycle schedule to 10 s would need 208)."""
    from qtos_amd import capi
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.knots200()
    assert [len(f) for f in cfg.phase_durations] == [17] * 4
    assert all(abs(sum(f) - 10.0) < 1e-12 for f in cfg.phase_durations)
    import dataclasses
    d, act = capi.analyze(dataclasses.replace(cfg, reduce_base=False, reduce_swing=False))
    assert (d.n_base_nodes, d.n_dyn_times, d.n_vars, d.n_cons) == (201, 202, 3160, 4558)
    assert act.max() <= d.front == 128 and d.n_stages == 356
    dr, actr = capi.analyze(dataclasses.replace(cfg, reduce_swing=False))     # reduced base: 5685 -> 3321 unknowns
    # (round 6: the order with the late force nodes, HostModel::order_rule 1, is the smaller one here -- 112 slots instead of 128)
    assert (dr.n_vars, dr.n_cons) == (3160, 4558) and actr.max() <= dr.front == 112 and dr.n_stages == 208
    ds, acts = capi.analyze(cfg)     # ... and reduced swings (the default): 32 swings x 8 unknowns fewer; 96 slots (112 with rule 0)
    assert (ds.n_unknowns, ds.n_stages) == (dr.n_unknowns - 8 * 32, 192) and acts.max() <= ds.front == 96
    os.environ["QTOS_ORDER"] = "0"
    try:
        d0, _ = capi.analyze(cfg)
    finally:
        del os.environ["QTOS_ORDER"]
    assert (d0.n_stages, d0.front) == (192, 112)
    d2, _ = capi.analyze(PlannerConfig.knots100(duration=10.0, reduce_base=False))
    assert d2.front > 128


@pytest.mark.parametrize("which", ["knots100", "trot", "knots200", "reference_compat", "duration12", "full_system"])
def test_sweep_schedule_of_the_slack_steps(which):
    """The schedule by which the idle waves of the backward sweep form ds = Ji dx (qtos_planner.hip build_sweep_tasks; the
    reference has no counterpart: Ipopt forms J d inside its line search).  Host-only invariants: every inequality row of
    the working set exactly once; a row runs no earlier than the step behind the one that solves the stage of its earliest
    column -- every column of the row is then known, positions never decrease along the sweep --; the 16-bit copy of the
    column positions agrees with the list (checked inside the call)."""
    import dataclasses
    from qtos_amd import capi
    from qtos_amd.config import PlannerConfig
    cfg = {"knots100": PlannerConfig.knots100(), "trot": PlannerConfig.knots100(gait="trot"), "knots200": PlannerConfig.knots200(),
           "reference_compat": PlannerConfig.reference_compat(), "duration12": PlannerConfig.reference_compat(reduce_base=True, duration=12.0),
           "full_system": dataclasses.replace(PlannerConfig.knots100(), reduce_base=False)}[which]
    d, _ = capi.analyze(cfg)
    rows, entries, pos_min, pos_max = capi.analyze_sweep(cfg)
    live = rows >= 0
    assert live.sum() > 0 and len(np.unique(rows[live])) == live.sum()          # each row once
    assert rows.shape[0] >= d.n_stages                                        # a round per step of the chain at least
    step = np.repeat(np.arange(rows.shape[0])[:, None], 16, axis=1)
    stage_of_first_col = pos_min // 16
    assert (step[live] >= d.n_stages - stage_of_first_col[live]).all()        # behind the step that solves that stage
    assert (pos_max[live] < 16 * d.n_stages).all() and (entries[live] >= 1).all()
    assert not live[0].any()                                                  # nothing is known in the first step
    # rows of one round belong to one stage, and the stages come in the order the sweep meets them
    st = np.where(live, stage_of_first_col, -1).max(axis=1)
    assert all((stage_of_first_col[i][live[i]] == st[i]).all() for i in range(rows.shape[0]))
    seq = st[st >= 0]
    assert (np.diff(seq) <= 0).all()


def test_kronecker_structure_of_the_range_of_motion_blocks():
    """Every column of a range-of-motion block's Jacobian is a static multiple of a column of R(theta)^T or of
    d(R^T (p - r))/d theta (towr range_of_motion_constraint.cc; Symbolic::kron_meta): on the 100-knot walk 252 of the 320
    inequality blocks (the others are the friction pyramids and terrain rows), at most 8 in a stage record; every entry of
    G' S G and G' w through the 33 sums of a block equals the direct three-term sum to rounding (random matrices and weights)."""
    from qtos_amd import capi
    from qtos_amd.config import PlannerConfig
    # (the structure as found with every swing row in the system: reduce_swing folds the mid nodes' columns onto the footholds,
    #  whose weights then differ per dimension -- the experiment's analysis does not cover that)
    n_blocks, n_kron, most, worst = capi.analyze_kron(PlannerConfig.knots100(reduce_swing=False))
    assert (n_blocks, n_kron, most) == (320, 252, 8) and worst < 1e-11
    n_blocks, n_kron, most, worst = capi.analyze_kron(PlannerConfig.knots100(gait="trot", reduce_swing=False))
    assert 0 < n_kron < n_blocks and worst < 1e-11


def test_bad_parameters_are_rejected(hip_lib, cfg):
    from qtos_amd import capi
    p = capi.params_from_config(cfg)
    p.n_phases[0] = 4  # even: does not end in stance
    d = capi.QtosDims()
    assert hip_lib.qtos_analyze(C.byref(p), C.byref(d), None, 0) == -1
    h = C.c_void_p()
    assert hip_lib.qtos_planner_create(C.byref(p), 1, 0, C.byref(h)) < 0 and not h


def test_random_height_equals_reference():
    """Terrain randomiser vs the reference's own `random_height_shift` outputs (python `random` seeded;
    fixture generated by tests/golden/make_golden.py from QTOS/generateHeightField.py:692-730)."""
    import json
    import random
    from qtos_amd import heightfield, workloads
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "random_height.json")))
    base = heightfield.build_map([workloads.tile(t) for t in d["tiles"]], d["mesh_scale"])
    assert np.array_equal(base, np.array(d["base"]))
    for c in d["cases"]:
        m = heightfield.random_height_shift(base, c["shift"], random.Random(c["seed"]))
        assert np.array_equal(m, np.array(c["map"]))
    assert np.array_equal(base, np.array(d["base"]))   # the input is not modified
