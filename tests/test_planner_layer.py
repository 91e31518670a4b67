"""the stitcher and the global planner against outputs of the
reference's own code (fixtures made by tests/golden/make_golden.py importing /root/reference)."""
import json
import os

import numpy as np

from conftest import GOLDEN, load_gv

FIX = json.load(open(os.path.join(GOLDEN, "planner.json")))


def test_path_solver_matches_reference():
    from qtos_amd.global_planner import PathSolver
    for case in FIX["path_solver"]:
        ps = PathSolver(np.array(case["map"]), case["start"], case["goal"], case["step_size"], 0.1)
        assert [list(c) for c in ps.path] == case["path"]
        assert ps.predicted_t == case["predicted_t"]
        for t, x, y in zip(case["t"], case["x"], case["y"]):
            assert abs(float(ps.spine_x_track(t)) - x) < 1e-14 and abs(float(ps.spine_y_track(t)) - y) < 1e-14
    # the values the solver log carries as -g of solve #1 (logs/towr_log.out:8-10)
    flat = FIX["path_solver"][0]
    assert flat["x"][2] == 0.520000318742596


def test_global_planner_pairs_match_reference():
    from qtos_amd.global_planner import GlobalPlanner
    ref = FIX["global_planner"]
    gp = GlobalPlanner(np.zeros((20, 40)), [0, 0, 0.24], [2.5, 0, 0.24], step_size=1.0, resolution=0.1,
                       lookahead=ref["lookahead"])
    for t in ref["update_times"]:
        gp.update(t)
    popped = []
    while not gp.empty():
        s, g = gp.pop()
        popped.append((s, g))
    assert len(popped) == len(ref["popped"])
    for (s, g), r in zip(popped, ref["popped"]):     # LIFO order
        assert np.abs(np.array(s) - r["start"]).max() < 1e-14
        assert np.abs(np.array(g) - r["goal"]).max() < 1e-14
    # logged -g of solve #2 (logs/towr_log.out:140) is the goal of the pair pushed at t = 0
    assert abs(popped[-1][1][0] - 0.9100042764299588) < 1e-15


def _gait_rows():
    gv = load_gv("gv1")
    # the fixture keeps every 10th row; rebuild the 1 kHz time base for index arithmetic
    rows = np.zeros((5001, 37))
    rows[:, 0] = np.round(np.arange(5001) / 1000.0, 3)
    rows[gv["row_idx"]] = gv["rows"]
    return rows, gv


def test_stitcher_indices_match_reference():
    """Row selection of Combiner._state and the splice of Combiner.combine."""
    from qtos_amd.stitcher import Stitcher
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd.config import PlannerConfig
    gv = load_gv("gv1")
    O = Oracle(oracle_dict(PlannerConfig.reference_compat()))
    rows = O.sample(gv["x"], 0.0)                    # the full 1 kHz plan (matches gait.csv to 1e-6)
    rows = np.array([[float("%g" % v) for v in r] for r in rows])   # as printed in the CSV
    for case in FIX["stitch"]:
        st = Stitcher(lookahead=case["lookahead"], height_set=(0.0,))
        st.cutoff_idx = case["cutoff_idx"]
        state = st.state(rows, case["last_timestep"])
        assert st.lookahead == case["lookahead_after"] and st.next_traj_step == case["next_traj_step"]
        for k, v in case["state"].items():
            assert np.abs(np.array(state[k]) - np.array(v)).max() < 2e-4   # CSV print precision
        new = rows.copy()
        new[:, 0] += 100.0
        comb = st.combine(rows, new[:50])
        assert list(comb.shape) == case["combined_shape"]
        assert abs(comb[0, 0] - case["combined_t_first"]) < 1e-9
        assert abs(comb[comb[:, 0] < 50][-1, 0] - case["combined_t_last_old"]) < 1e-9
        assert abs(comb[comb[:, 0] > 50][0, 0] - case["combined_t_first_new"]) < 1e-9
        clean = Stitcher(lookahead=case["lookahead"], mode="clean")
        clean.cutoff_idx, clean.next_traj_step = case["cutoff_idx"], case["next_traj_step"]
        assert len(clean.combine(rows, new[:50])) == len(comb) + 1 or case["cutoff_idx"] == 0


def test_plan_args_fill_the_solver_flags():
    from qtos_amd import flags
    from qtos_amd.stitcher import Stitcher, row_state
    gv = load_gv("gv2")
    st = Stitcher(lookahead=3750)
    state = row_state(gv["rows"][0])
    args = st.plan_args({"-resolution": 0.01}, state, runtime=0.006, goal=[0.91, 0.0, 0.24])
    assert abs(args["-t"] - 3.756) < 1e-12
    start, goal, t0 = flags.problem_arrays(args)
    assert np.allclose(start[0:3], gv["inputs"]["s"], atol=1e-6)
    assert np.allclose(start[18:21], gv["inputs"]["s_vel_flag"], atol=1e-6)


def test_feasibility_probe_and_stamp_match_reference():
    """patch enumeration + failure neighbourhood of PATH_MAP."""
    from qtos_amd import feasibility, heightfield
    ref = FIX["path_map"]
    tiles = [heightfield.read_tile(os.path.join(GOLDEN, "heightfields", t + ".txt")) for t in ref["tiles"]]
    m = heightfield.build_map(tiles, 1)
    patches = feasibility.probe_patches(m, ref["multi_map_shift"], 0.1)
    assert len(patches) == len(ref["patches"])
    for p, r in zip(patches, ref["patches"]):
        assert list(p[0]) == r[0] and list(p[1]) == r[1] and list(p[2]) == r[2] and list(p[3]) == r[3]
    assert sorted(map(list, feasibility.diamond(1))) == sorted(ref["hull"])
    a = feasibility.patch_args(patches[0][0], patches[0][1])
    assert a["-s"][2] == patches[0][0][2] + 0.24 and a["-r"] == 5.0 and a["-e4"][0] == patches[0][0][0] - 0.21
    bm = feasibility.stamp(m.shape, patches, [0 if i % 3 else 1 for i in range(len(patches))])
    assert bm.shape == m.shape and set(np.unique(bm)) <= {0, 1} and bm.sum() > 0
    s_idx = patches[0][2]
    assert bm[s_idx[0], s_idx[1]] == 1 or bm[tuple(patches[0][3])] == 1
