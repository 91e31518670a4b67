#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's committed artefacts.

Runs ONLY in the build container (reads /root/reference); the GPU box and the test-suite use the
committed outputs.  Inputs:
  GV1  /root/reference/test/data/traj/gait.csv               (5001 x 37, one full 5 s plan)
  GV2  /root/reference/data/traj/towr.csv rows 1254..6254    (second full plan, t0 = 3.756)
  GV3  /root/reference/data/traj/towr.csv rows 0..1253       (partial first plan)
  log  /root/reference/logs/towr_log.out                     (solver inputs, NLP dimensions)
Outputs: gv1.npz, gv2.npz (TOWR-ordered node vector fitted to the 1 kHz samples, the samples
themselves on a 10 ms grid + phase boundary rows, solver inputs), nlp_dims.json.

The node vector is obtained by *reading* base nodes off the CSV (rows k*100) and by linear
least-squares of the cubic-Hermite phase splines to the foot / force samples (the CSV holds neither
foot velocities nor force derivatives).  No reference source text is copied: only data.
"""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

# phase schedule (verified against the force columns at 1 ms resolution)
UNNORM = {
    0: [0.8, 0.3, 1.7, 0.3, 1.7, 0.3, 1.45, 0.51, 1.06],   # FL  (-e1)
    1: [1.8, 0.3, 1.7, 0.3, 1.7, 0.3, 1.21, 0.51, 0.30],   # FR  (-e2)
    2: [0.3, 0.3, 1.7, 0.3, 1.7, 0.3, 1.70, 0.38, 1.44],   # HL  (-e3)
    3: [1.3, 0.3, 1.7, 0.3, 1.7, 0.3, 1.33, 0.51, 0.68],   # HR  (-e4)
}
T_TOTAL = 5.0


def phase_durations():
    s = T_TOTAL / 8.12
    return [[d * s for d in UNNORM[e]] for e in range(4)]


def hermite_w(T, t, deriv=0):
    t = np.asarray(t, float)
    if deriv == 0:
        return np.stack([1 - 3 * t**2 / T**2 + 2 * t**3 / T**3, t - 2 * t**2 / T + t**3 / T**2,
                         3 * t**2 / T**2 - 2 * t**3 / T**3, -t**2 / T + t**3 / T**2], -1)
    raise NotImplementedError


def fit_plan(rows, first_row_forces_valid=True):
    """rows: (5001, 37) with local time k/1000.  Returns x (1040,) in TOWR variable order.

    GV2's first row (towr.csv row 1254) is the hand-over row of the PREVIOUS plan (the stitcher
    drops the new plan's first row, QTOS/combiner.py:131,305): its state columns equal the new
    plan's start state but its force columns belong to the old plan, so they are excluded."""
    assert rows.shape == (5001, 37)
    t = np.arange(5001) / 1000.0
    ph = phase_durations()
    x = np.zeros(1040)
    # base nodes: rows k*100
    for k in range(51):
        r = rows[k * 100]
        x[6 * k:6 * k + 3] = r[1:4]
        x[6 * k + 3:6 * k + 6] = r[19:22]
        x[306 + 6 * k:306 + 6 * k + 3] = r[4:7]
        x[306 + 6 * k + 3:306 + 6 * k + 6] = r[22:25]
    if not first_row_forces_valid:
        # the new plan itself starts at rest: towr.csv rows 1255.. ramp up from zero velocity, i.e.
        # the reference solver did not apply the s_vel / s_ang_vel flags (the flags carry no dash,
        # QTOS/utils.py:26; logs/towr_log.out:161-166 shows them mis-parsed)
        x[3:6] = 0.0
        x[306 + 3:306 + 6] = 0.0
    resid = {}
    for e in range(4):
        bounds = np.concatenate([[0.0], np.cumsum(ph[e])])
        pos = rows[:, 7 + 3 * e:10 + 3 * e]
        frc = rows[:, 25 + 3 * e:28 + 3 * e]
        off_m, off_f = 612 + 35 * e, 752 + 72 * e
        stance_pos = []
        for s in range(5):
            a, b = bounds[2 * s], bounds[2 * s + 1]
            sel = (t > a + 2e-3) & (t < b - 2e-3)
            p = np.median(pos[sel], axis=0)
            stance_pos.append(p)
            x[off_m + 8 * s:off_m + 8 * s + 3] = p
        # swing mid nodes: LSQ of (px,vx,py,vy,pz) with known end points and vz = 0
        worst = 0.0
        for s in range(4):
            a, b = bounds[2 * s + 1], bounds[2 * s + 2]
            Th = (b - a) / 2
            p0, p1 = stance_pos[s], stance_pos[s + 1]
            sel1 = (t >= a) & (t <= a + Th)
            sel2 = (t > a + Th) & (t <= b)
            for d in range(3):
                w1 = hermite_w(Th, t[sel1] - a)        # nodes: stance(p0,0) -> mid(pm,vm)
                w2 = hermite_w(Th, t[sel2] - a - Th)   # nodes: mid(pm,vm) -> stance(p1,0)
                A = np.concatenate([w1[:, 2:4], w2[:, 0:2]], 0)
                rhs = np.concatenate([pos[sel1, d] - w1[:, 0] * p0[d], pos[sel2, d] - w2[:, 2] * p1[d]])
                if d == 2:
                    A = A[:, :1]
                sol, *_ = np.linalg.lstsq(A, rhs, rcond=None)
                worst = max(worst, np.abs(A @ sol - rhs).max())
                base = off_m + 8 * s + 3
                if d < 2:
                    x[base + 2 * d], x[base + 2 * d + 1] = sol
                else:
                    x[base + 4] = sol[0]
        resid["ee%d_swing_fit" % e] = worst
        # force nodes: per stance 3 polys / 4 nodes; nodes adjacent to a swing are zero
        worst = 0.0
        c = 0
        for s in range(5):
            a, b = bounds[2 * s], bounds[2 * s + 1]
            Tp = (b - a) / 3
            free = [j for j in range(4) if not ((j == 0 and s > 0) or (j == 3 and s < 4))]
            for d in range(3):
                Arows, rhs = [], []
                for j in range(3):
                    lo, hi = a + j * Tp, a + (j + 1) * Tp
                    sel = (t >= lo - 1e-12) & (t <= hi + 1e-12) & (t > a + 1e-9) & (t < b - 1e-9)
                    if s == 0 and j == 0 and first_row_forces_valid:
                        sel |= (t == 0.0)
                    if s == 4 and j == 2:
                        sel |= (t == 5.0)
                    w = hermite_w(Tp, np.clip(t[sel] - lo, 0, Tp))
                    blk = np.zeros((sel.sum(), 8))
                    blk[:, 2 * j:2 * j + 2] = w[:, 0:2]
                    blk[:, 2 * j + 2:2 * j + 4] = w[:, 2:4]
                    Arows.append(blk)
                    rhs.append(frc[sel, d])
                A = np.concatenate(Arows, 0)
                rhs = np.concatenate(rhs)
                cols = [2 * j + q for j in free for q in range(2)]
                sol, *_ = np.linalg.lstsq(A[:, cols], rhs, rcond=None)
                worst = max(worst, np.abs(A[:, cols] @ sol - rhs).max())
                for jj, j in enumerate(free):
                    x[off_f + 6 * (c + jj) + 2 * d] = sol[2 * jj]
                    x[off_f + 6 * (c + jj) + 2 * d + 1] = sol[2 * jj + 1]
            c += len(free)
        assert c == 12
        resid["ee%d_force_fit" % e] = worst
    return x, resid


def parse_log(path):
    txt = open(path).read()
    dims = {
        "n_vars_free": int(re.search(r"Total number of variables\.+:\s+(\d+)", txt).group(1)),
        "n_eq": int(re.search(r"Total number of equality constraints\.+:\s+(\d+)", txt).group(1)),
        "n_ineq": int(re.search(r"Total number of inequality constraints\.+:\s+(\d+)", txt).group(1)),
        "ineq_lower_only": int(re.search(r"only lower bounds:\s+(\d+)\n\s+inequality constraints with lower and", txt).group(1)),
        "ineq_both": int(re.search(r"inequality constraints with lower and upper bounds:\s+(\d+)", txt).group(1)),
        "ineq_upper_only": int(re.findall(r"inequality constraints with only upper bounds:\s+(\d+)", txt)[0]),
        "jac_nnz_eq": int(re.search(r"equality constraint Jacobian\.+:\s+(\d+)", txt).group(1)),
        "jac_nnz_ineq": int(re.search(r"inequality constraint Jacobian\.:\s+(\d+)", txt).group(1)),
    }
    sets = re.findall(r"^\s+([a-z\-_0-9]+)\s+(\d+)\s+(\d+)\.+(\d+)", txt, re.M)
    seen, var_sets, con_sets = set(), [], []
    for name, c, a, b in sets:
        if name in seen:
            continue
        seen.add(name)
        (var_sets if name.startswith(("base", "ee-")) else con_sets).append([name, int(c), int(a), int(b)])
    dims["variable_sets"] = var_sets
    dims["constraint_sets"] = con_sets
    dims["inf_pr_iter0"] = [float(v) for v in re.findall(r"^\s+0\s+0\.0000000e\+00\s+(\S+)", txt, re.M)]
    dims["iterations"] = [int(v) for v in re.findall(r"Number of Iterations\.+:\s+(\d+)", txt)]
    return dims


def boundary_fixtures():
    """Outputs of the reference's own boundary code (imported here, never shipped): cmd_args strings,
    heightfield file text, and the tile -> map -> solver-map pipeline."""
    import hashlib
    import tempfile
    import types
    sys.modules.setdefault("pybullet", types.ModuleType("pybullet"))
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from QTOS import utils as rutils
        from QTOS import generateHeightField as ghf
        out = {"cmd_args": [], "heightfield": {}}
        arg_sets = [
            {"-g": [0.520000318742596, 3.541584398612406e-07, 0.24], "-s": [0, 0, 0.24],
             "-e1": [0.21, 0.19, 0.0], "-e2": [0.21, -0.19, 0.0], "-e3": [-0.21, 0.19, 0.0],
             "-e4": [-0.21, -0.19, 0.0], "-s_ang": [0, 0, 0], "-resolution": 0.01, "sim": object},
            {"-s": [0.335266, -0.0123145, 0.221551], "-g": [0.9100042764299588, 0.0, 0.24],
             "-s_ang": [-0.0422299, -0.0416417, 0.00880732], "-e1": [0.548758, 0.14327, 0],
             "-t": 3.756, "s_vel": [0.135043, -0.422901, -0.014325],
             "s_ang_vel": [-0.98238, -0.256968, 0.932169], "-resolution": 0.01, "-r": 5.0,
             "f_steps": 2500, "-duration": None},
        ]
        for a in arg_sets:
            printable = {k: v for k, v in a.items() if not isinstance(v, type)}
            out["cmd_args"].append({"args": printable, "string": rutils.cmd_args(a)})
        for name, maps, scale in (("exp_1", ["plane", "plane"], 1), ("exp_5", ["climb_2", "climb_1"], 11),
                                  ("exp_3", ["feasibility", "feasibility_1", "plane"], 1)):
            obj = object.__new__(ghf.Height_Map_Generator)
            ghf.Maps.__init__(obj, maps, 20, scale)
            towr = obj.towr_map_adjustment(np.transpose(obj.map.copy()), shift_z=0.0, shift_down_num=0)
            with tempfile.NamedTemporaryFile("r", suffix=".txt") as tf:
                obj.create_height_file(tf.name, towr)
                text = open(tf.name).read()
            out["heightfield"][name] = {
                "tiles": maps, "mesh_scale": scale, "map_shape": list(obj.map.shape),
                "towr_shape": list(towr.shape), "sha256": hashlib.sha256(text.encode()).hexdigest(),
                "n_chars": len(text), "head": text[:160], "text": text if len(text) < 20000 else None,
                "map_checksum": float(np.sum(obj.map * np.arange(obj.map.size).reshape(obj.map.shape))),
                "resolution": 1 / (obj.map.shape[0] / 2),
            }
        return out
    finally:
        os.chdir(cwd)


def planner_fixtures():
    """Outputs of the reference's global planner / stitcher (imported here, never shipped)."""
    import tempfile
    import types
    sys.modules.setdefault("pybullet", types.ModuleType("pybullet"))
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)   # the reference writes ./data/plots/... relative to the cwd
    os.makedirs("data/plots", exist_ok=True)
    try:
        import io
        import contextlib
        from QTOS import planner as rpl
        from QTOS import combiner as rcb
        import QTOS.config.global_cfg as gcfg
        out = {"path_solver": [], "global_planner": [], "stitch": []}
        wall = np.zeros((20, 40))
        wall[4:16, 18:21] = 0.5          # a wall across the straight line, passable on both sides
        for name, m, start, goal in (("flat", np.zeros((20, 40)), [0, 0, .24], [2.5, 0, .24]),
                                     ("wall", wall, [0, 0, .24], [2.5, 0, .24])):
            with contextlib.redirect_stdout(io.StringIO()):
                ps = rpl.PATH_Solver(m, start, goal, {"step_size": 1.0}, grid_res=0.1, visual=False)
            ts = [0.0, 1.3, 5.0, 12.5, 20.0, float(ps.predicted_t)]
            out["path_solver"].append({
                "name": name, "map": m.tolist(), "start": start, "goal": goal, "step_size": 1.0,
                "path": [list(map(int, c)) for c in ps.path], "predicted_t": float(ps.predicted_t),
                "t": ts, "x": [float(ps.spine_x_track(t)) for t in ts], "y": [float(ps.spine_y_track(t)) for t in ts]})
        # Global_Planner.update -> pop on the exp_1 map (SURVEY.md 8c cross-check values)
        gcfg.ROBOT_CFG.robot_goal = [2.5, 0, 0.24]
        sim = types.SimpleNamespace(height_map=np.zeros((20, 40)), bool_map=None)
        args = {"args": {"resolution": 0.1}, "sim": sim, "-s": [0, 0, 0.24], "step_size": 1.0}
        with contextlib.redirect_stdout(io.StringIO()):
            gp = rpl.Global_Planner(args, lookahead=3750)
        seq = []
        for t in (0.0, 2.5, 5.0):
            gp.update(t, [0.0, 0.0], [0.0, 0.0], np.zeros(3))
        while not gp.empty():
            s_, g_ = gp.pop()
            seq.append({"start": [float(v) for v in s_], "goal": [float(v) for v in g_]})
        out["global_planner"] = {"lookahead": 3750, "update_times": [0.0, 2.5, 5.0], "popped": seq}
        # Combiner._state / combine on the canned plan
        gait_path = os.path.join(REF, "test/data/traj/gait.csv")
        for last_t, look, cutoff in ((0.006, 3750, 2500), (1.25, 2750, 1200), (0.0, 100, 0)):
            cb = object.__new__(rcb.Combiner)
            cb.current_traj = gait_path
            cb.new_traj = os.path.join(tmp, "new.csv")
            cb.last_timestep, cb.lookahead, cb.lookahead_original = last_t, look, look
            cb.cutoff_idx, cb.next_traj_step = cutoff, 0
            cb.height_set = {0.0}
            with contextlib.redirect_stdout(io.StringIO()):
                st = cb._state()
            # a "new plan" = the canned plan shifted in time; combine old + new like the reference
            new = np.loadtxt(gait_path, delimiter=",")
            new[:, 0] += 100.0
            np.savetxt(cb.new_traj, new[:50], delimiter=",", fmt="%g")
            cb.combine()
            comb = np.loadtxt(cb.new_traj, delimiter=",")
            out["stitch"].append({
                "last_timestep": last_t, "lookahead": look, "cutoff_idx": cutoff,
                "state": {k: [float(x) for x in v] for k, v in st.items()},
                "lookahead_after": int(cb.lookahead), "next_traj_step": int(cb.next_traj_step),
                "combined_shape": list(comb.shape), "combined_t_first": float(comb[0, 0]),
                "combined_t_last_old": float(comb[comb[:, 0] < 50][-1, 0]),
                "combined_t_first_new": float(comb[comb[:, 0] > 50][0, 0]),
                "combined_sum": float(comb.sum())})
        # PATH_MAP: patch enumeration and failure-stamp neighbourhoods on the exp_3 map
        os.chdir(REF)
        from QTOS import generateHeightField as ghf
        obj = object.__new__(ghf.Height_Map_Generator)
        ghf.Maps.__init__(obj, ["feasibility", "feasibility_1", "plane"], 20, 1)
        pm = object.__new__(ghf.PATH_MAP)
        pm.origin_shift_x = pm.origin_shift_y = 1.0

        class _Q:
            def __init__(self):
                self.items = []

            def put(self, d):
                self.items.append(d)
        pm.data_queue = _Q()
        pm.probe_map(obj.map, 3, 0.1)
        nb = ((-3, 0), (3, 0), (0, -3), (0, 3))
        out["path_map"] = {
            "tiles": ["feasibility_test", "feasibility_test_1", "plane"], "multi_map_shift": 3,
            "patches": [[list(map(float, d.map_coords_start)), list(map(float, d.map_coords_goal)),
                         list(map(int, d.map_idx_start)), list(map(int, d.map_idx_goal))]
                        for d in pm.data_queue.items],
            "hull": pm.find_convex_hull(nb).tolist(),
        }
        os.chdir(tmp)
        return out
    finally:
        os.chdir(cwd)


def random_height_fixtures():
    """Outputs of the reference's terrain randomiser (`Height_Map_Generator.random_height_shift`,
    QTOS/generateHeightField.py:692-730) on the exp_5 map at mesh_scale 1, python `random` seeded."""
    import random
    import types
    sys.modules.setdefault("pybullet", types.ModuleType("pybullet"))
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from QTOS import generateHeightField as ghf
        obj = object.__new__(ghf.Height_Map_Generator)
        ghf.Maps.__init__(obj, ["climb_2", "climb_1"], 20, 1)
        base = np.array(obj.map, dtype=float)
        out = {"tiles": ["climb_2", "climb_1"], "mesh_scale": 1, "base": base.tolist(), "cases": []}
        for seed, shift in ((0, 1), (1, 10), (2, 10)):
            random.seed(seed)
            m = obj.random_height_shift(base.copy(), shift)
            out["cases"].append({"seed": seed, "shift": shift, "map": np.asarray(m).tolist()})
        return out
    finally:
        os.chdir(cwd)


def main():
    gait = np.loadtxt(os.path.join(REF, "test/data/traj/gait.csv"), delimiter=",")
    towr = np.loadtxt(os.path.join(REF, "data/traj/towr.csv"), delimiter=",")
    gv2 = towr[1254:6255].copy()
    gv3 = towr[:1254].copy()
    keep = sorted(set(range(0, 5001, 10)))
    nominal_feet = [[0.21, 0.19, 0.0], [0.21, -0.19, 0.0], [-0.21, 0.19, 0.0], [-0.21, -0.19, 0.0]]
    inputs = {
        "gv1": dict(s=[0, 0, 0.24], s_ang=[0, 0, 0], ee=nominal_feet,
                    g=[float(gait[-1, 1]), 0.0, 0.24], s_vel=[0, 0, 0], s_ang_vel=[0, 0, 0], t0=0.0),
        # logs/towr_log.out:140-166 (flags), velocities = towr.csv row 1254 cols 19-24
        "gv2": dict(s=[0.335266, -0.0123145, 0.221551], s_ang=[-0.0422299, -0.0416417, 0.00880732],
                    ee=[[0.548758, 0.14327, 0], [0.606166, -0.175889, 0], [0.105142, 0.175699, 0],
                        [0.141849, -0.176012, 0]],
                    g=[0.9100042764299588, 0.0, 0.24], s_vel=[0, 0, 0], s_ang_vel=[0, 0, 0], t0=3.756,
                    s_vel_flag=gv2[0, 19:22].tolist(), s_ang_vel_flag=gv2[0, 22:25].tolist()),
    }
    for name, rows in (("gv1", gait), ("gv2", gv2)):
        x, resid = fit_plan(rows, first_row_forces_valid=(name == "gv1"))
        print(name, {k: "%.2e" % v for k, v in resid.items()})
        np.savez_compressed(os.path.join(OUT, name + ".npz"), x=x, row_idx=np.array(keep),
                            rows=rows[keep], inputs=json.dumps(inputs[name]),
                            phase_durations=np.array(phase_durations()))
    np.savez_compressed(os.path.join(OUT, "gv3_partial.npz"), rows=gv3[::10], row_idx=np.arange(0, 1254, 10))
    json.dump(boundary_fixtures(), open(os.path.join(OUT, "boundary.json"), "w"), indent=1)
    json.dump(planner_fixtures(), open(os.path.join(OUT, "planner.json"), "w"))
    json.dump(random_height_fixtures(), open(os.path.join(OUT, "random_height.json"), "w"))
    dims = parse_log(os.path.join(REF, "logs/towr_log.out"))
    json.dump(dims, open(os.path.join(OUT, "nlp_dims.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in dims.items() if not k.endswith("_sets")}))


if __name__ == "__main__":
    sys.exit(main())
