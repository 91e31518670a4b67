"""One-off stress check: GPU vs oracle over whole batches (iterations and nodes)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from oracle.oracle import Oracle, oracle_dict
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
def sweep(tag, cfg, start, goal, maps=None, cell=None, mid=None, n=int(__import__("os").environ.get("SWEEP_N", 96))):
    P = capi.Planner(cfg, max_batch=len(start))
    if maps is not None: P.set_heightfields(maps, cell)
    nodes, status, iters, viol = P.plan(start, goal, map_id=mid)
    P.close()
    same = diff_it = both_ok = 0; worst = 0.0
    orc = {}
    for b in range(n):
        m = 0 if mid is None else int(mid[b])
        if m not in orc:
            h = None if maps is None else (maps if maps.ndim == 2 else maps[m])
            orc[m] = Oracle(oracle_dict(cfg), height=h, hcell=cell if cell else 0.1)
        O = orc[m]; s = start[b]
        xo, info = O.solve(O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), goal[b], s[18:21], s[21:24]))
        if info.status == 0 and status[b] == 0: both_ok += 1
        if info.iters == iters[b] and info.status == status[b]:
            e = float(np.abs(nodes[b] - xo).max()); worst = max(worst, e); same += e < 1e-6
        else: diff_it += 1
    print("%-18s gpu converged %d/%d | first %d vs oracle: identical (same iterations, nodes < 1e-6) %d, different iteration count %d, both converged %d, worst node diff among same-iteration %.1e" % (tag, (status == 0).sum(), len(start), n, same, diff_it, both_ok, worst))
B = 256
s, g = workloads.flat_goals(B, 0)
sweep("flat knots100", PlannerConfig.knots100(), s, g)
sweep("flat ref_compat", PlannerConfig.reference_compat(), s, g)
sweep("flat trot knots100", PlannerConfig.knots100(gait="trot"), s, g)
sweep("flat trot compat", PlannerConfig.reference_compat(gait="trot"), s, g)
t = workloads.exp5_terrain(); s5, g5 = workloads.step_goals(B, seed=1, terrain=t)
sweep("exp5 knots100", PlannerConfig.knots100(), s5, g5, t[0], t[1])
maps, cell = workloads.mixed_terrains(); sm, gm, mid = workloads.mixed_goals(B, seed=2, terrains=(maps, cell))
sweep("mixed knots100", PlannerConfig.knots100(), sm, gm, maps, cell, mid)
maps, cell = workloads.random_terrains(); sr, gr, midr = workloads.mpc_goals(B, terrains=(maps, cell))
sweep("random knots200", PlannerConfig.knots200(), sr, gr, maps, cell, midr, n=48)
