# round 6: the shifted replans of the receding windows under both order rules: status / iterations / violation per window, GPU and oracle
import os, sys, subprocess
code = '''
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from oracle.oracle import Oracle, oracle_dict, oracle_options
from oracle.projection import project_nodes
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from qtos_amd.replan import ShiftedWindows
for preset in ("receding_windows", "knots200"):
    cfg = getattr(PlannerConfig, preset)()
    maps, cell = workloads.random_terrains()
    P = Planner(cfg, max_batch=4); P.set_heightfields(maps, cell)
    start, goal, map_id = workloads.mpc_goals(4, seed=5, terrains=(maps, cell))
    W = ShiftedWindows(P, start, goal - start[:, 0:3], map_id, advance=2.5)
    oracles = [Oracle(oracle_dict(cfg), height=maps[m], hcell=cell) for m in map_id]
    var_free = P.structure()[1]
    for k in range(6):
        W.warm_mode = "shifted" if k >= 4 else "none"
        nodes, status = W.replan(); torch.cuda.synchronize()
        if k < 4: continue
        st, gl = W.start.cpu().numpy(), W.goal.cpu().numpy()
        warm = project_nodes(W.warm.cpu().numpy(), oracles[0].L, var_free)
        it, vi = W.iters.cpu().numpy(), W.viol.cpu().numpy()
        for b in range(4):
            O = oracles[b]
            q = O.problem(st[b, 0:3], st[b, 3:6], st[b, 6:18].reshape(4, 3), gl[b])
            xo, info = O.solve(q, x0=warm[b], opts=oracle_options(cfg, O, match_eliminated=True))
            tr = P.trace(b)
            print("order %s %s replan %d window %d: gpu status %d iters %d viol %.2e | oracle status %d iters %d inf_pr %.2e | gap %.2e | gpu viol trace %s" %
                  (os.environ.get("QTOS_ORDER", "auto"), preset, k, b, int(status[b]), int(it[b]), vi[b], info.status, info.iters, info.inf_pr, np.abs(nodes[b].cpu().numpy() - xo).max(),
                   " ".join("%.1e" % v for v in tr[:, 0])))
    P.close()
'''
for o in (None, "0"):
    env = dict(os.environ)
    if o is not None: env["QTOS_ORDER"] = o
    else: env.pop("QTOS_ORDER", None)
    subprocess.run([sys.executable, "-c", code], env=env)
