import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle.oracle import Oracle, oracle_dict
from oracle.projection import project_nodes
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from qtos_amd.replan import ShiftedWindows
for sw in (True, False):
    cfg = PlannerConfig.knots200(reduce_swing=sw)
    maps, cell = workloads.random_terrains()
    P = Planner(cfg, max_batch=4)
    P.set_heightfields(maps, cell)
    start, goal, map_id = workloads.mpc_goals(4, seed=5, terrains=(maps, cell))
    W = ShiftedWindows(P, start, goal - start[:, 0:3], map_id, advance=2.5)
    oracles = [Oracle(oracle_dict(cfg), height=maps[m], hcell=cell) for m in map_id]
    var_free = P.structure()[1]
    for k in range(6):
        W.warm_mode = "shifted" if k >= 4 else "none"
        nodes, status = W.replan()
        torch.cuda.synchronize()
        st, gl = W.start.cpu().numpy(), W.goal.cpu().numpy()
        warm = project_nodes(W.warm.cpu().numpy(), oracles[0].L, var_free) if k >= 4 else [None] * 4
        it = W.iters.cpu().numpy()
        out = []
        for b in range(4):
            O = oracles[b]
            q = O.problem(st[b, 0:3], st[b, 3:6], st[b, 6:18].reshape(4, 3), gl[b])
            xo, info = O.solve(q, x0=warm[b])
            out.append((int(status[b]), info.status, int(it[b]), info.iters, float(np.abs(nodes[b].cpu().numpy() - xo).max())))
        print('swing', sw, 'replan', k, ' '.join('%d/%d it %d/%d %.1e' % o for o in out))
    P.close()
