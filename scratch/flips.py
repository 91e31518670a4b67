import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads, heightfield
from qtos_amd.config import PlannerConfig
terr = workloads.exp5_terrain()
start, goal = workloads.step_goals(256, seed=1, terrain=terr)
sel = [59, 70, 143, 109]
hxy, cell = terr
xs = {}
for mi in range(2, 12):
    cfg = PlannerConfig.knots100(max_iter=mi, stall_iters=0)
    P = capi.Planner(cfg, max_batch=len(sel)); P.set_heightfields(hxy, cell)
    nodes, status, iters, viol = P.plan(start[sel], goal[sel])
    xs[mi] = (nodes.copy(), viol.copy(), iters.copy())
    P.close()
# stance nodes: ee-motion layout per foot: off = 612 + 35 e (reference_compat) -- for knots100 find via dims
d = capi.analyze(PlannerConfig.knots100())[0]
nb = d.n_base_nodes
off0 = 2 * 6 * nb   # base lin + ang
print("ee-motion offset", off0)
for pi, b in enumerate(sel):
    print("problem", b)
    for mi in range(2, 12):
        nodes, viol, iters = xs[mi]
        if iters[pi] < mi: continue
        cells = []
        for e in range(4):
            o = off0 + 35 * e
            for s in range(1, 5):
                p = nodes[pi, o + 8 * s: o + 8 * s + 3]
                ix = int(np.floor((p[0] + 1.0) / cell + 0.5)); iy = int(np.floor((p[1] + 1.0) / cell + 0.5))
                h = float(heightfield.height_at(hxy, cell, p[0], p[1], mode=1))
                cells.append((ix, iy, round(h, 3), round(p[2] - h, 4)))
        bad = [(i, c) for i, c in enumerate(cells) if abs(c[3]) > 1e-3]
        print("  it", mi, "viol %.3e" % viol[pi], "off-terrain stance nodes:", bad)
