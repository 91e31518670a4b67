import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots200(honor_start_velocity=True)
B = 64
P = capi.Planner(cfg, max_batch=B)
maps, cell = workloads.random_terrains(); P.set_heightfields(maps, cell)
start, goal, mid = workloads.mpc_goals(B, terrains=(maps, cell))
nodes, status, iters, viol = P.plan(start, goal, map_id=mid)
print(0, np.bincount(status, minlength=3), iters.mean())
for k in range(1, 70):
    row = P.sample(nodes, 0.0, hz=50.0, n_rows=2)[:, 1]
    start = row[:, 1:25].copy()
    nodes2, status, iters, viol = P.plan(start, goal, map_id=mid, warm=nodes)
    nodes = nodes2
    if k % 3 == 0 or (status != 0).any():
        print(k, np.bincount(status, minlength=3), "it %.2f" % iters.mean(), "viol %.1e" % np.nanmax(viol), "x0 %.3f vx %.3f z %.3f nan %d" % (start[0, 0], start[0, 18], start[0, 2], np.isnan(nodes).sum()), "feet z", start[0, 8::3][:4].round(4))
