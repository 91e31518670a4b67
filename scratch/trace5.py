import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots100()
P = capi.Planner(cfg, max_batch=256)
terr = workloads.exp5_terrain()
P.set_heightfields(terr[0][None], terr[1])
start, goal = workloads.step_goals(256, 1, terr)
nodes, status, iters, viol = P.plan(start, goal)
print("converged", (status == 0).sum(), "iters hist", np.bincount(iters))
np.set_printoptions(linewidth=220, precision=2)
for b in np.nonzero(status != 0)[0][:6]:
    t = P.trace(b)
    print("problem", b, "viol:", t[:, 0]); print("   alpha:", t[:, 2])
