import sys, itertools; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
def run(kw, B=64):
    out = []
    P = capi.Planner(PlannerConfig.knots100(**kw), max_batch=B)
    t = workloads.exp1_terrain(); P.set_heightfields(t[0], t[1])
    s, g = workloads.flat_goals(B, 0)
    r = P.plan(s, g); tm = P.timing()
    T = np.asarray(P.trace(0))[:7, 0]
    out.append("flat it=%.2f max=%d ok=%d kkt=%d ch=%d  trace %s" % (np.mean(r[2]), r[2].max(), (r[1] == 0).sum(), tm["kkt_launches"], tm["chord_launches"], " ".join("%.1e" % v for v in T)))
    t = workloads.mixed_terrains(); P.set_heightfields(t[0], t[1])
    s, g, m = workloads.mixed_goals(B, seed=2, terrains=t)
    r = P.plan(s, g, map_id=m); tm = P.timing()
    out.append("mixed it=%.2f max=%d ok=%d kkt=%d ch=%d" % (np.mean(r[2]), r[2].max(), (r[1] == 0).sum(), tm["kkt_launches"], tm["chord_launches"]))
    t = workloads.exp5_terrain(); P.set_heightfields(t[0], t[1])
    s, g = workloads.step_goals(B, seed=1, terrain=t)
    r = P.plan(s, g); tm = P.timing()
    out.append("exp5 it=%.2f max=%d ok=%d kkt=%d ch=%d" % (np.mean(r[2]), r[2].max(), (r[1] == 0).sum(), tm["kkt_launches"], tm["chord_launches"]))
    return " | ".join(out)
grid = []
for mu in (0.1, 0.02, 0.5):
    for sp in (0.2, 0.35, 0.5):
        grid.append(dict(mu_init=mu, slack_push=sp))
for kw in grid:
    try: print(kw, run(kw), flush=True)
    except Exception as e: print(kw, "ERR", e)
