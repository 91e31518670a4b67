import sys, itertools; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
B = 256
def run(kw):
    out = []
    P = capi.Planner(PlannerConfig.knots100(**kw), max_batch=B)
    for name in ("flat", "mixed", "exp5"):
        kk = ch = 0; its = []; ok = 0; tt = 0.0
        for seed in range(6):
            if name == "flat":
                t = workloads.exp1_terrain(); P.set_heightfields(t[0], t[1]); s, g = workloads.flat_goals(B, 1000 * seed); m = None
            elif name == "mixed":
                t = workloads.mixed_terrains(); P.set_heightfields(t[0], t[1]); s, g, m = workloads.mixed_goals(B, seed=2 + 1000 * seed, terrains=t)
            else:
                t = workloads.exp5_terrain(); P.set_heightfields(t[0], t[1]); s, g = workloads.step_goals(B, seed=1 + 1000 * seed, terrain=t); m = None
            r = P.plan(s, g, map_id=m); tm = P.timing()
            kk += tm["kkt_launches"]; ch += tm["chord_launches"]; its.append(np.mean(r[2])); ok += (r[1] == 0).sum(); tt += tm["total_seconds"]
        out.append("%s it=%.2f kkt=%d ch=%d ok=%d/%d ms/batch=%.2f" % (name, np.mean(its), kk, ch, ok, 6 * B, 1e3 * tt / 6))
    return " | ".join(out)
grid = [dict(), dict(mu_init=0.5, slack_push=0.5), dict(mu_init=0.5, slack_push=0.35), dict(mu_init=0.3, slack_push=0.5), dict(mu_init=1.0, slack_push=0.5), dict(mu_init=0.5, slack_push=0.7)]
for kw in grid:
    try: print(kw, run(kw), flush=True)
    except Exception as e: print(kw, "ERR", e)
