"""two-phase solve settings (foothold_hold_from x foothold_hold_tol) on the terrain workloads: ms per batch, launches, converged (batch 256 x 5 seeds)"""
import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
B = 256
NSEED = 8
def run(name, kw):
    P = capi.Planner((PlannerConfig.knots200 if name == "mpc" else PlannerConfig.knots100)(**kw), max_batch=B)
    kk = ch = ok = 0; tt = 0.0; its = []; mx = []
    for seed in range(NSEED):
        if name == "mpc":
            t = workloads.random_terrains(); P.set_heightfields(t[0], t[1]); s, g, m = workloads.mpc_goals(B, seed=5 + 1000 * seed, terrains=t)
        elif name == "mixed":
            t = workloads.mixed_terrains(); P.set_heightfields(t[0], t[1]); s, g, m = workloads.mixed_goals(B, seed=2 + 1000 * seed, terrains=t)
        else:
            t = workloads.exp5_terrain(); P.set_heightfields(t[0], t[1]); s, g = workloads.step_goals(B, seed=1 + 1000 * seed, terrain=t); m = None
        P.plan(s, g, map_id=m)
        r = P.plan(s, g, map_id=m); tm = P.timing()
        kk += tm["kkt_launches"]; ch += tm["chord_launches"]; its.append(np.mean(r[2])); mx.append(r[2].max()); ok += (r[1] == 0).sum(); tt += tm["total_seconds"]
    return "%s %.2f ms it %.2f max %s kkt %d ch %d ok %d" % (name, 1e3 * tt / NSEED, np.mean(its), mx, kk, ch, ok)
NSEED = 8
for hf in (2, 1):
    for ht in (0.25, 3.0, 10.0, 1e9):
        kw = dict(foothold_hold_from=hf, foothold_hold_tol=ht)
        print(kw, " | ".join(run(n, kw) for n in ("exp5", "mixed", "mpc")), flush=True)
