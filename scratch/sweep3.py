"""chord-step settings (chord_tol x chord_max) over the workloads: ms per batch, factor / chord launches, converged (batch 256 x 5 seeds)"""
import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
B = 256
def run(name, kw):
    if name == "knots200": cfg = PlannerConfig.knots200(**kw)
    elif name == "trot": cfg = PlannerConfig.knots100(gait="trot", **kw)
    elif name == "compat": cfg = PlannerConfig.reference_compat(**kw)
    else: cfg = PlannerConfig.knots100(**kw)
    P = capi.Planner(cfg, max_batch=B)
    kk = ch = ok = 0; tt = 0.0; its = []
    for seed in range(5):
        if name in ("flat", "trot", "knots200", "compat"):
            t = workloads.exp1_terrain(); P.set_heightfields(t[0], t[1]); s, g = workloads.flat_goals(B, 1000 * seed); m = None
        elif name == "mixed":
            t = workloads.mixed_terrains(); P.set_heightfields(t[0], t[1]); s, g, m = workloads.mixed_goals(B, seed=2 + 1000 * seed, terrains=t)
        else:
            t = workloads.exp5_terrain(); P.set_heightfields(t[0], t[1]); s, g = workloads.step_goals(B, seed=1 + 1000 * seed, terrain=t); m = None
        P.plan(s, g, map_id=m)
        r = P.plan(s, g, map_id=m); tm = P.timing()
        kk += tm["kkt_launches"]; ch += tm["chord_launches"]; its.append(np.mean(r[2])); ok += (r[1] == 0).sum(); tt += tm["total_seconds"]
    return "%s %.2f ms it %.2f kkt %d ch %d ok %d" % (name, 1e3 * tt / 5, np.mean(its), kk, ch, ok)
for ct in (1e-3, 2e-3, 4e-3, 1e-2):
    for cm in (2, 3):
        kw = dict(chord_tol=ct, chord_max=cm)
        print(kw, " | ".join(run(n, kw) for n in ("flat", "trot", "exp5", "mixed", "knots200", "compat")), flush=True)
