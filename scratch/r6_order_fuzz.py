# round 6: the elimination order the planner keeps (QtosDims.order_rule) over a spread of transcriptions -- horizons, knot spacings,
# gaits, reductions -- with the accuracy of one KKT solve (barrier weights over six decades) and a small cold batch per configuration:
# no configuration may pick an order that breaks the unpivoted factorisation
import sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qtos_amd import workloads
from qtos_amd.capi import Planner, analyze
from qtos_amd.config import PlannerConfig
cfgs = []
for gait in ("walk", "trot"):
    for dur in (2.5, 4.0, 5.0, 8.0, 10.0, 12.0, 20.0):
        for dt in (0.1, 0.05):
            if dt == 0.05 and dur > 10.0:
                continue
            cfgs.append(("%s %4.1f s dt %.2f" % (gait, dur, dt), PlannerConfig(gait=gait, duration=dur, dt_base=dt, dt_dynamic=dt)))
cfgs.append(("knots200", PlannerConfig.knots200()))
cfgs.append(("knots100 full swings", PlannerConfig.knots100(reduce_swing=False)))
worst = 0.0
for name, cfg in cfgs:
    try:
        d, _ = analyze(cfg)
        P = Planner(cfg, max_batch=8)
    except Exception as e:
        print("%-26s no planner: %s" % (name, str(e)[:60])); continue
    s, gl = workloads.flat_goals(8, seed=11)
    gl[:, 0] = s[:, 0] + (gl[:, 0] - s[:, 0]) * (cfg.duration / 5.0 if cfg.duration <= 8.0 else 1.0)
    n, st, it, v = P.plan(s, gl)
    x0 = P.initial_guess(s[:2], gl[:2])
    rng = np.random.default_rng(0)
    x = x0 + 0.01 * rng.standard_normal(x0.shape)
    sig = 10.0 ** rng.uniform(-3, 3, (2, P.m)); w = rng.standard_normal((2, P.m)) * np.sqrt(sig)
    P.debug_newton(s[:2], gl[:2], x, sig, w)
    dx, res = P.debug_residual(2, refine=False)
    worst = max(worst, float(res.max()))
    print("%-26s rule %d front %3d stages %3d kernel %-15s residual %.1e  converged %d / 8 in %d..%d iterations" %
          (name, P.dims.order_rule, P.dims.front, P.dims.n_stages, P.kkt_kernel(), res.max(), int((st == 0).sum()), it.min(), it.max()))
    P.close()
print("worst residual %.1e" % worst)
