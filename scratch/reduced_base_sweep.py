import sys, dataclasses; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
rng = np.random.default_rng(3)
B = 4
cases = [("T5 ref", PlannerConfig.reference_compat()), ("T2.5", PlannerConfig.reference_compat(duration=2.5)), ("T2.5 dt0.2", PlannerConfig.reference_compat(duration=2.5, dt_dynamic=0.2, dt_base=0.2)),
         ("T12", PlannerConfig.reference_compat(duration=12.0)), ("trot100", PlannerConfig.knots100(gait="trot")), ("T8", PlannerConfig.reference_compat(duration=8.0))]
for name, base in cases:
    s, g = workloads.flat_goals(B, seed=11)
    g[:, 0] = s[:, 0] + (g[:, 0] - s[:, 0]) * (base.duration / 5.0 if base.duration <= 8.0 else 1.0)
    out = {}
    sig = w = None
    for rb in (False, True):
        P = capi.Planner(dataclasses.replace(base, reduce_base=rb), max_batch=B)
        x0 = P.initial_guess(s, g)
        rk, _, _ = P.structure()
        I = rk == 2
        if sig is None:
            sig = np.zeros((B, P.m)); w = np.zeros((B, P.m))
            sig[:, I] = 10.0 ** rng.uniform(-2, 2, (B, I.sum())); w[:, I] = rng.standard_normal((B, I.sum()))
        dx = P.debug_newton(s, g, x0, sig, w)
        _, res = P.debug_residual(B, refine=False)
        dxr, res1 = P.debug_residual(B, refine=True)
        nodes, status, iters, viol = P.plan(s, g)
        out[rb] = (dxr, nodes, iters, res.max(), res1.max(), P.dims.n_unknowns, P.dims.front)
        P.close()
    d = np.abs(out[True][0] - out[False][0]).max() / np.abs(out[False][0]).max()
    print("%-10s unknowns %d/%d front %d/%d | first solve residual full %.1e reduced %.1e (refined %.1e) | step reduced vs full (rel) %.1e | iters full %s reduced %s | nodes diff %.2e" %
          (name, out[False][5], out[True][5], out[False][6], out[True][6], out[False][3], out[True][3], out[True][4], d, out[False][2].tolist(), out[True][2].tolist(), np.abs(out[True][1] - out[False][1]).max()))
