# round 6: accuracy of one KKT solve under the two order rules (QTOS_ORDER=0 | 1) with barrier weights spanning many decades
import os, subprocess, sys
code = '''
import sys, os, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
for name, cfg in (("walk", PlannerConfig.knots100()), ("knots200", PlannerConfig.knots200()), ("trot", PlannerConfig.knots100(gait="trot"))):
    P = Planner(cfg, max_batch=4)
    s, gl = workloads.flat_goals(4, seed=5)
    x0 = P.initial_guess(s, gl)
    rng = np.random.default_rng(0)
    x = x0 + 0.01 * rng.standard_normal(x0.shape)
    out = []
    for lo, hi in ((-1, 1), (-3, 3), (-3, 6)):
        sig = 10.0 ** rng.uniform(lo, hi, (4, P.m)); w = rng.standard_normal((4, P.m)) * np.sqrt(sig)
        P.debug_newton(s, gl, x, sig, w)
        dx, res = P.debug_residual(4, refine=False)
        dx2, res2 = P.debug_residual(4, refine=True)
        out.append("sig 1e%d..1e%d: %.1e (refined %.1e)" % (lo, hi, res.max(), res2.max()))
    print("QTOS_ORDER=%s %-9s front %3d %s" % (os.environ.get("QTOS_ORDER"), name, P.dims.front, "  ".join(out)))
    P.close()
'''
for o in ("0", "1"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QTOS_ORDER=o))
