# round 6: numerical check of candidate elimination orders (exploration library, QTOS_EXP_TF / TFD / TB / TG): residual of one KKT
# solve at the straight-line start with random barrier weights, convergence and iterations of 64 flat problems, per gait
import os, subprocess, sys
code = '''
import sys, os, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
tag = os.environ["TAG"]
for g in os.environ.get("GAITS", "walk trot").split():
    cfg = PlannerConfig.knots100(gait=g)
    try:
        P = Planner(cfg, max_batch=64)
    except Exception as e:
        print(tag, g, "no planner", e); continue
    s, gl = workloads.flat_goals(64, seed=5)
    n, st, it, v = P.plan(s, gl)
    t = P.timing()
    x0 = P.initial_guess(s[:4], gl[:4])
    rng = np.random.default_rng(0)
    sig = rng.uniform(0.1, 10.0, (4, P.m)); w = rng.standard_normal((4, P.m))
    P.debug_newton(s[:4], gl[:4], x0, sig, w)
    dx, res = P.debug_residual(4, refine=False)
    print("%-28s %s front %3d stages %3d kernel %-16s kkt %.4f ms/launch  residual %.2e  converged %d iters %d..%d" %
          (tag, g, P.dims.front, P.dims.n_stages, P.kkt_kernel(), 1e3 * t["kkt_seconds"] / max(t["kkt_launches"], 1), float(res.max()), int((st == 0).sum()), int(it.min()), int(it.max())))
    P.close()
'''
cands = [("product", {})] + [(" ".join("%s=%s" % (k[9:], v) for k, v in c.items()), c) for c in [
    dict(QTOS_EXP_GUARD="2", QTOS_EXP_NB="30", QTOS_EXP_TF="0.5", QTOS_EXP_TB="-1.0"),
    dict(QTOS_EXP_GUARD="2", QTOS_EXP_NB="48", QTOS_EXP_TF="0.5", QTOS_EXP_TB="-1.0"),
    dict(QTOS_EXP_GUARD="2", QTOS_EXP_NB="60", QTOS_EXP_TF="0.5", QTOS_EXP_TB="-1.0"),
    dict(QTOS_EXP_GUARD="2", QTOS_EXP_NB="48", QTOS_EXP_TF="0.5", QTOS_EXP_TB="-1.0", QTOS_EXP_TLOB="0.2"),
    dict(QTOS_EXP_GUARD="2", QTOS_EXP_NB="48", QTOS_EXP_TF="0.5", QTOS_EXP_TFD="0.75", QTOS_EXP_TB="-1.0"),
    dict(QTOS_EXP_GUARD="2", QTOS_EXP_NB="48", QTOS_EXP_TF="0.75", QTOS_EXP_TFD="0.5", QTOS_EXP_TB="-1.0")]]
for tag, c in cands:
    env = dict(os.environ, TAG=tag[:60], QTOS_LIB="libqtos_exptf.so" if c else "libqtos_planner.so", GAITS="walk", **c)
    subprocess.run([sys.executable, "-c", code], env=env)
