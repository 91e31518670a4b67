# round 6: where a candidate elimination order loses accuracy on the GPU: per stage the largest entry of the factor panel V, with the
# unknowns eliminated in the worst stages (exploration library, see r6_order2.py)
import os, subprocess, sys
code = '''
import sys, os, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from oracle.oracle import Oracle, oracle_dict
tag = os.environ["TAG"]
cfg = PlannerConfig.knots100(gait=os.environ.get("GAIT", "walk"))
O = Oracle(oracle_dict(cfg)); L_ = O.L
P = Planner(cfg, max_batch=4)
s, gl = workloads.flat_goals(4, seed=5)
x0 = P.initial_guess(s, gl)
rng = np.random.default_rng(0)
sig = rng.uniform(0.1, 10.0, (4, P.m)); w = rng.standard_normal((4, P.m))
P.debug_newton(s, gl, x0, sig, w)
dx, res = P.debug_residual(4, refine=False)
pan, ps = P.factor(0)
rk, vf, order = P.structure()
n = P.n
def fam(u):
    if u < 0: return "dummy"
    if u < n:
        if u < L_.off_ang: return "base-lin-node"
        if u < L_.off_eem[0]: return "base-ang-node"
        for e in range(4):
            if L_.off_eem[e] <= u < L_.off_eem[e] + L_.n_eem[e]: return "ee-motion"
        return "ee-force"
    nsol = n + 630 if cfg.reduce_base else n
    return "coef" if u < nsol else None
mx = np.abs(pan[:, 1:, :]).max(axis=(1, 2))
wmx = np.abs(pan[:, 0, :]).max(axis=1)
print(tag, "residual %.2e" % res.max(), "front", P.dims.front, "max |V| over stages %.2e" % mx.max())
worst = np.argsort(-mx)[:6]
for k in sorted(worst):
    ids = order[16 * k:16 * k + 16]
    print("   stage %3d max |V| %.2e  |w| %.2e  pivots:" % (k, mx[k], wmx[k]), [int(u) for u in ids])
P.close()
'''
for tag, c in [("product", None), ("tf0.5_tb-1_guard", (0.5, 0.5, -1.0, 0.0)), ("tf0_tb-1_guard", (0.0, 0.0, -1.0, 0.0)), ("tf0.5_tb0_guard", (0.5, 0.5, 0.0, 0.0))]:
    env = dict(os.environ, TAG=tag, QTOS_LIB="libqtos_planner.so" if c is None else "libqtos_exptf.so")
    if c is not None:
        env.update(QTOS_EXP_TF=str(c[0]), QTOS_EXP_TFD=str(c[1]), QTOS_EXP_TB=str(c[2]), QTOS_EXP_TG=str(c[3]), QTOS_EXP_GUARD="2", QTOS_EXP_NB="24")
    subprocess.run([sys.executable, "-c", code], env=env)
