# round 6: the KKT solve's accuracy on short trot horizons under the order of rounds 1 - 5 (order_fuzz.log: 6.5e-3 / 1.2e-3):
# residual of one solve with barrier weights over six decades, the largest factor-panel entries per stage and the unknowns
# eliminated there, per library (QTOS_LIB) -- run with the product library and with an experiment build
import os, subprocess, sys
code = '''
import sys, os, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from oracle.oracle import Oracle, oracle_dict
gait, dur, dt = os.environ["CFG"].split(",")
cfg = PlannerConfig(gait=gait, duration=float(dur), dt_base=float(dt), dt_dynamic=float(dt))
O = Oracle(oracle_dict(cfg)); L_ = O.L
P = Planner(cfg, max_batch=8)
s, gl = workloads.flat_goals(8, seed=11)
gl[:, 0] = s[:, 0] + (gl[:, 0] - s[:, 0]) * (cfg.duration / 5.0)
nn, st, it, v = P.plan(s, gl)
x0 = P.initial_guess(s[:2], gl[:2])
rng = np.random.default_rng(0)
x = x0 + 0.01 * rng.standard_normal(x0.shape)
sig = 10.0 ** rng.uniform(-3, 3, (2, P.m)); w = rng.standard_normal((2, P.m)) * np.sqrt(sig)
P.debug_newton(s[:2], gl[:2], x, sig, w)
dx, res = P.debug_residual(2, refine=False)
dx2, res2 = P.debug_residual(2, refine=True)
pan, ps = P.factor(0)
rk, vf, order = P.structure()
n = P.n
def fam(u):
    if u < 0: return "dummy"
    if u < n:
        if u < L_.off_ang: return "lin%d" % u
        if u < L_.off_eem[0]: return "ang%d" % (u - L_.off_ang)
        for e in range(4):
            if L_.off_eem[e] <= u < L_.off_eem[e] + L_.n_eem[e]: return "p%d.%d" % (e, u - L_.off_eem[e])
        for e in range(4):
            if L_.off_eef[e] <= u < L_.off_eef[e] + L_.n_eef[e]: return "f%d.%d" % (e, u - L_.off_eef[e])
        return "var%d" % u
    return "s%d" % (u - n)
mx = np.abs(pan[:, 1:, :]).max(axis=(1, 2))
wmx = np.abs(pan[:, 0, :]).max(axis=1)
print("%-12s %-20s rule %d front %3d stages %3d %-16s residual %.1e refined %.1e  max |V| %.2e  converged %d / 8 in %d..%d" % (os.environ["QTOS_LIB"][8:-3], os.environ["CFG"], P.dims.order_rule, P.dims.front, P.dims.n_stages, P.kkt_kernel(), res.max(), res2.max(), mx.max(), int((st == 0).sum()), it.min(), it.max()))
if os.environ.get("VERB"):
    worst = np.argsort(-mx)[:5]
    for k in sorted(worst):
        ids = order[16 * k:16 * k + 16]
        print("   stage %3d max |V| %.2e  |w| %.2e  pivots:" % (k, mx[k], wmx[k]), " ".join(fam(int(u)) for u in ids))
    print("   first stages:")
    for k in range(4):
        ids = order[16 * k:16 * k + 16]
        print("   stage %3d max |V| %.2e  pivots:" % (k, mx[k]), " ".join(fam(int(u)) for u in ids))
P.close()
'''
libs = sys.argv[1:] or ["libqtos_planner.so"]
for c in ("trot,2.5,0.05", "trot,5.0,0.1", "trot,2.5,0.1", "trot,4.0,0.1", "trot,5.0,0.05", "walk,5.0,0.1", "walk,2.5,0.05", "walk,5.0,0.05"):
    for lib in libs:
        for order in ("", "1"):
            env = dict(os.environ, CFG=c, QTOS_LIB=lib)
            if order:
                env["QTOS_ORDER"] = order
                env.pop("VERB", None)
            subprocess.run([sys.executable, "-c", code], env=env)
