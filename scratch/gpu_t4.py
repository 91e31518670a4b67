import sys, json; sys.path.insert(0, '.')
import numpy as np
from qtos_amd.config import PlannerConfig
from qtos_amd.capi import Planner
from oracle.oracle import Oracle
cfg = PlannerConfig.reference_compat()
P = Planner(cfg, max_batch=2)
O = Oracle(cfg.oracle_dict())
gv = np.load('tests/golden/gv1.npz'); inp = json.loads(str(gv['inputs']))
start = np.concatenate([inp['s'], inp['s_ang'], np.ravel(inp['ee']), inp['s_vel'], inp['s_ang_vel']])[None]
goal = np.array(inp['g'])[None]
rng = np.random.default_rng(1)
x = gv['x'][None] + 0.01 * rng.standard_normal((1, P.n))
q = O.problem(inp['s'], inp['s_ang'], inp['ee'], inp['g'])
xl, xh = O.var_bounds(q); fx = xl == xh; x[:, fx] = xl[fx]
rk, vf, order = P.structure(); I = rk == 2
free = np.nonzero(~fx)[0]; E = np.nonzero(rk == 1)[0]; Ii = np.nonzero(I)[0]
nf, nE = len(free), len(E)
Jo, go = O.jacobian(x[0]), O.constraints(x[0])
JE, JI = Jo[np.ix_(E, free)], Jo[np.ix_(Ii, free)]
for name, sscale, wscale in (("eq only", 0.0, 0.0), ("sig only", 1.0, 0.0), ("w only", 0.0, 1.0), ("both", 1.0, 1.0)):
    sig = np.zeros((1, P.m)); w = np.zeros((1, P.m))
    sig[:, I] = sscale * rng.uniform(0.5, 2.0, I.sum()); w[:, I] = wscale * rng.standard_normal(I.sum())
    dx = P.debug_newton(start, goal, x, sig, w)[0]
    K = np.zeros((nf + nE, nf + nE))
    K[:nf, :nf] = cfg.delta_x * np.eye(nf) + JI.T @ (sig[0, Ii][:, None] * JI)
    K[nf:, :nf] = JE; K[:nf, nf:] = JE.T; K[nf:, nf:] = -cfg.eps_dual * np.eye(nE)
    rhs = np.concatenate([-JI.T @ w[0, Ii], -go[E]])
    ref = np.linalg.solve(K, rhs)[:nf]
    print(name, 'err', np.abs(dx[free] - ref).max(), 'scale', np.abs(ref).max())
import ctypes as C
P.lib.qtos_debug_stream_check.argtypes = [C.c_void_p]
print('stream mismatches', P.lib.qtos_debug_stream_check(P.h))
print('nan in dx', np.isnan(dx).sum(), 'first nan vars', np.nonzero(np.isnan(dx))[0][:10])
