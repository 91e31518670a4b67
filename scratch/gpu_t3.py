import sys, json; sys.path.insert(0, '.')
import numpy as np
from qtos_amd.config import PlannerConfig
from qtos_amd.capi import Planner
from oracle.oracle import Oracle
cfg = PlannerConfig.reference_compat()
P = Planner(cfg, max_batch=8)
O = Oracle(cfg.oracle_dict())
gv = np.load('tests/golden/gv1.npz'); inp = json.loads(str(gv['inputs']))
start = np.concatenate([inp['s'], inp['s_ang'], np.ravel(inp['ee']), inp['s_vel'], inp['s_ang_vel']])[None]
goal = np.array(inp['g'])[None]
rng = np.random.default_rng(1)
B=4
x = gv['x'][None] + 0.01 * rng.standard_normal((B, P.n))
q = O.problem(inp['s'], inp['s_ang'], inp['ee'], inp['g'])
xl, xh = O.var_bounds(q); fx = xl == xh; x[:, fx] = xl[fx]
rk, vf, order = P.structure(); I = rk == 2
sig = np.zeros((B, P.m)); w = np.zeros((B, P.m))
sig[:, I] = 10.0 ** rng.uniform(-3, 3, (B, I.sum())); w[:, I] = rng.standard_normal((B, I.sum()))
st = np.repeat(start, B, 0); gl = np.repeat(goal, B, 0)
dx1 = P.debug_newton(st, gl, x, sig, w)
dx2 = P.debug_newton(st, gl, x, sig, w)
print('run-to-run identical', np.array_equal(dx1, dx2), np.abs(dx1-dx2).max())
free = np.nonzero(~fx)[0]; E = np.nonzero(rk == 1)[0]; Ii = np.nonzero(I)[0]
nf, nE = len(free), len(E)
for b in range(B):
    Jo, go = O.jacobian(x[b]), O.constraints(x[b])
    JE, JI = Jo[np.ix_(E, free)], Jo[np.ix_(Ii, free)]
    K = np.zeros((nf + nE, nf + nE))
    K[:nf, :nf] = cfg.delta_x * np.eye(nf) + JI.T @ (sig[b, Ii][:, None] * JI)
    K[nf:, :nf] = JE; K[:nf, nf:] = JE.T; K[nf:, nf:] = -cfg.eps_dual * np.eye(nE)
    rhs = np.concatenate([-JI.T @ w[b, Ii], -go[E]])
    ref = np.linalg.solve(K, rhs)[:nf]
    import ctypes as C
    from oracle.oracle import lib as olib
    sol = rhs.copy(); Kc = np.ascontiguousarray(K)
    olib().qo_ldlt_solve_dense(nf + nE, Kc.ctypes.data_as(C.POINTER(C.c_double)), sol.ctypes.data_as(C.POINTER(C.c_double)))
    print('   cpu spread', np.abs(sol[:nf]-ref).max(), 'gpu-oracle', np.abs(dx1[b,free]-sol[:nf]).max(), 'cond', np.linalg.cond(K))
    e = np.abs(dx1[b, free] - ref)
    print(b, 'err', e.max(), 'scale', np.abs(ref).max(), 'argmax var', free[e.argmax()], 'resid', np.abs(K @ np.concatenate([dx1[b,free], np.zeros(nE)])[:nf+nE] )[:0].sum())
