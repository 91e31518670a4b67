import sys, json, time
sys.path.insert(0, '.')
import numpy as np
from qtos_amd.config import PlannerConfig
from qtos_amd.capi import Planner
from oracle.oracle import Oracle
cfg = PlannerConfig.reference_compat()
P = Planner(cfg, max_batch=8)
d = P.dims
print('dims', {k: getattr(d, k) for k, _ in d._fields_})
O = Oracle(cfg.oracle_dict())
gv = np.load('tests/golden/gv1.npz'); inp = json.loads(str(gv['inputs']))
def startvec(inp):
    return np.concatenate([inp['s'], inp['s_ang'], np.ravel(inp['ee']), inp['s_vel'], inp['s_ang_vel']])
start = startvec(inp)[None]; goal = np.array(inp['g'])[None]
rng = np.random.default_rng(0)
x = gv['x'] + 0.01 * rng.standard_normal(P.n)
q = O.problem(inp['s'], inp['s_ang'], inp['ee'], inp['g'], inp['s_vel'], inp['s_ang_vel'], inp['t0'])
xl, xh = O.var_bounds(q); fx = xl == xh; x[fx] = xl[fx]
g, J = P.debug_eval(start, goal, x[None])
go = O.constraints(x); Jo = O.jacobian(x)
print('g err', np.abs(g[0] - go).max())
rk, vf, order = P.structure()
Jo2 = Jo.copy(); Jo2[:, fx] = 0; Jo2[rk == 0] = 0
print('J err', np.abs(J[0] - Jo2).max(), 'nnz', (J[0] != 0).sum(), (Jo2 != 0).sum())
# newton step parity against a dense solve
lo, hi = O.con_bounds()
sig = np.zeros(P.m); w = np.zeros(P.m)
I = rk == 2
sig[I] = rng.uniform(0.1, 10, I.sum()); w[I] = rng.standard_normal(I.sum())
dx = P.debug_newton(start, goal, x[None], sig[None], w[None])[0]
free = np.nonzero(~fx)[0]; E = np.nonzero(rk == 1)[0]; Ii = np.nonzero(I)[0]
JE = Jo[np.ix_(E, free)]; JI = Jo[np.ix_(Ii, free)]
nf, nE = len(free), len(E)
K = np.zeros((nf + nE, nf + nE))
K[:nf, :nf] = cfg.delta_x * np.eye(nf) + JI.T @ (sig[Ii][:, None] * JI)
K[nf:, :nf] = JE; K[:nf, nf:] = JE.T; K[nf:, nf:] = -cfg.eps_dual * np.eye(nE)
rhs = np.concatenate([-JI.T @ w[Ii], -go[E]])
sol = np.linalg.solve(K, rhs)
print('newton dx err', np.abs(dx[free] - sol[:nf]).max(), 'scale', np.abs(sol[:nf]).max())
# full solve
t = time.time(); nodes, status, iters, viol = P.plan(start, goal); print('plan secs', time.time() - t)
xo, info = O.solve(q)
print('gpu status', status, iters, viol, 'oracle', info.status, info.iters, info.inf_pr)
print('solution diff', np.abs(nodes[0] - xo).max())
print('trace\n', P.trace(0))
print(P.timing())
