"""Numerical check of the block elimination with an explicit L^-1 (numpy emulation of k_kkt's algebra)
against LAPACK, on the same KKT matrix the parity test uses; and the GPU result next to it."""
import sys, json; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi
from qtos_amd.config import PlannerConfig
from oracle.oracle import Oracle
sys.path.insert(0, 'tests')
from test_gpu_parity import start_vector, oracle_problem
cfg = PlannerConfig.reference_compat()
P = capi.Planner(cfg, max_batch=4)
O = Oracle(cfg.oracle_dict())
gv1 = np.load('tests/golden/gv1.npz', allow_pickle=True)
inp = json.loads(str(gv1['inputs'])) if gv1['inputs'].dtype.kind in 'US' else gv1['inputs'].item()
rng = np.random.default_rng(1)
B = 4
x = gv1['x'][None] + 0.01 * rng.standard_normal((B, P.n))
lo, hi = O.var_bounds(oracle_problem(O, inp)); fx = lo == hi
x[:, fx] = lo[fx]
start = np.repeat(start_vector(inp)[None], B, 0); goal = np.repeat(np.array(inp['g'])[None], B, 0)
rk, vf, order = P.structure()
I = rk == 2
sig = np.zeros((B, P.m)); w = np.zeros((B, P.m))
sig[:, I] = 10.0 ** rng.uniform(-3, 3, (B, I.sum())); w[:, I] = rng.standard_normal((B, I.sum()))
dx = P.debug_newton(start, goal, x, sig, w)
b = 0
free, E, Ii = np.nonzero(~fx)[0], np.nonzero(rk == 1)[0], np.nonzero(I)[0]
nf, nE = len(free), len(E)
Jo, go = O.jacobian(x[b]), O.constraints(x[b])
JE, JI = Jo[np.ix_(E, free)], Jo[np.ix_(Ii, free)]
K = np.zeros((nf + nE, nf + nE))
K[:nf, :nf] = cfg.delta_x * np.eye(nf) + JI.T @ (sig[b, Ii][:, None] * JI)
K[nf:, :nf] = JE; K[:nf, nf:] = JE.T; K[nf:, nf:] = -cfg.eps_dual * np.eye(nE)
rhs = np.concatenate([-JI.T @ w[b, Ii], -go[E]])
ref = np.linalg.solve(K, rhs)
# permutation into elimination order
pos_of_var = {v: i for i, v in enumerate(free)}
pos_of_row = {r: nf + i for i, r in enumerate(E)}
perm = np.array([pos_of_var[u] if u < P.n else pos_of_row[u - P.n] for u in order])
N = len(perm); NS = (N + 15) // 16; Np = NS * 16
Kp = np.eye(Np); Kp[:N, :N] = K[np.ix_(perm, perm)]
bp = np.zeros(Np); bp[:N] = rhs[perm]
S = Kp.copy(); y = bp.copy()
Vs, ws = [], []
maxL = 0; maxLi = 0
for k in range(NS):
    p = slice(16 * k, 16 * k + 16); r = slice(16 * k + 16, Np)
    D = S[p, p]
    # LDL^T without pivoting
    L = np.eye(16); d = np.zeros(16); Aw = D.copy()
    for i in range(16):
        d[i] = Aw[i, i]
        L[i + 1:, i] = Aw[i + 1:, i] / d[i]
        Aw[i + 1:, i + 1:] -= np.outer(L[i + 1:, i], Aw[i, i + 1:])
    Li = np.eye(16)
    for i in range(16):
        for kk in range(i):
            Li[i, :kk + 1] -= L[i, kk] * Li[kk, :kk + 1] if False else 0
    Li = np.linalg.inv(L)   # explicit inverse (what the kernel forms by substitution)
    maxL = max(maxL, np.abs(L).max()); maxLi = max(maxLi, np.abs(Li).max())
    Pn = S[r, p]
    Y = Pn @ Li.T
    yF = Li @ y[p]
    S[r, r] -= (Y / d) @ Y.T
    y[r] -= (Y / d) @ yF
    Vs.append((Y / d) @ Li); ws.append(Li.T @ (yF / d))
xsol = np.zeros(Np)
for k in reversed(range(NS)):
    p = slice(16 * k, 16 * k + 16); r = slice(16 * k + 16, Np)
    xsol[p] = ws[k] - Vs[k].T @ xsol[r]
xe = np.zeros(nf + nE); xe[perm] = xsol[:N]
pan, ps = P.factor(b)
# slot of every position: position 16k+j sits in slot ps[k][j]
slot_of_pos = ps.ravel()
for k in range(min(NS, 12)):
    # next occupant of each slot after stage k
    nxt = {}
    for ppos in range(16 * (k + 1), N):
        nxt.setdefault(int(slot_of_pos[ppos]), ppos)
    rows = np.array(sorted(nxt.values()))
    Vg = np.array([pan[k, 1 + slot_of_pos[pp]] for pp in rows])
    Ve = Vs[k][rows - 16 * (k + 1)]
    # occupants that have not entered yet have zero rows in both
    eV = np.abs(Vg - Ve).max(); ew = np.abs(pan[k, 0] - ws[k]).max()
    bad = np.argwhere(np.abs(Vg - Ve) > 1e-6 * max(1.0, np.abs(Ve).max()))
    print("stage %d: |V| %.2e err V %.2e  err w %.2e (|w| %.2e)  bad rows(pos-rel,col) %s" % (k, np.abs(Ve).max(), eV, ew, np.abs(ws[k]).max(), bad[:6].tolist()))
np.set_printoptions(linewidth=200, precision=4, suppress=False)
k = 0
nxt = {}
for ppos in range(16, N):
    nxt.setdefault(int(slot_of_pos[ppos]), ppos)
rows = np.array(sorted(nxt.values()))
print("stage0 pivot slots", ps[0], "diag K", np.diag(Kp)[:16])
print("w gpu", pan[0, 0]); print("w emu", ws[0])
for rr in (8, 9, 0):
    print("row pos-rel", rr, "slot", slot_of_pos[rows[rr]]); print("  gpu", pan[0, 1 + slot_of_pos[rows[rr]]]); print("  emu", Vs[0][rows[rr] - 16])
print("max |L| %.3e  max |L^-1| %.3e" % (maxL, maxLi))
print("emulated explicit-inverse elimination vs LAPACK: %.3e (scale %.3e)" % (np.abs(xe - ref).max(), np.abs(ref).max()))
print("GPU vs LAPACK: %.3e" % np.abs(dx[b, free] - ref[:nf]).max())
err = np.abs(dx[b, free] - ref[:nf]); pos = np.array([list(order).index(v) for v in free])
srt = np.argsort(pos)
print("GPU error by elimination position (every 64th):", np.round(err[srt][::64], 5))
