import sys; sys.path.insert(0, '.')
import numpy as np
from oracle.oracle import Oracle
from qtos_amd import workloads
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots100()
O = Oracle(cfg.oracle_dict())
L = O.L
nb = L.n_base_nodes
def prob(s, g): return O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g)
sT = workloads.rest_start(0.0); gT = np.array([0.45, 0.0, 0.24])
xT, iT = O.solve(prob(sT, gT)); print("template", iT.status, iT.iters)
def guess(s, g):
    x = xT.copy()
    dx, dy = g[0] - s[0], g[1] - s[1]
    sc = dx / 0.45
    # base lin nodes: [pos3, vel3] per node
    for i in range(nb):
        o = L.off_lin + 6 * i
        prog = xT[o] / 0.45
        x[o] = s[0] + xT[o] * sc
        x[o + 1] = s[1] + xT[o + 1] + prog * dy
        x[o + 3] = xT[o + 3] * sc
        x[o + 4] = xT[o + 4] + xT[o + 3] / 0.45 * dy
    for e in range(4):
        o = L.off_eem[e]; n = L.n_eem[e]
        # layout per foot: 5 stance nodes x 3 + 4 swing nodes x 5 = 35, interleaved: stance s at 8 s (3 vals), swing after it (5 vals)
        ref = xT[o:o + 2].copy()
        foot0 = s[6 + 3 * e: 8 + 3 * e]
        k = 0
        while k < n:
            is_stance = (k % 8) == 0
            px, py = xT[o + k], xT[o + k + 1]
            prog = (px - ref[0]) / 0.45
            x[o + k] = foot0[0] + (px - ref[0]) * sc
            x[o + k + 1] = foot0[1] + (py - ref[1]) + prog * dy
            if is_stance: k += 3
            else:
                x[o + k + 3] = xT[o + k + 3] * sc
                x[o + k + 4] = xT[o + k + 4] + xT[o + k + 3] / 0.45 * dy
                k += 5
    for e in range(4):
        o = L.off_eef[e]; n = L.n_eef[e]
        for k in range(0, n, 6):     # force nodes: f xyz, fdot xyz
            for h in (0, 3):
                x[o + k + h] = xT[o + k + h] * sc
                x[o + k + h + 1] = xT[o + k + h + 1] + xT[o + k + h] / 0.45 * dy
    return x
start, goal = workloads.flat_goals(12, seed=0)
for b in range(12):
    q = prob(start[b], goal[b])
    xc, ic = O.solve(q)
    x0 = guess(start[b], goal[b])
    v0 = O.max_violation(x0)
    xw, iw = O.solve(q, x0=x0)
    print(b, "cold iters", ic.iters, "| template guess viol0 %.2e" % v0, "iters", iw.iters, "status", iw.status, "viol %.1e" % iw.inf_pr)
