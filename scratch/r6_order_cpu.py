# round 6: unpivoted numpy LDL^T of the full system's KKT matrix in the planner's order, over short horizons (CPU only) --
# where does the growth of rule 0 on the short trots (profiles/r06_experiments/order_fuzz.log) come from?
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, oracle_dict
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig


def run(cfg, rule, verbose=False):
    old = os.environ.get("QTOS_ORDER")
    if rule is not None:
        os.environ["QTOS_ORDER"] = str(rule)
    try:
        d, _ = capi.analyze(cfg)
        order = capi.analyze_order(cfg)
    finally:
        if old is None:
            os.environ.pop("QTOS_ORDER", None)
        else:
            os.environ["QTOS_ORDER"] = old
    O = Oracle(oracle_dict(cfg))
    n = O.n
    rng = np.random.default_rng(5)
    s, gl = workloads.flat_goals(1, seed=3)
    gl[:, 0] = s[:, 0] + (gl[:, 0] - s[:, 0]) * (cfg.duration / 5.0)
    q = O.problem(s[0, 0:3], s[0, 3:6], s[0, 6:18].reshape(4, 3), gl[0])
    x = O.initial_guess(q) + 0.01 * rng.standard_normal(n)
    lo, hi = O.var_bounds(q)
    fx = lo == hi
    x[fx] = lo[fx]
    clo, chi = O.con_bounds()
    Jo, go = O.jacobian(x), O.constraints(x)
    E = np.array(sorted(int(u - n) for u in order if u >= n))
    free = np.array(sorted(int(u) for u in order if 0 <= u < n))
    Ii = np.nonzero(clo != chi)[0]
    nf, nE = len(free), len(E)
    sig, w = 10.0 ** rng.uniform(-3, 3, len(Ii)), rng.standard_normal(len(Ii))
    JE, JI = Jo[np.ix_(E, free)], Jo[np.ix_(Ii, free)]
    K = np.zeros((nf + nE, nf + nE))
    K[:nf, :nf] = cfg.delta_x * np.eye(nf) + JI.T @ (sig[:, None] * JI)
    K[nf:, :nf] = JE
    K[:nf, nf:] = JE.T
    K[nf:, nf:] = -cfg.eps_dual * np.eye(nE)
    rhs = np.concatenate([-JI.T @ w, -go[E]])
    pos_of = {int(v): i for i, v in enumerate(free)}
    pos_of.update({n + int(r): nf + i for i, r in enumerate(E)})
    real = np.nonzero(order >= 0)[0]
    perm = np.array([pos_of[int(u)] for u in order[real]])
    A, y = K[np.ix_(perm, perm)], rhs[perm]
    N = len(perm)
    L, dd, W = np.eye(N), np.zeros(N), A.copy()
    for i in range(N):
        dd[i] = W[i, i]
        c = W[i + 1:, i] / dd[i]
        L[i + 1:, i] = c
        W[i + 1:, i + 1:] -= np.outer(c, W[i, i + 1:])
    xs = np.linalg.solve(L.T, np.linalg.solve(L, y) / dd)
    ref = np.linalg.solve(A, y)
    err = np.abs(xs - ref).max() / np.abs(ref).max()
    res = np.abs(A @ xs - y).max()
    Lmax = np.abs(L).max()
    if verbose:
        # where are the small pivots / the large multipliers?
        col = np.abs(L - np.eye(N)).max(axis=0)
        worst = np.argsort(-col)[:12]
        for i in sorted(worst):
            u = int(order[real][i])
            print("   pos %4d unknown %5d (%s) pivot %+.3e max|L col| %.2e" % (i, u, "var" if u < n else "row %d" % (u - n), dd[i], col[i]))
    return d, err, res, np.abs(dd).min(), Lmax


if __name__ == "__main__":
    v = "-v" in sys.argv
    for gait, dur, dt in (("trot", 2.5, 0.05), ("trot", 5.0, 0.1), ("trot", 2.5, 0.1), ("trot", 4.0, 0.1), ("walk", 2.5, 0.05), ("trot", 5.0, 0.05)):
        for rule in (0, 1):
            cfg = PlannerConfig(gait=gait, duration=dur, dt_base=dt, dt_dynamic=dt, reduce_base=False, reduce_swing=False)
            d, err, res, pmin, lmax = run(cfg, rule, v)
            print("%s %.1f s dt %.2f full system rule %d: front %3d stages %3d  rel err %.2e  residual %.2e  min|pivot| %.1e  max|L| %.2e" %
                  (gait, dur, dt, rule, d.front, d.n_stages, err, res, pmin, lmax), flush=True)
