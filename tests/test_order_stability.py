"""The chain of fronts eliminates the KKT matrix WITHOUT pivoting, 16 unknowns at a time, in the order the symbolic analysis picks
(time stamps; long-lived unknowns at the end of their life; a multiplier behind variables of its row).  That order is what keeps
the factorisation accurate: round 6 searched other time keys for smaller fronts (force nodes later: the 100-knot walk fits 96
slots instead of 112, -15 % per KKT launch) and found every one of them numerically broken on the GPU -- growth 1e11 instead of
1e8 (profiles/r06_experiments/order_keys.log).  This test pins the property on the CPU: the KKT matrix of the reference's NLP
(every row with its multiplier) at a perturbed golden plan, eliminated in the planner's order by an unpivoted numpy LDL^T,
solves to 1e-7 of a pivoted dense solve, with no factor entry beyond 1 / eps_dual."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("transcription,order", [("reference_compat", "auto"), ("knots100_trot", "auto"), ("reference_compat", "1"), ("knots100_trot", "2")])
def test_unpivoted_elimination_in_the_planners_order_is_accurate(transcription, order):
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd import capi
    from qtos_amd.config import PlannerConfig
    cfg = (PlannerConfig.reference_compat(reduce_base=False, reduce_swing=False) if transcription == "reference_compat"
           else PlannerConfig.knots100(gait="trot", reduce_base=False, reduce_swing=False))
    # order "1": round 6's order with the late force nodes forced (QTOS_ORDER=1; on these full systems the planner itself keeps the
    # order of rounds 1 - 5) -- what its guard at the first dynamics knot is for: without it the growth is 1.8e11
    old_env = os.environ.get("QTOS_ORDER")
    # order "2": the guard without the late force nodes (what a reduced base runs since the second half of round 6; here, with
    # every base row in the system, it differs from rule 0 by the guard alone)
    if order in ("1", "2"):
        os.environ["QTOS_ORDER"] = order
    try:
        d, _ = capi.analyze(cfg)
        order = capi.analyze_order(cfg)
    finally:
        if old_env is None:
            os.environ.pop("QTOS_ORDER", None)
        else:
            os.environ["QTOS_ORDER"] = old_env
    assert len(order) == d.n_stages * d.pivots and (order >= 0).sum() == d.n_unknowns
    O = Oracle(oracle_dict(cfg))
    n = O.n
    rng = np.random.default_rng(5)
    if transcription == "reference_compat":
        g = np.load(os.path.join(ROOT, "tests", "golden", "gv1.npz"))
        inp = json.loads(str(g["inputs"]))
        q = O.problem(inp["s"], inp["s_ang"], np.array(inp["ee"]), inp["g"])
        x = g["x"] + 0.01 * rng.standard_normal(n)
    else:
        from qtos_amd import workloads
        s, gl = workloads.flat_goals(1, seed=3)
        q = O.problem(s[0, 0:3], s[0, 3:6], s[0, 6:18].reshape(4, 3), gl[0])
        x = O.initial_guess(q) + 0.01 * rng.standard_normal(n)
    lo, hi = O.var_bounds(q)
    fx = lo == hi
    x[fx] = lo[fx]
    clo, chi = O.con_bounds()
    Jo, go = O.jacobian(x), O.constraints(x)
    E = np.array(sorted(int(u - n) for u in order if u >= n))
    free = np.array(sorted(int(u) for u in order if 0 <= u < n))
    assert not fx[free].any() and (clo[E] == chi[E]).all()          # the order lists free variables and equality rows
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
    for i in range(N):                       # unpivoted LDL^T, right-looking
        dd[i] = W[i, i]
        c = W[i + 1:, i] / dd[i]
        L[i + 1:, i] = c
        W[i + 1:, i + 1:] -= np.outer(c, W[i, i + 1:])
    xs = np.linalg.solve(L.T, np.linalg.solve(L, y) / dd)
    ref = np.linalg.solve(A, y)
    err = np.abs(xs - ref).max() / np.abs(ref).max()
    print(transcription, "unknowns", N, "rel error %.2e" % err, "min |pivot| %.1e" % np.abs(dd).min(), "max |L| %.2e" % np.abs(L).max())
    assert err < 1e-7 and np.abs(L).max() <= 1.05 / cfg.eps_dual
