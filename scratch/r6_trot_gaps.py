# round 6: the trot's gap to the oracle under the order the planner keeps now (rule 2) and under the order of rounds 1 - 5 (QTOS_ORDER=0),
# per KKT kernel; the gap between two kernels of the product -- what the 5e-6 gates of the trot's parity tests were made of
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import _batch_vs_oracle
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig

def plan(cfg, B, start, goal, kkt=None, order=None):
    for k, v in (("QTOS_KKT", kkt), ("QTOS_ORDER", order)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    try:
        P = Planner(cfg, max_batch=B)
    finally:
        os.environ.pop("QTOS_KKT", None); os.environ.pop("QTOS_ORDER", None)
    r = P.plan(start, goal); name = P.kkt_kernel(); rule = P.dims.order_rule
    P.close()
    return r, name, rule

for gait in ("trot", "walk"):
    for kw in ({}, {"reduce_swing": False}, {"reduce_base": False}):
        cfg = PlannerConfig.knots100(gait=gait, **kw)
        B = 32
        start, goal = workloads.flat_goals(B, seed=0)
        for order in (None, "0"):
            base = None
            for kkt in ("2", "4", "6"):
                (n, s, it, v), name, rule = plan(cfg, B, start, goal, kkt, order)
                same, worst = _batch_vs_oracle(cfg, start, goal, range(16), status=s, iters=it, nodes=n, tol=1e-4)
                if base is None: base = n
                print("%-5s %-24s order %-4s rule %d %-16s converged %2d/%d iters %d..%d  gap to oracle %.2e (same %d/16)  gap to k_kkt2 %.2e" %
                      (gait, kw or "default", order or "auto", rule, name, int((s == 0).sum()), B, it.min(), it.max(), worst, same, float(np.abs(n - base).max())), flush=True)
