# KKT time per launch of the long-horizon fronts (`-duration 12 / 20`), with and without short stages (QTOS_NO_SHORT_STAGES=1)
import sys, subprocess, os
code = r'''
import sys; sys.path.insert(0, '.')
import numpy as np, os
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
dur, rb = float(os.environ["DUR"]), os.environ["RB"] == "1"
cfg = PlannerConfig.reference_compat(duration=dur, reduce_base=rb)
B = 128
P = capi.Planner(cfg, max_batch=B)
s, g = workloads.flat_goals(B, 0)
g = s[:, 0:3] + (g - s[:, 0:3]) * dur / 5.0
ts, tt = [], []
for i in range(6):
    nodes, status, iters, viol = P.plan(s, g); t = P.timing(); ts.append(t["kkt_seconds"] / max(t["kkt_launches"], 1)); tt.append(t["total_seconds"])
print("duration %4.1f reduce_base %d short_stages %s: front %3d, %3d stages, %4d unknowns; kkt ms/launch %.4f (x%d), whole solve %.3f ms; converged %d/%d, iters max %d" %
      (dur, rb, "off" if os.environ.get("QTOS_NO_SHORT_STAGES") else "on ", P.dims.front, P.dims.n_stages, P.dims.n_unknowns, 1e3 * np.median(ts[1:]), t["kkt_launches"], 1e3 * np.median(tt[1:]), int((status == 0).sum()), B, int(iters.max())))
'''
for dur in ("12.0", "20.0", "8.0"):
    for rb in ("1", "0"):
        for ns in ("", "1"):
            env = dict(os.environ, DUR=dur, RB=rb)
            if ns: env["QTOS_NO_SHORT_STAGES"] = ns
            else: env.pop("QTOS_NO_SHORT_STAGES", None)
            subprocess.run([sys.executable, "-c", code], env=env)
