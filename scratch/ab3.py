import sys, subprocess, os
code = r'''
import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
import os
from qtos_amd.config import PlannerConfig
P = capi.Planner(PlannerConfig.knots100(), max_batch=256)
s, g = workloads.flat_goals(256, 0)
ts, tt = [], []
for i in range(12):
    P.plan(s, g); t = P.timing(); ts.append(t["kkt_seconds"] / t["kkt_launches"]); tt.append(t["total_seconds"])
print(os.environ["QTOS_LIB"], "QTOS_TSORT", os.environ.get("QTOS_TSORT"), "kkt ms/launch: median %.4f; whole solve ms: median %.4f" % (1e3 * np.median(ts[2:]), 1e3 * np.median(tt[2:])))
'''
libs = sys.argv[1].split(",")
for rep in range(2):
    for lib in libs:
        for v in sys.argv[2:]:
            subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QTOS_LIB=lib, QTOS_TSORT=v))
