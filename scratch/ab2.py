import sys, subprocess, os
code = r'''
import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
import os
from qtos_amd.config import PlannerConfig
P = capi.Planner(PlannerConfig.knots100(), max_batch=256)
s, g = workloads.flat_goals(256, 0)
ts, tt, tc = [], [], []
for i in range(12):
    P.plan(s, g); t = P.timing(); ts.append(t["kkt_seconds"] / t["kkt_launches"]); tt.append(t["total_seconds"]); tc.append(t["chord_seconds"])
print(os.environ["QTOS_LIB"], "kkt ms/launch: median %.4f; whole solve ms: median %.4f; chord %.4f; other (start, steps, gaps) %.4f" % (1e3 * np.median(ts[2:]), 1e3 * np.median(tt[2:]), 1e3 * np.median(tc[2:]), 1e3 * np.median(np.array(tt[2:]) - np.array(tc[2:]) - np.array(ts[2:]) * t["kkt_launches"])))
'''
for rep in range(3):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QTOS_LIB=lib))
