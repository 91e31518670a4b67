import sys, subprocess, os
code = r'''
import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
import os
from qtos_amd.config import PlannerConfig
P = capi.Planner(PlannerConfig.knots100(), max_batch=256)
s, g = workloads.flat_goals(256, 0)
ts = []
for i in range(12):
    P.plan(s, g); t = P.timing(); ts.append(t["kkt_seconds"] / t["kkt_launches"])
print(os.environ["QTOS_LIB"], "kkt ms: median %.4f min %.4f" % (1e3 * np.median(ts[2:]), 1e3 * min(ts[2:])))
'''
for rep in range(3):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QTOS_LIB=lib))
