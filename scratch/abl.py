"""diagnostic: launch time of k_kkt2 with parts of a stage removed (timing only: the results of these builds are wrong)"""
import sys, subprocess, os
code = r'''
import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
import os
from qtos_amd.config import PlannerConfig
P = capi.Planner(PlannerConfig.knots100(max_iter=3, chord_tol=0.0), max_batch=256)
s, g = workloads.flat_goals(256, 0)
ts = []
for i in range(8):
    P.plan(s, g); t = P.timing(); ts.append(t["kkt_seconds"] / max(t["kkt_launches"], 1))
print("%-34s kkt ms/launch: median %.4f" % (os.environ["QTOS_LIB"], 1e3 * np.median(ts[2:])))
'''
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QTOS_LIB=lib))
