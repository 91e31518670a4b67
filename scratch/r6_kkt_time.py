# round 6: KKT ms per launch of library builds whose results may be wrong (timing-only ablations): fixed iteration count, no convergence needed
# usage: python scratch/r6_kkt_time.py lib1.so lib2.so ...   (three alternating passes, trot and walk)
import os, subprocess, sys
code = r'''
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
out = []
for gait in ("trot", "walk"):
    P = capi.Planner(PlannerConfig.knots100(gait=gait, max_iter=6, chord_tol=0.0), max_batch=256)
    s, g = workloads.flat_goals(256, 0)
    ts = []
    for rep in range(6):
        P.plan(s, g)
        t = P.timing()
        ts.append(1e3 * t["kkt_seconds"] / max(t["kkt_launches"], 1))
    out.append("%s kkt %.4f ms (%d launches)" % (gait, float(np.median(ts[1:])), t["kkt_launches"]))
    P.close()
print("  ".join(out))
'''
for rep in range(3):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QTOS_LIB=lib), capture_output=True, text=True, timeout=600)
        print("%-22s %s" % (lib, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
