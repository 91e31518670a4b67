"""sha1 of the plans of the flat and the mixed batch per development library (QTOS_KKT=2: the fronts a development build holds):
python scratch/ab_hash.py lib1.so lib2.so ..."""
import sys, os, subprocess
code = r'''
import sys, hashlib; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
out = []
P = capi.Planner(PlannerConfig.knots100(), max_batch=256)
t = workloads.mixed_terrains(); P.set_heightfields(t[0], t[1])
for name, (s, g, m) in (("flat", workloads.flat_goals(256, 0) + (None,)), ("mixed", workloads.mixed_goals(256, seed=2, terrains=t))):
    r = P.plan(s, g, map_id=m)
    out.append("%s %s it=%.3f ok=%d" % (name, hashlib.sha1(np.ascontiguousarray(r[0]).tobytes()).hexdigest()[:10], np.mean(r[2]), int((r[1] == 0).sum())))
print("  ".join(out))
'''
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QTOS_LIB=lib, QTOS_KKT="2"), capture_output=True, text=True, timeout=600)
    print("%-22s %s" % (lib, (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1]))
