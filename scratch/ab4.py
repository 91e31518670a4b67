# A/B of the KKT kernels inside ONE library: the default (k_kkt2) against QTOS_KKT=3 (k_kkt3 where it applies).
# usage: python scratch/ab4.py [lib ...]   (each lib is run with QTOS_KKT=2 and without)
import sys, subprocess, os
code = r'''
import sys; sys.path.insert(0, '.')
import numpy as np, os, hashlib
from qtos_amd import capi, workloads
if os.environ.get("QTOS_LIB"): capi.LIB_PATH = capi.LIB_PATH.replace("libqtos_planner.so", os.environ["QTOS_LIB"])
from qtos_amd.config import PlannerConfig
gait = os.environ.get("AB_GAIT", "walk")
tr = os.environ.get("AB_CFG", "knots100")
cfg = getattr(PlannerConfig, tr)(gait=gait) if gait != "walk" else getattr(PlannerConfig, tr)()
P = capi.Planner(cfg, max_batch=256)
s, g = workloads.flat_goals(256, 0)
ts, tt, tc = [], [], []
for i in range(12):
    nodes, status, iters, viol = P.plan(s, g); t = P.timing(); ts.append(t["kkt_seconds"] / max(t["kkt_launches"], 1)); tt.append(t["total_seconds"]); tc.append(t["chord_seconds"])
print("%-28s KKT=%s %s kkt ms/launch %.4f (x%d); whole solve ms %.4f; chord %.4f; conv %d/256 iters max %d; sha %s" % (os.environ.get("QTOS_LIB", "default") + " " + tr + " F=%d" % P.dims.front, os.environ.get("QTOS_KKT", "2"), gait,
      1e3 * np.median(ts[2:]), t["kkt_launches"], 1e3 * np.median(tt[2:]), 1e3 * np.median(tc[2:]), int((status == 0).sum()), int(iters.max()), hashlib.sha1(nodes.tobytes()).hexdigest()[:10]))
np.save("/tmp/ab4_nodes_%s_%s.npy" % (os.environ.get("QTOS_KKT", "2"), gait), nodes)
'''
libs = sys.argv[1:] or [""]
for gait in os.environ.get("AB_GAITS", "walk").split(","):
    for rep in range(2):
        for lib in libs:
            for kkt in ("2", os.environ.get("AB_KKT", "3")):
                env = dict(os.environ, AB_GAIT=gait)
                if lib: env["QTOS_LIB"] = lib
                env["QTOS_KKT"] = kkt
                subprocess.run([sys.executable, "-c", code], env=env)
    import numpy as np
    try:
        a, b = np.load("/tmp/ab4_nodes_2_%s.npy" % gait), np.load("/tmp/ab4_nodes_%s_%s.npy" % (os.environ.get("AB_KKT", "3"), gait))
        print(gait, "max |kkt3 - kkt2| over the nodes of the batch: %.3e" % np.abs(a - b).max())
    except Exception as e:
        print("no comparison:", e)
