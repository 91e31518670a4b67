# A/B of a k_step switch inside ONE library (QTOS_SPEC_JAC=0/1): whole solve time, iterations and the plans' hash per workload
# usage: python scratch/ab5.py     (env AB_VAR names the switch, AB_VALS its two values)
import sys, subprocess, os
code = r'''
import sys; sys.path.insert(0, '.')
import numpy as np, os, hashlib
from qtos_amd import capi, workloads
if os.environ.get("QTOS_LIB"): capi.LIB_PATH = capi.LIB_PATH.replace("libqtos_planner.so", os.environ["QTOS_LIB"])
from qtos_amd.config import PlannerConfig
wl = os.environ["AB_WL"]
gait = "trot" if wl == "trot" else "walk"
mk = getattr(PlannerConfig, os.environ.get("AB_CFG", "knots100"))
cfg = mk(gait=gait) if gait != "walk" else mk()
mid = None
if wl == "exp5": ter = workloads.exp5_terrain(); s, g = workloads.step_goals(256, 1, ter)
elif wl == "mixed": ter = workloads.mixed_terrains(); s, g, mid = workloads.mixed_goals(256, 2, ter)
else: ter = workloads.exp1_terrain(); s, g = workloads.flat_goals(256, 0)
P = capi.Planner(cfg, max_batch=256)
P.set_heightfields(ter[0], ter[1])
kw = {} if mid is None else {"map_id": mid}
tt, tk, tc = [], [], []
for i in range(12):
    nodes, status, iters, viol = P.plan(s, g, **kw); t = P.timing(); tt.append(t["total_seconds"]); tk.append(t["kkt_seconds"] / max(t["kkt_launches"], 1)); tc.append(t["chord_seconds"] / max(t["chord_launches"], 1))
print("%-6s %s=%s whole solve ms %.4f (kkt %.4f / launch, chord %.4f); kkt %d chord %d; conv %d/256 iters mean %.3f max %d; sha %s" % (wl, os.environ["AB_VAR"], os.environ.get(os.environ["AB_VAR"], "-"),
      1e3 * np.median(tt[2:]), 1e3 * np.median(tk[2:]), 1e3 * np.median(tc[2:]), t["kkt_launches"], t["chord_launches"], int((status == 0).sum()), iters.mean(), int(iters.max()), hashlib.sha1(nodes.tobytes()).hexdigest()[:10]))
np.save("/tmp/ab5_%s_%s.npy" % (wl, os.environ.get(os.environ["AB_VAR"], "-")), nodes)
'''
var = os.environ.get("AB_VAR", "QTOS_SPEC_JAC")
vals = os.environ.get("AB_VALS", "0,1").split(",")
import numpy as np
for wl in os.environ.get("AB_WLS", "walk,trot,exp5,mixed").split(","):
    for rep in range(2):
        for v in vals:
            env = dict(os.environ, AB_WL=wl, AB_VAR=var)
            env[var] = v
            subprocess.run([sys.executable, "-c", code], env=env)
    try:
        a, b = np.load("/tmp/ab5_%s_%s.npy" % (wl, vals[0])), np.load("/tmp/ab5_%s_%s.npy" % (wl, vals[1]))
        print(wl, "max difference of the plans: %.3e" % np.abs(a - b).max())
    except Exception as e:
        print("no comparison:", e)
