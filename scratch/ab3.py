"""A/B of library builds (names in csrc/, first = baseline): sha1 of the plans of four workloads (bit-identity) and timing of
the benchmark batch split into factor launches / chord / the rest (k_start, k_step, gaps)"""
import sys, subprocess, os
code = r'''
import sys, hashlib; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
import os
from qtos_amd.config import PlannerConfig
def sha(P, s, g, m=None):
    r = P.plan(s, g, map_id=m)
    nodes = r["nodes"] if isinstance(r, dict) else r[0]
    tag = "gpurun_out/ab3_%d.npy" % len(out)
    d = ""
    if os.environ.get("AB3_BASE") == "1": np.save(tag, nodes)
    elif os.path.exists(tag): d = " d=%.1e" % np.abs(np.load(tag) - nodes).max()
    return hashlib.sha1(np.ascontiguousarray(nodes).tobytes()).hexdigest()[:8] + d + " it=%.2f" % np.mean(r[2])
out = []
P = capi.Planner(PlannerConfig.knots100(), max_batch=256)
t = workloads.exp1_terrain(); P.set_heightfields(t[0], t[1])
s, g = workloads.flat_goals(256, 0)
out.append("flat " + sha(P, s, g))
ts, tt, tc = [], [], []
for i in range(12):
    P.plan(s, g); t = P.timing(); ts.append(t["kkt_seconds"] / t["kkt_launches"]); tt.append(t["total_seconds"]); tc.append(t["chord_seconds"])
other = np.array(tt[2:]) - np.array(tc[2:]) - np.array(ts[2:]) * t["kkt_launches"]
del P
P = capi.Planner(PlannerConfig.knots100(), max_batch=64)
t = workloads.mixed_terrains(); P.set_heightfields(t[0], t[1])
s, g, m = workloads.mixed_goals(64, seed=2, terrains=t)
out.append("mixed " + sha(P, s, g, m))
del P
P = capi.Planner(PlannerConfig.knots100(gait="trot"), max_batch=32)
t = workloads.exp1_terrain(); P.set_heightfields(t[0], t[1])
s, g = workloads.flat_goals(32, 3)
out.append("trot " + sha(P, s, g))
del P
P = capi.Planner(PlannerConfig.reference_compat(), max_batch=32)
t = workloads.exp5_terrain(); P.set_heightfields(t[0], t[1])
s, g = workloads.step_goals(32, seed=1, terrain=t)
out.append("compat-exp5 " + sha(P, s, g))
del P
P = capi.Planner(PlannerConfig.knots100(reduce_base=False), max_batch=32)
t = workloads.exp5_terrain(); P.set_heightfields(t[0], t[1])
s, g = workloads.step_goals(32, seed=1, terrain=t)
out.append("full-hold-exp5 " + sha(P, s, g))
print("%-26s kkt %.4f whole %.4f chord %.4f other %.4f | %s" % (os.environ["QTOS_LIB"], 1e3 * np.median(ts[2:]), 1e3 * np.median(tt[2:]), 1e3 * np.median(tc[2:]), 1e3 * np.median(other), "  ".join(out)))
'''
for rep in range(int(os.environ.get("REPS", "2"))):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QTOS_LIB=lib, AB3_BASE="1" if lib == sys.argv[1] else "0"))
