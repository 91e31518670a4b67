import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
import os
capi.LIB_PATH = capi.LIB_PATH.replace("libqtos_planner.so", os.environ.get("QTOS_LIB", "libqtos_planner_stamps.so"))
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots100(max_iter=56)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = capi.Planner(cfg, max_batch=NB)
start, goal = workloads.flat_goals(NB, 0)
P.plan(start, goal)
names = ["AB tail (stores, barrier)", "C wave0 factor", "C rest (to barrier)", "drain", "backward (total/NS)", "top: install+prefetch", "AB loads + yt/zt MFMA", "AB acc/vt MFMA"]
tot = np.zeros(8); t1 = np.zeros(4); t7 = np.zeros(4)
for b in (0, 5, NB - 1):
    t = np.zeros((cfg.max_iter + 1, 4))
    P.lib.qtos_debug_trace(P.h, b, t.ctypes.data_as(C.POINTER(C.c_double)))
    tot += t[30:32].ravel(); t1 += t[32]; t7 += t[33]
NS = P.dims.n_stages
print("per stage cycles (s_memtime = shader clock), mean of 3 problems:")
for n, v in zip(names, tot / 3 / NS):
    print("  %-22s %8.0f" % (n, v))
print("  wave1: tile MFMA %.0f, extraction %.0f, assembly %.0f | wave7 rhs %.0f" % (t1[2] / 3 / NS, t1[0] / 3 / NS, t1[1] / 3 / NS, t7[0] / 3 / NS))
print("sum per stage", tot.sum() / 3 / NS, "stages", NS, P.timing())
ks = np.zeros(8)
for b in (0, 5, NB - 1):
    t = np.zeros((cfg.max_iter + 1, 4))
    P.lib.qtos_debug_trace(P.h, b, t.ctypes.data_as(C.POINTER(C.c_double)))
    ks += t[34:36].ravel()
print("k_step (iteration 1) cycles: ds %.0f | dz+reductions %.0f | line search %.0f | update+infeasibility %.0f | linearise %.0f | barrier terms %.0f" % tuple(ks[:6] / 3))
t = np.zeros((cfg.max_iter + 1, 4))
P.lib.qtos_debug_trace(P.h, 0, t.ctypes.data_as(C.POINTER(C.c_double)))
print("linearise phases (cycles): stage x %.0f | dynamics knots %.0f | dynamics columns %.0f | rom instances %.0f | rom columns %.0f | force/terrain/linear %.0f" % tuple(t[36:38].ravel()[:6]))
print("phase C, cycles per stage until each wave reaches the barrier (waves 0..7):", (t[38:40].ravel() / NS).round(0))
print("line-search evaluation phases (cycles): stage x %.0f | dynamics knots %.0f | rom instances %.0f | force/terrain/linear %.0f" % tuple(t[40:42].ravel()[:4]))
print("phase AB, cycles per stage until each wave reaches the barrier (waves 0..7):", (t[42:44].ravel() / NS).round(0))
