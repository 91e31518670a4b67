import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
capi.LIB_PATH = capi.LIB_PATH.replace("libqtos_planner.so", "libqtos_planner_stamps.so")
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots100(max_iter=40)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = capi.Planner(cfg, max_batch=NB)
start, goal = workloads.flat_goals(NB, 0)
P.plan(start, goal)
names = ["S1 Ysolve+retire", "S2 assemble", "S3 gather", "S4 wave0 ldlt", "S4 rest (to barrier)", "install", "drain", "backward"]
tot = np.zeros(8)
for b in (0, 5, NB - 1):
    t = np.zeros((cfg.max_iter + 1, 4))
    P.lib.qtos_debug_trace(P.h, b, t.ctypes.data_as(C.POINTER(C.c_double)))
    st = t[30:32].ravel()
    print('   wave1: tiles, rhs, panel:', (t[32] / P.dims.n_stages).round(0))
    tot += st
    print(b, (st / P.dims.n_stages).round(0))
print("per stage cycles (100 MHz ticks? s_memtime = shader clock):")
for n, v in zip(names, tot / 3 / P.dims.n_stages):
    print("  %-22s %8.0f" % (n, v))
print("sum per stage", tot.sum() / 3 / P.dims.n_stages, "stages", P.dims.n_stages, P.timing())
