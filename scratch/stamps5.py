"""Per-wave cycle stamps of k_kkt5 (diagnostic build: scratch/build.sh -> libqtos_planner_stamps.so; QTOS_KKT=6)."""
import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np, os
from qtos_amd import capi, workloads
capi.LIB_PATH = capi.LIB_PATH.replace("libqtos_planner.so", os.environ.get("QTOS_LIB", "libqtos_planner_stamps.so"))
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots100(max_iter=80, gait=os.environ.get("GAIT", "walk"))
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = capi.Planner(cfg, max_batch=NB)
start, goal = workloads.flat_goals(NB, 0)
P.plan(start, goal)
NS = P.dims.n_stages
acc = np.zeros((16, 12))
for b in (0, 5, NB - 1):
    t = np.zeros((cfg.max_iter + 1, 4))
    P.lib.qtos_debug_trace(P.h, b, t.ctypes.data_as(C.POINTER(C.c_double)))
    acc += t[16:64].reshape(16, 12).view(np.uint64).astype(np.float64)
acc /= 3 * (NS / 2 + 2)
print("cycles per STEP (pair of stages) by wave: 0 phase-1 work | 1 barrier 1 | 2 phase-2 work | 3 barrier 2 | 4 phase-3 role work | 5 phase-3 assembly | 6 barrier 3 | 7 sweep (per step) | 8 phase 1: V, W chain | 9, 10 next columns of the two stages")
for w in range(12):
    print("wave %2d: " % w + " ".join("%6.0f" % v for v in acc[w][:11]), " sum(0..6) %6.0f" % acc[w][:7].sum())
print("timing", P.timing(), "front", P.dims.front, "stages", NS)
