import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
import os
capi.LIB_PATH = capi.LIB_PATH.replace("libqtos_planner.so", os.environ.get("QTOS_LIB", "libqtos_planner_stamps.so"))
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots100(max_iter=80)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = capi.Planner(cfg, max_batch=NB)
start, goal = workloads.flat_goals(NB, 0)
P.plan(start, goal)
NS = P.dims.n_stages
acc = np.zeros((16, 12))
for b in (0, 5, NB - 1):
    t = np.zeros((cfg.max_iter + 1, 4))
    P.lib.qtos_debug_trace(P.h, b, t.ctypes.data_as(C.POINTER(C.c_double)))
    acc += (t[16:64].reshape(16, 12).view(np.uint64).astype(np.float64) if os.environ.get("QTOS_KKT", "2") == "2" else t[16:64].reshape(16, 12))   # (k_kkt2: integer counters)
acc /= 3 * NS
print("cycles per stage by wave: 0 AB work | 1 AB wait | 2 C role work (after prefetch issue; update waves: extraction only) | 3 assembly | 4 C wait | 5 update MFMA loop | 6 backward/NS | 7 top | 8 prefetch issue | 9 inequality products (k_kkt3)")
for w in range(16):
    print("wave %2d: " % w + " ".join("%6.0f" % v for v in acc[w][:12]))
print("stage total (wave 0):", acc[0][[0, 1, 2, 3, 4, 7, 8]].sum(), "timing", P.timing())
ks = np.zeros(8)
for b in (0, 5, NB - 1):
    t = np.zeros((cfg.max_iter + 1, 4))
    P.lib.qtos_debug_trace(P.h, b, t.ctypes.data_as(C.POINTER(C.c_double)))
    ks += t[70:72].reshape(8)
print("k_step cycles (iteration 1): ds=Ji dx | ratio tests + th0 | line search evals | updates + infeasibility | eval with Jacobian | barrier terms:", (ks / 3).round(0))
es = np.zeros(16); el = np.zeros(16)
for b in (0, 5, NB - 1):
    t = np.zeros((cfg.max_iter + 1, 4))
    P.lib.qtos_debug_trace(P.h, b, t.ctypes.data_as(C.POINTER(C.c_double)))
    es += t[72:76].reshape(16); el += t[76:80].reshape(16)
print("eval_all<true> stamps:", (es / 3).round(0))
print("eval_all<false> stamps:", (el / 3).round(0))

hd = np.zeros(4); hc = np.zeros(4); hw_ = np.zeros(16)
for b in (0, 5, NB - 1):
    t = np.zeros((cfg.max_iter + 1, 4))
    P.lib.qtos_debug_trace(P.h, b, t.ctypes.data_as(C.POINTER(C.c_double)))
    hd += t[64]; hc += t[65]; hw_ += t[66:70].reshape(16)
print("helper turn of wave 13 (last sweep): sums | loads | barrier wait per step | turns:", (hd / 3).round(0))
print("chain wave per step (last sweep): partial sums -> x | stores + block product | prefetch | barrier:", (hc / 3).round(0))
print("work before the barrier per step, waves 0..15 (chain, row waves):", (hw_ / 3).round(0))
