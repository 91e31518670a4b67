# scaling of the CPU oracle over the host's threads (bench.py cpu_baseline): plans/s and the solver's own timers per thread count
import sys, time, os; sys.path.insert(0, '.')
import numpy as np
from oracle.oracle import Oracle, oracle_dict, oracle_options
from qtos_amd.config import PlannerConfig
from qtos_amd import workloads
cfg = PlannerConfig.knots100()
O = Oracle(oracle_dict(cfg))
s, g = workloads.flat_goals(256, 0)
qs = [O.problem(a[0:3], a[3:6], a[6:18].reshape(4, 3), b) for a, b in zip(s, g)]
opts = oracle_options(cfg, O)
cores = len(os.sched_getaffinity(0))
for nt in [1, 8, 16, 32, 64, 128, 256]:
    if nt > cores: break
    reps = [qs[i % 256] for i in range(4 * nt)]
    for _ in range(2): O.solve_batch(reps[:nt], n_threads=nt, opts=opts)
    t = time.perf_counter(); x, inf = O.solve_batch(reps, n_threads=nt, opts=opts); dt = time.perf_counter() - t
    print("threads %3d: %8.1f plans/s  (x%.1f of 1 thread's %s)  mean factor %.1f ms, eval %.1f ms, wall per solve per thread %.1f ms" %
          (nt, len(reps) / dt, 0, "", 1e3 * np.mean([i.factor_secs for i in inf]), 1e3 * np.mean([i.eval_secs for i in inf]), 1e3 * dt * nt / len(reps)))
