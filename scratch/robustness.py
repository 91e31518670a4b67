"""Convergence statistics over many seeded batches (evidence for DESIGN.md section 6)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
B = 256
cfg = PlannerConfig.knots100()
P = capi.Planner(cfg, max_batch=B)
def stats(tag, gen, seeds, setup=None):
    tot = conv = 0; hist = np.zeros(30, int); worst = 0.0
    for sd in seeds:
        args = gen(sd)
        nodes, status, iters, viol = P.plan(args[0], args[1], map_id=args[2] if len(args) > 2 else None)
        tot += len(status); conv += int((status == 0).sum()); hist += np.bincount(iters, minlength=30)[:30]
        worst = max(worst, float(viol[status == 0].max()))
    nz = np.nonzero(hist)[0]
    print("%-22s %5d problems, converged %5d (%.2f %%), iterations %s, worst converged violation %.1e" % (tag, tot, conv, 100.0 * conv / tot, {int(i): int(hist[i]) for i in nz}, worst))
h1, c1 = workloads.exp1_terrain(); P.set_heightfields(h1, c1)
stats("flat", lambda sd: workloads.flat_goals(B, sd), range(10))
P.build_init_table()
stats("flat, table start", lambda sd: workloads.flat_goals(B, sd), range(10))
P.set_init_table()
t5 = workloads.exp5_terrain(); P.set_heightfields(t5[0], t5[1])
stats("exp5", lambda sd: workloads.step_goals(B, seed=sd, terrain=t5), range(1, 11))
maps, cell = workloads.mixed_terrains(); P.set_heightfields(maps, cell)
stats("mixed", lambda sd: workloads.mixed_goals(B, seed=sd, terrains=(maps, cell)), range(2, 12))
P.close()
P = capi.Planner(PlannerConfig.knots200(), max_batch=B)
maps, cell = workloads.random_terrains(); P.set_heightfields(maps, cell)
stats("random, 200 knots", lambda sd: workloads.mpc_goals(B, seed=sd, terrains=(maps, cell)), range(5, 10))
P.close()
P = capi.Planner(PlannerConfig.knots100(gait="trot"), max_batch=B)
P.set_heightfields(h1, c1)
stats("flat, trot", lambda sd: workloads.flat_goals(B, sd), range(10))
