import sys; sys.path.insert(0, '.')
import numpy as np, os
from qtos_amd import capi, workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from oracle.oracle import Oracle, oracle_dict, oracle_options
for name, mk in [('refc', lambda sw: PlannerConfig.reference_compat(reduce_swing=sw)), ('walk', lambda sw: PlannerConfig.knots100(reduce_swing=sw))]:
    for sw in (False, True):
        cfg = mk(sw)
        B = 8
        P = Planner(cfg, max_batch=B)
        start, goal = workloads.flat_goals(B, seed=0)
        nodes, status, iters, viol = P.plan(start, goal)
        O = Oracle(oracle_dict(cfg))
        qs = [O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g) for s, g in zip(start, goal)]
        xo, infos = O.solve_batch(qs, n_threads=8, opts=oracle_options(cfg, O))
        print(name, 'sw', sw, 'kernel', P.kkt_kernel(), 'stages', P.dims.n_stages, 'status', status.tolist(), 'iters', iters.tolist(), 'oracle iters', [i.iters for i in infos], 'status', [i.status for i in infos],
              'max diff %.2e' % np.abs(nodes - xo).max(), 'viol %.1e' % viol.max())
        g0 = P.initial_guess(start[:1], goal[:1])
        print('   initial guess diff vs oracle raw guess: %.2e' % np.abs(g0[0] - O.initial_guess(qs[0])).max())
        P.close()
