"""Launch time of the KKT kernel of a (possibly ablated: wrong results) development library: scratch/abl5.py lib1 lib2 ..."""
import sys, os, subprocess
code = '''
import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots100(max_iter=6, chord_tol=0.0)
P = capi.Planner(cfg, max_batch=256)
start, goal = workloads.flat_goals(256, 0)
ts = []
for rep in range(4):
    P.plan(start, goal)
    t = P.timing()
    ts.append(t['kkt_seconds'] / max(t['kkt_launches'], 1))
print('%.4f ms per launch (%d launches)' % (1e3 * min(ts[1:]), t['kkt_launches']))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, QTOS_LIB='libqtos_%s.so' % lib, QTOS_KKT=os.environ.get('QTOS_KKT', '6'))
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    print('%-10s %s' % (lib, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]))
