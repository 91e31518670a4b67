#!/bin/bash
# round 6: ordering keys of the elimination order (force nodes, base coefficients, dynamics multipliers shifted in time) -- exploration
# library libqtos_exptf.so with QTOS_EXP_TF / TFD / TB / TG against the product library on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
X="--cpu-sample 0 --no-parity --no-second-gait"
run() { # name, env...
  name=$1; shift
  for g in walk trot; do
    env "$@" python bench.py $X --gait $g 2>/dev/null | python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$name', '$g', j['value'], 'plans/s', j['ms_per_step'], 'ms/step', j['roofline']['kernel'], j['roofline']['avg_launch_ms'], 'ms/launch', j['roofline'].get('chord_avg_launch_ms'), 'stages', j['config']['kkt_stages'], 'front', j['config']['front'], 'converged', j['config']['converged'], 'iters', j['config']['iterations_mean'])
except Exception as e: print('$name $g FAILED', e)"
  done
}
{
run product QTOS_LIB=libqtos_planner.so
run order_a QTOS_LIB=libqtos_exptf.so QTOS_EXP_TF=0.25 QTOS_EXP_TFD=0.25 QTOS_EXP_TB=-1.0 QTOS_EXP_TG=-0.5
run order_b QTOS_LIB=libqtos_exptf.so QTOS_EXP_TF=0.5 QTOS_EXP_TFD=0.5 QTOS_EXP_TB=-0.5 QTOS_EXP_TG=-0.5
run product2 QTOS_LIB=libqtos_planner.so
python3 - <<'PY'
import os, sys, subprocess
code = '''
import sys, os, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
tag = os.environ["TAG"]
for g in ("walk", "trot"):
    cfg = PlannerConfig.knots100(gait=g)
    P = Planner(cfg, max_batch=64)
    s, gl = workloads.flat_goals(64, seed=5)
    n, st, it, v = P.plan(s, gl)
    np.save("/tmp/ord_%s_%s.npy" % (g, tag), n)
    # accuracy of one KKT solve: residual of the first Newton system at the straight-line start
    x0 = P.initial_guess(s[:4], gl[:4])
    rng = np.random.default_rng(0)
    sig = rng.uniform(0.1, 10.0, (4, P.m)); w = rng.standard_normal((4, P.m))
    P.debug_newton(s[:4], gl[:4], x0, sig, w)
    dx, res = P.debug_residual(4, refine=False)
    print(tag, g, "converged", int((st == 0).sum()), "iters max", int(it.max()), "kkt residual", float(res.max()), "front", P.dims.front, "stages", P.dims.n_stages)
    P.close()
'''
envs = {"product": dict(QTOS_LIB="libqtos_planner.so"),
        "order_a": dict(QTOS_LIB="libqtos_exptf.so", QTOS_EXP_TF="0.25", QTOS_EXP_TFD="0.25", QTOS_EXP_TB="-1.0", QTOS_EXP_TG="-0.5"),
        "order_b": dict(QTOS_LIB="libqtos_exptf.so", QTOS_EXP_TF="0.5", QTOS_EXP_TFD="0.5", QTOS_EXP_TB="-0.5", QTOS_EXP_TG="-0.5")}
for tag, e in envs.items():
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TAG=tag, **e))
import numpy as np
for g in ("walk", "trot"):
    a = np.load("/tmp/ord_%s_product.npy" % g)
    for tag in ("order_a", "order_b"):
        b = np.load("/tmp/ord_%s_%s.npy" % (g, tag))
        print(g, tag, "max |plans - product's| =", float(np.abs(a - b).max()))
PY
} > $O/r6_order.log 2>&1
grep -v amdgpu.ids $O/r6_order.log
