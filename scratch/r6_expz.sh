#!/bin/bash
# round 6, verdict item 6a: what stance-z elimination would buy, measured on FLAT ground with the experiment library
# (scratch/devbuild.sh expz -DQTOS_EXP_FIX_STANCE_Z: stance z and their terrain rows out of the KKT system; model.hpp) against the
# product library on one box.  Output: gpurun_out/r6_expz.log
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
X="--cpu-sample 0 --no-parity --no-second-gait"
{
for rep in 1 2; do
for g in walk trot; do
  for lib in libqtos_planner.so libqtos_expz.so; do
    QTOS_LIB=$lib python bench.py $X --gait $g 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$g', '$lib', j['value'], 'plans/s', j['ms_per_step'], 'ms/step', j['roofline']['kernel'], j['roofline']['avg_launch_ms'], 'ms/launch', 'unknowns', j['config']['kkt_unknowns'], 'stages', j['config']['kkt_stages'], 'front', j['config']['front'], 'converged', j['config']['converged'])"
  done
done
done
python3 - <<'PY'
import os, sys, subprocess, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
code = '''
import sys, os, numpy as np, hashlib
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
out = {}
for g in ("walk", "trot"):
    P = Planner(PlannerConfig.knots100(gait=g), max_batch=64)
    s, gl = workloads.flat_goals(64, seed=5)
    n, st, it, v = P.plan(s, gl)
    np.save("/tmp/expz_%s_%s.npy" % (g, os.environ.get("QTOS_LIB", "x")), n)
    print(g, os.environ.get("QTOS_LIB"), int((st == 0).sum()), it.max(), it.mean())
    P.close()
'''
for lib in ("libqtos_planner.so", "libqtos_expz.so"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QTOS_LIB=lib))
import numpy as np
for g in ("walk", "trot"):
    a, b = np.load("/tmp/expz_%s_libqtos_planner.so.npy" % g), np.load("/tmp/expz_%s_libqtos_expz.so.npy" % g)
    print(g, "max |plans(product) - plans(experiment)| =", float(np.abs(a - b).max()))
PY
} > $O/r6_expz.log 2>&1
cat $O/r6_expz.log | grep -v amdgpu.ids
