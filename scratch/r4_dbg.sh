#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
QTOS_KKT=${1:-5} timeout 600 python - > $O/r4_dbg.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, os
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.reference_compat()
B = 2
s, g = workloads.flat_goals(B, 0)
rng = np.random.default_rng(3)
res = {}
for kkt in ("2", os.environ["QTOS_KKT"]):
    os.environ["QTOS_KKT"] = kkt
    P = capi.Planner(cfg, max_batch=B)
    x0 = P.initial_guess(s, g)
    rk, _, order = P.structure()
    I = rk == 2
    rng = np.random.default_rng(3)
    sig = np.zeros((B, P.m)); w = np.zeros((B, P.m))
    sig[:, I] = 10.0 ** rng.uniform(-2, 2, (B, I.sum())); w[:, I] = rng.standard_normal((B, I.sum()))
    dx = P.debug_newton(s, g, x0, sig, w)
    pan, ps = P.factor(0)
    res[kkt] = (dx.copy(), pan.copy())
    print("kkt", kkt, "front", P.dims.front, "stages", P.dims.n_stages, "dx max", np.abs(dx).max())
    P.close()
(d2, p2), (d5, p5) = res["2"], res[os.environ["QTOS_KKT"]]
print("dx diff", np.abs(d2 - d5).max())
for k in range(p2.shape[0]):
    e = np.abs(p2[k] - p5[k]).max(); sc = np.abs(p2[k]).max()
    if e > 1e-9 * max(sc, 1): print("stage", k, "panel diff", e, "scale", sc, "w diff", np.abs(p2[k,0]-p5[k,0]).max(), "rows differing", np.nonzero(np.abs(p2[k,1:]-p5[k,1:]).max(axis=1) > 1e-9*max(sc,1))[0][:12]); 
    if k > 6 and e > 1e-9: break
PY
head -40 $O/r4_dbg.log
