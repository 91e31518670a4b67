"""Which parts of the time-shifted previous plan make a good starting point, and with which slack push?"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from qtos_amd.replan import ShiftedWindows
from oracle.oracle import Oracle, oracle_dict
B = 64
maps, cell = workloads.random_terrains()
start, goal, map_id = workloads.mpc_goals(B, seed=5, terrains=(maps, cell))
for tname, mk in (("knots200", PlannerConfig.knots200), ("knots100", PlannerConfig.knots100)):
  foff = Oracle(oracle_dict(mk())).L.off_eef[0]
  for push in (0.01, 0.05, 0.2):
    cfg = mk(warm_slack_push=push)
    P = Planner(cfg, max_batch=B); P.set_heightfields(maps, cell)
    for sets in ("all", "base", "base+feet", "cold"):
        W = ShiftedWindows(P, start, goal - start[:, 0:3], map_id, advance=2.5, warm='shifted')
        W._mix, W._force_off = sets, foff
        its, conv = [], []
        for k in range(7):
            W.replan(); torch.cuda.synchronize()
            if k > 0:
                its.append(W.iters.float().mean().item()); conv.append((W.status == 0).float().mean().item())
        print(tname, "push", push, "%-10s" % sets, "mean iters %.2f max %d" % (np.mean(its), int(W.iters.max())), "conv %.3f" % np.mean(conv), flush=True)
    P.close()
