"""diagnostic: where a replan of a set of receding windows spends its time (host side), sets alone vs side by side"""
import sys, time, threading; sys.path.insert(0, '.')
import numpy as np, torch
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from qtos_amd.replan import ShiftedWindows
dev = torch.device("cuda", 0)
cfg = PlannerConfig.knots200(chord_tol=0.0)
terrain = workloads.random_terrains()
B, nset = 256, 4
start, goal, mid = workloads.mpc_goals(B, seed=5, terrains=terrain)
gstep = goal - start[:, 0:3]
per = B // nset
Ws = []
for j in range(nset):
    P = Planner(cfg, max_batch=per, device=0); P.set_heightfields(terrain[0], terrain[1])
    sl = slice(j * per, (j + 1) * per)
    Ws.append(ShiftedWindows(P, start[sl], gstep[sl], mid[sl], advance=2.5, x_range=(0.0, 2.2), stream=torch.cuda.Stream(dev)))
def run(W, K, out):
    tb = tw = tp = 0.0
    for _ in range(K):
        t0 = time.perf_counter(); W.begin(); t1 = time.perf_counter(); W.P.wait(); t2 = time.perf_counter(); W.poll(); W.stream.synchronize(); t3 = time.perf_counter()
        tb += t1 - t0; tw += t2 - t1; tp += t3 - t2
    out.append((tb / K, tw / K, tp / K))
for W in Ws: run(W, 3, [])
o = []; t0 = time.perf_counter(); run(Ws[0], 20, o); print("one set alone: %.2f ms per replan; begin %.2f wait %.2f poll+sync %.2f" % ((time.perf_counter() - t0) / 20 * 1e3, *(1e3 * v for v in o[0])))
o = []; th = [threading.Thread(target=run, args=(W, 20, o)) for W in Ws]
t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]
print("four sets, one thread each: %.2f ms per replan of all; per set begin/wait/poll ms:" % ((time.perf_counter() - t0) / 20 * 1e3), [tuple(round(1e3 * v, 2) for v in x) for x in o])
print("iterations of the last replans:", [int(W.iters.max()) for W in Ws])
# one host thread for all four sets (begin / poll in turn)
K = 20
t0 = time.perf_counter(); left = [K] * len(Ws)
while any(left):
    for j, W in enumerate(Ws):
        if left[j] and W.poll():
            W.begin(); left[j] -= 1
while not all(W.poll() for W in Ws): pass
torch.cuda.synchronize()
print("four sets, ONE host thread: %.2f ms per replan of all" % ((time.perf_counter() - t0) / K * 1e3))
