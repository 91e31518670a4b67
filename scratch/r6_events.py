# round 6: what the per-kernel HIP events cost a batch: wall time per batch of 256 (trot, walk) with the events on / off
import sys, os, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
dev = torch.device("cuda", 0)
for gait in ("trot", "walk"):
    cfg = PlannerConfig.knots100(gait=gait)
    P = Planner(cfg, max_batch=256)
    sets = [workloads.flat_goals(256, seed=100 + i) for i in range(40)]
    S = torch.as_tensor(np.stack([s for s, g in sets]), device=dev); G = torch.as_tensor(np.stack([g for s, g in sets]), device=dev)
    nodes = torch.empty((256, P.n), dtype=torch.float64, device=dev); st = torch.empty(256, dtype=torch.int32, device=dev)
    it = torch.empty(256, dtype=torch.int32, device=dev); vi = torch.empty(256, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    def run(n):
        for i in range(n):
            rc = P.lib.qtos_plan_batch_device(P.h, 256, S[i % 40].data_ptr(), G[i % 40].data_ptr(), None, None, nodes.data_ptr(), st.data_ptr(), it.data_ptr(), vi.data_ptr(), C.c_void_p(stream.cuda_stream))
            assert rc == 0
    for rep in range(3):
        for on in (1, 0):
            P.set_kernel_events(on)
            run(10); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(200); torch.cuda.synchronize(); el = time.perf_counter() - t0
            print("%s events %s: %.4f ms per batch, %.0f plans/s" % (gait, "on " if on else "off", 1e3 * el / 200, 256 * 200 / el))
    P.close()
