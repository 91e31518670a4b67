# round 6: latency of the blocking call the reference makes (scripts/main.py:49-57, 90-92: docker exec ./main + docker cp of the CSV) through
# the drop-in boundary -- LocalPlanner.solve(args) from the flag dictionary to the 37-column CSV on disk, one plan at a time, and the
# reference's own batch of 32 probes (QTOS/generateHeightField.py:344-386) through solve_batch
import os, sys, time, json, tempfile
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from qtos_amd import csvio
from qtos_amd.planner import LocalPlanner
g = np.load(os.path.join(ROOT, "tests", "golden", "gv1.npz"))
inp = json.loads(str(g["inputs"]))
def args_for(dx, dy):
    return {"-g": [inp["g"][0] + dx, inp["g"][1] + dy, inp["g"][2]], "-s": inp["s"], "-s_ang": [0, 0, 0], "-e1": inp["ee"][0], "-e2": inp["ee"][1],
            "-e3": inp["ee"][2], "-e4": inp["ee"][3], "-t": 3.756, "-resolution": 0.01, "scripts": {}}
rng = np.random.default_rng(0)
out = {}
for name, kw in (("reference_compat walk (the reference's transcription)", {}),):
    t0 = time.perf_counter(); lp = LocalPlanner(max_batch=32); lp.planner(); t_create = time.perf_counter() - t0
    tmp = tempfile.mkdtemp()
    csv = os.path.join(tmp, "towr.csv")
    lat, t_plan, t_samp, t_csv = [], [], [], []
    for i in range(25):
        a = args_for(*rng.uniform(-0.2, 0.2, 2))
        t0 = time.perf_counter(); st = lp.solve(a, out_csv=csv); t1 = time.perf_counter()
        assert st == 0
        lat.append(t1 - t0)
        # the parts, once more on the same problem
        t0 = time.perf_counter(); lp.solve_batch([a], sample=False); t1 = time.perf_counter()
        lp.solve_batch([a]); t2 = time.perf_counter()
        csvio.write_csv(csv, lp.last["rows"][0]); t3 = time.perf_counter()
        t_plan.append(t1 - t0); t_samp.append((t2 - t1) - (t1 - t0)); t_csv.append(t3 - t2)
    med = lambda v: 1e3 * float(np.median(v[5:]))
    print("%s: planner creation %.2f s (once); one plan, flags -> CSV on disk: median %.2f ms (solve %.2f ms incl. transfers, 1 kHz sampling %.2f ms, CSV text %.2f ms), %d iterations" %
          (name, t_create, med(lat), med(t_plan), med(t_samp), med(t_csv), int(lp.last["iters"][0])))
    batch = [args_for(*rng.uniform(-0.2, 0.2, 2)) for _ in range(32)]
    lp.solve_batch(batch)
    ts = []
    for i in range(10):
        t0 = time.perf_counter(); sts = lp.solve_batch(batch); ts.append(time.perf_counter() - t0)
    print("   the reference's batch of 32 probes (solve_batch, rows sampled, no files): median %.2f ms, statuses all 0: %s" % (1e3 * float(np.median(ts)), all(s == 0 for s in sts)))
    lp.close()
print("reference: 0.75 s of Ipopt per plan in its own log (logs/towr_log.out:81-82: 1.34 plans/s) + docker exec / docker cp")
