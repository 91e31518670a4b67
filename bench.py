#!/usr/bin/env python3
"""Headline benchmark: NLP solves/sec of the batched local planner on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: this process launches
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a child
BEFORE it touches the GPU and exits with the child's code (the driver's own torchrun launch is used as it
is: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the environment).

One "step" = one pass of the hot path over one batch: B planning problems per GPU (seeded synthetic
start/goal on the exp_1 flat heightfield, BASELINE.json configs[1]; a fresh seeded batch every step,
all resident in HBM before the timed region) solved to convergence on the device -- initial guess,
constraint/Jacobian assembly, KKT factor+solve, line search --, followed for N > 1 by the single RCCL
all-gather that re-assembles the plan batch on every rank.  `value` = converged plans of all ranks /
wall time (max over ranks).  The default line is the TROT (the gait BASELINE.json's metric names); the
walk of the reference's golden plans is timed right behind it with the same --steps as the `walk` block.
Behind the W warm-up steps further untimed steps run until three in a row agree to 2 % (--settle, at
most 50; `warmup_extra` on the line), then EXACTLY K steps are timed.

Extra objects on the JSON line:
  step_ms, step_ms_series, gap_ms_per_step, device_ms_per_step, host_round_trips_per_step ...
               where a step's time went: the host's clock around every timed step (distribution + the first 25)
               and the planner's own HIP events split into kernels and the gaps between them.
               (The per-kernel events are recorded on every --events-every-th timed step, default 4: thirteen event packets
               between the kernels of a batch cost it 1.9 %; `steps_with_kernel_events` says how many steps carried them.)
  roofline     dominant kernel (k_kkt3 / k_kkt2): algorithmic bytes per launch (SURVEY.md 8d formula on the
               planner's actual stage sizes) / average launch duration from HIP events; counter-based
               matrix-pipe utilisation from the newest profiles/*_pmc_sq.json (same command).
  cpu_baseline the CPU oracle (a port of the same algorithm) on a bounded sample of the same workload,
               rank 0, N = 1 only: 1 thread and OpenMP over the batch on all host cores.
  parity       the metric's second half ("CoM L-inf vs TOWR"): the reference's two golden plans
               (tests/golden/gv1.npz, gv2.npz) re-solved on the GPU before the timed region.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Host threads that are not part of the measurement must not spin: the timed loop is ONE host thread submitting a batch and waiting
# for one word, inside a container whose CPU quota (cgroup cpu.max: 16 CPUs of time on the GPU boxes) is far below the CPUs it may run
# on (256).  OpenMP / MKL worker teams that numpy or torch started for the set-up (workloads, parity) busy-wait behind their last
# parallel region (KMP_BLOCKTIME 200 ms, libgomp's spin count); a few hundred spinning threads spend the group's quota within a CFS
# period and the kernel throttles the WHOLE group -- this thread included -- until the next one: 9 - 11 ms stalls every ~130 ms
# (profiles/r06_bench_steps500.json of the first pass: step 13 of the series; BENCH_r05's walk leg lost exactly one such stall in its
# 20 steps).  Set before numpy / torch are imported; `cgroup_throttled_ms` on the line says what the timed region still lost.
for _k, _v in (("OMP_WAIT_POLICY", "passive"), ("KMP_BLOCKTIME", "0"), ("GOMP_SPINCOUNT", "0"), ("MKL_NUM_THREADS", "16"), ("OPENBLAS_NUM_THREADS", "16")):
    os.environ.setdefault(_k, _v)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
REF_LOG_PLANS_PER_S = 1.0 / 0.745  # logs/towr_log.out:81-82, unknown CPU -- not this metric's baseline


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="plans per GPU per step")
    ap.add_argument("--transcription", default="knots100", choices=["knots100", "reference_compat", "knots200"])
    ap.add_argument("--gait", default=None, choices=["walk", "trot"],
                    help="trot = the diagonal-pair trot BASELINE.json's metric names (default: `metric` / `value` / `roofline` of "
                         "the line; not pinned by any reference artefact); walk = the gait of the reference's golden plans (the "
                         "default command times it right behind the trot, with the same --steps, as the line's `walk` block)")
    ap.add_argument("--workload", default="exp1_flat", choices=["exp1_flat", "exp5_step", "mixed", "mpc_random"],
                    help="mixed = BASELINE configs[3] (with --gpus 8: 2048 plans); mpc_random = configs[4]: every "
                         "step is one 50 Hz replan of all windows on randomized heightfields (use with --transcription knots200)")
    ap.add_argument("--same-batch", action="store_true",
                    help="replay ONE seeded batch every step instead of a fresh seeded batch per step")
    ap.add_argument("--cpu-sample", type=int, default=96, help="plans timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-parity", action="store_true", help="skip the golden-plan parity block")
    ap.add_argument("--tol", type=float, default=None,
                    help="constraint-violation tolerance (default: the planner's 1e-4; the reference's Ipopt runs at ~1e-3)")
    ap.add_argument("--inflight", type=int, default=None,
                    help="batches in flight per GPU (each on its own planner handle + HIP stream + host thread). "
                         "Default 1 = the configuration BASELINE.json names; mpc_random: the windows of a GPU run as this "
                         "many independent sets (default 4), every set at the pace of its own slowest window")
    ap.add_argument("--chord-tol", type=float, default=None,
                    help="violation below which an iterate reached by a full step is followed by a solve that re-uses "
                         "the factorisation (default: the planner's 1e-3; 0 = every iteration factors)")
    ap.add_argument("--max-iter", type=int, default=None, help="Newton iteration limit per solve (default: the planner's 24)")
    ap.add_argument("--init", default="straight_line", choices=["straight_line", "table"],
                    help="starting point of the solves: towr's straight-line guess (the reference's behaviour, default) or "
                         "the interpolation of a table of nominal plans solved before the timed region (reported separately)")
    ap.add_argument("--advance", type=float, default=2.5,
                    help="mpc_random: seconds into its newest plan at which a window's next plan starts (the reference's "
                         "f_steps = 2500 rows, scripts/main.py:177; moved on until all feet are in contact)")
    ap.add_argument("--warm", default="none", choices=["none", "shifted"],
                    help="mpc_random: starting point of a replan -- towr's straight-line guess (the reference's behaviour, default) or the "
                         "previous plan shifted to the hand-over time (qtos_shift_warm)")
    ap.add_argument("--full-system", action="store_true",
                    help="every row of the reference's NLP in the KKT system (PlannerConfig.reduce_base off: 2885 unknowns / 181 "
                         "stages instead of 1721 / 108 on the 100-knot transcription) -- the system rounds 1 and 2 solved")
    ap.add_argument("--full-swings", action="store_true",
                    help="the swing mid nodes as unknowns with their rule as equality rows (PlannerConfig.reduce_swing off: 1721 "
                         "unknowns / 108 stages instead of 1593 / 100 on the walk) -- the system of rounds 3 and 4")
    ap.add_argument("--plain-mu", action="store_true",
                    help="the barrier parameter shrinks by mu <- 0.2 mu (PlannerConfig.mu_superlinear off: rounds 1 - 4) instead of "
                         "Ipopt's monotone update mu <- max(tol, min(0.2 mu, mu^1.5))")
    ap.add_argument("--superlinear-mu", action="store_true",
                    help="mpc_random: keep Ipopt's update of the barrier parameter (the default of every other workload) instead of "
                         "the plain mu <- 0.2 mu the receding windows run with")
    ap.add_argument("--force-torchrun", action="store_true",
                    help="launch the ranks through torch.distributed.run even for --gpus 1 (exercises the child-process path "
                         "and the RCCL all-gather at world size 1)")
    ap.add_argument("--child-timeout", type=float, default=1800.0, help="seconds after which the launcher ends its torchrun child")
    ap.add_argument("--no-second-gait", "--no-walk", "--no-trot", dest="no_second_gait", action="store_true",
                    help="skip the second leg of the default headline run (the other gait -- the walk behind the trot --, timed with the "
                         "same --steps / --warmup right behind the first)")
    ap.add_argument("--settle", type=int, default=50,
                    help="adaptive warm-up: behind the --warmup untimed steps, further untimed steps until three in a row lie within "
                         "--settle-tol of each other, at most this many (0 = none); the line reports how many ran")
    ap.add_argument("--settle-tol", type=float, default=0.02)
    ap.add_argument("--events-every", type=int, default=4,
                    help="the planner's per-kernel HIP events (what `roofline.avg_launch_ms`, `gap_ms_per_step`, `kernel_ms_per_step` are "
                         "measured from) are recorded on every N-th timed step, the first included (qtos_set_kernel_events): thirteen "
                         "event packets between the kernels of a batch cost it 0.046 ms = 1.9 %% (scratch/r6_events.py); 1 = every step")
    ap.add_argument("--no-pattern", action="store_true",
                    help="qtos_set_pattern_speculation(0): the host reads the counts in front of every Newton iteration and launches it "
                         "(rounds 1 - 5) instead of queueing the handle's launch pattern at submit time")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per k_kkt launch from a separate rocprofv3 --pmc pass "
                         "(default: the newest profiles/*_pmc_hbm.json, collected with this same command)")
    a = ap.parse_args(argv)
    if a.gait is None:
        # the metric's gait on the metric's workload (flat exp_1 goals); the terrain workloads and the 200-knot receding windows keep
        # the walk their goals, schedules (PlannerConfig.knots200) and -m gpu tests are written for
        a.gait = "trot" if (a.workload == "exp1_flat" and a.transcription != "knots200") else "walk"
    return a


def kkt_kernel_name(planner):
    """The factor + solve kernel qtos_planner_create selected for this planner (qtos_kkt_kernel: what rocprofv3 lists)."""
    return planner.kkt_kernel()


def physical_cores():
    """Distinct (package, core) pairs among the CPUs this process may run on (SMT siblings counted once); None if unknown."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        seen = set()
        for c in cpus:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            with open(base + "physical_package_id") as f:
                pk = f.read().strip()
            with open(base + "core_id") as f:
                co = f.read().strip()
            seen.add((pk, co))
        return len(seen)
    except (OSError, AttributeError, ValueError):
        return None


def cgroup_throttled():
    """(periods in which this container was throttled, microseconds it spent throttled) so far -- cgroup v2 cpu.stat / v1 cpu.stat;
    None if unknown."""
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = dict(ln.split()[:2] for ln in open(f) if len(ln.split()) >= 2)
            us = d.get("throttled_usec")
            if us is None and "throttled_time" in d:
                us = int(d["throttled_time"]) / 1000.0
            return int(d.get("nr_throttled", 0)), float(us or 0.0)
        except (OSError, ValueError):
            continue
    return None


def cpu_quota():
    """CPU time this container is given, in CPUs (cgroup v2 cpu.max / v1 cfs quota); None = unlimited or unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else round(float(q) / float(per), 2)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def count_gpus_without_hip():
    """GPUs of this node, counted without loading the HIP runtime into this process: the KFD topology in sysfs (a node
    with SIMDs is a GPU), cut down to the visible-devices lists; if sysfs is not there, a short-lived child process asks
    torch.  (torch.cuda.device_count() in THIS process may fall back to hipGetDeviceCount, which initialises HIP -- and
    the launcher must not have touched the GPU when it starts its torchrun child.)"""
    have = None
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        have = n
    except (OSError, ValueError):
        pass
    if have is None:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        have = int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            have = min(have, len([x for x in v.split(",") if x.strip() != ""]))
    return have


def relaunch_under_torchrun(args):
    """--gpus N > 1 (or --force-torchrun) outside a torch.distributed launch: become the launcher.  This process makes
    no GPU call and does not load the HIP runtime (count_gpus_without_hip): the ranks run in a child process whose exit
    code becomes ours."""
    have = count_gpus_without_hip()
    if have < args.gpus:
        print("bench.py: --gpus %d but this node has %d GPU(s)" % (args.gpus, have), file=sys.stderr)
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    argv = [a for a in sys.argv[1:] if a != "--force-torchrun"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        rc = child.wait(timeout=args.child_timeout)
    except subprocess.TimeoutExpired:
        print("bench.py: the torchrun child did not finish within %.0f s; ending its process group" % args.child_timeout, file=sys.stderr)
        try:
            os.killpg(child.pid, 15)
            child.wait(timeout=20)
        except Exception:
            try:
                os.killpg(child.pid, 9)
            except Exception:
                pass
        rc = 124
    sys.exit(rc)


def golden_parity(Planner, PlannerConfig, device):
    """BASELINE.json metric, second half: the reference's golden plans on the reference transcription.
    P1 = max constraint violation of the reference's own nodes under this NLP; P2 = L-inf move of CoM / feet
    when warm-started at them (north_star tolerance 1e-3 m); P3 = L-inf to them from the reference's cold
    start (the NLP has no cost: one of a 299-dimensional family of feasible plans, reported, not gated)."""
    import numpy as np
    out = {"tolerance_m": 1e-3, "source": "tests/golden/gv1.npz, gv2.npz = test/data/traj/gait.csv, data/traj/towr.csv[1254:] of the reference"}
    P = Planner(PlannerConfig.reference_compat(), max_batch=1, device=device)
    worst = {"p1_max_residual": 0.0, "p2_com_linf": 0.0, "p2_ee_linf": 0.0, "p3_com_linf": 0.0, "p3_ee_linf": 0.0}
    for name in ("gv1", "gv2"):
        d = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        inp = json.loads(str(d["inputs"]))
        x = d["x"]
        start = np.concatenate([inp["s"], inp["s_ang"], np.ravel(inp["ee"]), inp["s_vel"], inp["s_ang_vel"]])[None]
        goal = np.array(inp["g"])[None]

        def linf(nodes):   # CoM positions [m] (base-lin nodes), foot nodes [m]; Euler angles [rad] reported with the CoM in the tests
            dd = np.abs(nodes - x)
            return float(dd[:306].reshape(-1, 6)[:, :3].max()), float(dd[612:752].max())
        nodes, st, it, viol = P.plan(start, goal, warm=x[None])
        p1 = float(P.trace(0)[0, 0])
        c2, e2 = linf(nodes[0])
        nodes, st3, it3, viol3 = P.plan(start, goal)
        c3, e3 = linf(nodes[0])
        out[name] = {"p1_max_residual": p1, "p2_com_linf": c2, "p2_ee_linf": e2, "p2_status": int(st[0]), "p2_iters": int(it[0]),
                     "p3_com_linf": c3, "p3_ee_linf": e3, "p3_status": int(st3[0]), "p3_iters": int(it3[0])}
        for k, v in (("p1_max_residual", p1), ("p2_com_linf", c2), ("p2_ee_linf", e2), ("p3_com_linf", c3), ("p3_ee_linf", e3)):
            worst[k] = max(worst[k], v)
    P.close()
    out.update({k: float("%.3g" % v) for k, v in worst.items()})
    out["p2_within_tolerance"] = bool(worst["p2_com_linf"] < 1e-3 and worst["p2_ee_linf"] < 1e-3)
    out["note"] = ("P2 (warm start at the reference's plan) meets the 1e-3 m tolerance; from the reference's cold start the "
                   "cost-free NLP returns another feasible plan (P3, centimetres away)")
    return out


def percentile(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(round(q * (len(xs) - 1))))] if xs else None


def settle(step_fn, sync_fn, max_steps, tol, agree=None):
    """Adaptive warm-up: untimed steps until three in a row lie within `tol` of each other (host clock around a step that ends
    with the device idle), at most max_steps.  agree(done) -> bool: with several ranks every rank runs the SAME number of steps
    (a step holds a collective): the decision to stop is the ranks' AND, taken after every step.  Returns the steps run."""
    import time as _t
    last, n = [], 0
    while n < max_steps:
        sync_fn()
        t0 = _t.perf_counter()
        step_fn()
        sync_fn()
        last.append(_t.perf_counter() - t0)
        n += 1
        done = len(last) >= 3 and max(last[-3:]) <= (1.0 + tol) * min(last[-3:])
        if agree is not None:
            done = agree(done)
        if done:
            break
    return n


class LegTimes:
    """Per-step records of one timed leg: the host's clock around every step and the planner's own events (qtos_last_timing,
    qtos_last_timing_detail) -- what lets a reader of the line tell a slow host from a slow kernel."""

    def __init__(self):
        self.step_s, self.kkt_s, self.kkt_n, self.tot_s, self.chord_s, self.chord_n = [], 0.0, 0, 0.0, 0.0, 0
        self.gap_s, self.solve_s, self.stepk_s, self.start_s, self.informed, self.at_submit, self.slots = 0.0, 0.0, 0.0, 0.0, 0, 0, 0
        self.pattern_calls = self.pattern_misses = 0
        self.have_detail = True
        self.n_ev = 0          # steps whose kernels carried HIP events (--events-every)
        self.wall_ev_s = 0.0

    def add(self, P, wall, sampled=True):
        self.step_s.append(wall)
        if not sampled:        # (a step without per-kernel events: its wall time only)
            return
        self.n_ev += 1
        self.wall_ev_s += wall
        if self.have_detail and hasattr(P.lib, "qtos_last_timing_detail"):
            d = P.timing_detail()   # HIP events recorded on the launch stream around every kernel of the call: ONE read-out per step
            self.kkt_s += d["kkt_seconds"]; self.kkt_n += d["kkt_launches"]; self.tot_s += d["total_seconds"]
            self.chord_s += d["chord_seconds"]; self.chord_n += d["chord_launches"]
            self.gap_s += d["gap_seconds"]; self.solve_s += d["solve_seconds"]; self.stepk_s += d["step_seconds"]; self.start_s += d["start_seconds"]
            self.informed += d["informed_launches"]; self.at_submit += d["slots_at_submit"]; self.slots += d["slots"]
            self.pattern_calls, self.pattern_misses = d["pattern_calls"], d["pattern_misses"]
        else:
            self.have_detail = False
            tm = P.timing()
            self.kkt_s += tm["kkt_seconds"]; self.kkt_n += tm["kkt_launches"]; self.tot_s += tm["total_seconds"]
            self.chord_s += tm.get("chord_seconds", 0.0); self.chord_n += tm.get("chord_launches", 0)

    def summary(self, elapsed):
        n = max(len(self.step_s), 1)
        ne = max(self.n_ev, 1)     # the event-based figures are means over the steps that carried events
        ms = [1e3 * x for x in self.step_s]
        out = {"step_ms": {"min": round(min(ms), 4), "p50": round(percentile(ms, 0.5), 4), "p90": round(percentile(ms, 0.9), 4), "max": round(max(ms), 4)} if ms else None,
               "step_ms_series": [round(x, 3) for x in ms[:25]],
               # steps that took more than 1.5 x the median (a host stall: the device time of a step does not show them) and what
               # they cost the timed region in all: on some leases a 9 - 11 ms stall arrives every ~130 ms (round 6, first pass)
               "stalled_steps": {"count": sum(1 for x in ms if x > 1.5 * percentile(ms, 0.5)),
                                 "excess_ms": round(sum(x - percentile(ms, 0.5) for x in ms if x > 1.5 * percentile(ms, 0.5)), 3)} if ms else None,
               # the timed region also holds, per step, the read-out of the planner's events (hipEventSynchronize + elapsed times)
               "timed_region_minus_steps_ms_per_step": round(1e3 * (elapsed - sum(self.step_s)) / n, 4)}
        if self.have_detail:
            out.update({
                "steps_with_kernel_events": self.n_ev,
                "device_ms_per_step": round(1e3 * self.tot_s / ne, 4),
                # between the kernels of a call: a launch slot's last event -> the next slot's first (the host reading the counts and
                # launching; ~0 between slots queued at submit time) + what lies between the last slot and the end of the call
                "gap_ms_per_step": round(1e3 * self.gap_s / ne, 4),
                "kernel_ms_per_step": {"k_start": round(1e3 * self.start_s / ne, 4), "solve": round(1e3 * self.solve_s / ne, 4), "k_step_and_counts": round(1e3 * self.stepk_s / ne, 4)},
                "launch_slots_per_step": round(self.slots / ne, 2), "slots_queued_at_submit_per_step": round(self.at_submit / ne, 2),
                "host_round_trips_per_step": round(self.informed / ne + 1, 2),   # (launches that waited for the counts + the end of the call)
                "pattern_calls": self.pattern_calls, "pattern_misses": self.pattern_misses,
                # the host's share of a step: submit latency + the wait for the word that says the batch is finished (steps with events)
                "host_ms_per_step_outside_device_time": round(1e3 * (self.wall_ev_s - self.tot_s) / ne, 4),
            })
        return out


def main():
    args = parse_args()
    if (args.gpus > 1 or args.force_torchrun) and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # under torchrun (also with one rank: RCCL at world size 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()   # the ranks RCCL actually sees
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    from qtos_amd.dist import gather_plans
    kw = {} if args.tol is None else {"tol": args.tol}
    mpc = args.workload == "mpc_random"
    if args.max_iter is not None:
        kw["max_iter"] = args.max_iter
    if args.inflight is None:
        args.inflight = 4 if mpc else 1
    if args.chord_tol is not None:
        kw["chord_tol"] = args.chord_tol
    if mpc:
        # the receding windows' solver settings are the PRODUCT's, in one place: PlannerConfig.receding_windows (no chord steps,
        # plain barrier update; config.py says why).  --superlinear-mu / --chord-tol override them for the A/B lines of DESIGN.md
        if args.superlinear_mu:
            kw["mu_superlinear"] = True
        rw = PlannerConfig.receding_windows(**kw)
        kw.setdefault("chord_tol", rw.chord_tol)
        kw.setdefault("mu_superlinear", rw.mu_superlinear)
    if args.gait == "trot":
        kw["gait"] = "trot"
    if args.full_system:
        kw["reduce_base"] = False
    if args.full_swings:
        kw["reduce_swing"] = False
    if args.plain_mu:
        kw["mu_superlinear"] = False
    if mpc and args.inflight > 1 and args.batch % args.inflight:
        raise SystemExit("--inflight must divide --batch for mpc_random (the windows are split into that many sets)")
    cfg = {"knots100": PlannerConfig.knots100, "knots200": PlannerConfig.knots200,
           "reference_compat": PlannerConfig.reference_compat}[args.transcription](**kw)
    B = args.batch
    P = Planner(cfg, max_batch=B, device=local_rank)
    if args.no_pattern:
        P.set_pattern_speculation(False)
    d = P.dims
    n_sets = 1 if (args.same_batch or mpc) else args.steps + args.warmup   # one seeded batch per step
    terrain = None

    def seeded(set_index):
        seed = 1000 * set_index + rank     # weak scaling: B plans per GPU, every rank / step its own seed
        if mpc:
            return workloads.mpc_goals(B, seed=5 + seed, terrains=terrain)
        if args.workload == "mixed":
            return workloads.mixed_goals(B, seed=2 + seed, terrains=terrain)
        if args.workload == "exp5_step":
            s, g = workloads.step_goals(B, seed=1 + seed, terrain=terrain)
            return s, g, None
        s, g = workloads.flat_goals(B, seed=seed)
        return s, g, None

    if mpc:                          # BASELINE configs[4] shard: randomized heightfields, long-horizon goals
        terrain = workloads.random_terrains()
    elif args.workload == "mixed":   # BASELINE configs[3] shard: exp_1 / exp_3 / exp_5 patches, one map index per problem
        terrain = workloads.mixed_terrains()
    elif args.workload == "exp5_step":
        terrain = workloads.exp5_terrain()
    else:
        terrain = workloads.exp1_terrain()   # 40 x 20 cells of zeros: the terrain path is live
    P.set_heightfields(terrain[0], terrain[1])
    sets = [seeded(i) for i in range(n_sets)]
    start_np, goal_np, map_id_np = sets[0]

    if args.init == "table":
        P.build_init_table()
    parity = None
    if rank == 0 and not args.no_parity:
        parity = golden_parity(Planner, PlannerConfig, local_rank)
    # inputs and outputs resident in HBM before the timed region
    start_all = torch.as_tensor(np.stack([s[0] for s in sets]), dtype=torch.float64, device=dev).contiguous()
    goal_all = torch.as_tensor(np.stack([s[1] for s in sets]), dtype=torch.float64, device=dev).contiguous()
    map_all = None if map_id_np is None else torch.as_tensor(np.stack([s[2] for s in sets]), dtype=torch.int32, device=dev).contiguous()
    nodes = torch.empty((B, d.n_vars), dtype=torch.float64, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    viol = torch.empty((B,), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    # receding windows (mpc_random): qtos_amd.replan.ShiftedWindows -- hand-over row, moving goal, shifted warm start
    windows = None
    if mpc:
        # the B windows of a GPU in `inflight` sets, each on its own planner handle + stream + host thread: a set waits
        # for its slowest window, the CUs it leaves idle meanwhile serve the other sets
        from concurrent.futures import ThreadPoolExecutor
        from qtos_amd.replan import ShiftedWindows
        gstep = goal_np - start_np[:, 0:3]
        nset, per = args.inflight, B // args.inflight
        windows = []
        for j in range(nset):
            Pj = P if j == 0 else Planner(cfg, max_batch=per, device=local_rank)
            if j:
                Pj.set_heightfields(terrain[0], terrain[1])
            sl = slice(j * per, (j + 1) * per)
            if nset > 1:
                Pj.set_kernel_events(False)   # (several sets: no per-kernel figures are read on this line, the event packets between the kernels are left out)
            windows.append(ShiftedWindows(Pj, start_np[sl], gstep[sl], map_id_np[sl], advance=args.advance, x_range=(0.0, 2.2), warm=args.warm,   # walk up and down the ledges
                                          stream=torch.cuda.current_stream(dev) if nset == 1 else torch.cuda.Stream(dev)))
        # several sets: one host thread per set (every replan queues ~40 small kernels -- sampling, hand-over rows, the
        # solve --: issued by ONE thread in turn for four sets the launches arrive too slowly, 31 ms per replan of all
        # windows instead of 16; ShiftedWindows.begin / poll exist for callers that prefer it)
        mpc_pool = ThreadPoolExecutor(nset) if nset > 1 else None
    gwork = None
    if use_dist:
        from qtos_amd.dist import gather_buffers
        gwork = gather_buffers(B * world, d.n_vars, world, torch.float64, dev)
    state = {"i": 0, "timed": 0}
    # HIP events around the collective of every timed step (recorded on the stream the solve is queued on)
    gather_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if use_dist else None
    solved_dev = torch.zeros((), dtype=torch.int64, device=dev)     # converged plans, accumulated on the device
    iters_dev = torch.zeros((), dtype=torch.int64, device=dev)

    def step():
        nonlocal nodes, status, iters
        if mpc:
            if mpc_pool is None:
                windows[0].replan()
            else:
                def one(Wj):
                    Wj.replan()
                    Wj.stream.synchronize()
                list(mpc_pool.map(one, windows))
            nodes = torch.cat([Wj.nodes for Wj in windows])
            status = torch.cat([Wj.status for Wj in windows])
            iters = torch.cat([Wj.iters for Wj in windows])
            solved_dev.add_((status == 0).sum())
            iters_dev.add_(iters.sum())
        else:
            i = state["i"] % n_sets
            state["i"] += 1
            rc = P.lib.qtos_plan_batch_device(P.h, B, start_all[i].data_ptr(), goal_all[i].data_ptr(),
                                              None if map_all is None else map_all[i].data_ptr(), None,
                                              nodes.data_ptr(), status.data_ptr(), iters.data_ptr(),
                                              viol.data_ptr(), C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError("qtos_plan_batch_device failed: %d %s" % (rc, P.lib.qtos_last_error(P.h)))
            # (converged plans and iterations are tallied on the device by the planner itself: qtos_plan_totals)
        if use_dist:
            # (the result is a view of the reused gather buffers: the next step overwrites it, nothing here keeps it)
            if gather_ev is not None and state["timed"] < len(gather_ev):
                e0, e1 = gather_ev[state["timed"]]
                state["timed"] += 1
                e0.record()
                out_ = gather_plans(nodes, status, B * world, work=gwork)
                e1.record()
                return out_
            return gather_plans(nodes, status, B * world, work=gwork)
        return nodes, status

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # optional: several batches in flight: a pool of planner handles (own workspace + stream each) fed by THIS thread
    # (qtos_amd.pool.PlannerPool: submit to a free lane, poll the others -- no host threads)
    lanes = None
    if args.inflight > 1 and not mpc:
        from qtos_amd.pool import PlannerPool
        def pool_done(lane):      # (runs on the lane's stream: one pair of counters per lane)
            lane.solved_acc.add_((lane.status[:lane.n] == 0).sum())
            lane.iters_acc.add_(lane.iters[:lane.n].sum())
        lanes = PlannerPool(cfg, n_lanes=args.inflight, max_batch=B, device=local_rank, heightfields=(terrain[0], terrain[1]), on_done=pool_done)
        for L in lanes.lanes:
            with torch.cuda.stream(L.stream):
                L.solved_acc = torch.zeros((), dtype=torch.int64, device=dev)
                L.iters_acc = torch.zeros((), dtype=torch.int64, device=dev)
            if args.init == "table":
                L.P.set_init_table(*P.init_table)

        def pool_step(i):
            i %= n_sets
            return lanes.submit(start_all[i], goal_all[i], None if map_all is None else map_all[i])

    for _ in range(args.warmup):
        step()
    sync()
    # adaptive warm-up (untimed; the K timed steps follow): a fresh box has been seen to run its first tens of milliseconds with
    # 0.5 ms per batch between the kernels (BENCH_r05) -- host clocks, first touches -- and a 20-step window must not average that in
    warmup_extra = 0
    if args.settle > 0 and not mpc and not (args.inflight > 1):
        agree = None
        if use_dist:   # (every rank runs the same number of steps: the collective inside a step must pair up)
            def agree(done):
                f = torch.tensor([1 if done else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(f, op=dist.ReduceOp.MIN)
                return bool(f.item())
        warmup_extra = settle(step, sync, args.settle, args.settle_tol, agree)
    sync()
    state["timed"] = 0
    solved_dev.zero_()
    iters_dev.zero_()
    P.totals(reset=True)
    leg = LegTimes()
    solved_inflight = None
    if lanes:
        for j in range(2 * args.inflight):
            pool_step(j)
        lanes.drain()
        sync()
        for L in lanes.lanes:
            with torch.cuda.stream(L.stream):
                L.solved_acc.zero_()
                L.iters_acc.zero_()
        sync()
        t0 = time.perf_counter()
        for j in range(args.steps):
            last_lane = pool_step(args.warmup + j)
        lanes.drain()
        sync()
        solved_inflight = sum(int(L.solved_acc.item()) for L in lanes.lanes)
        iters_dev.fill_(sum(int(L.iters_acc.item()) for L in lanes.lanes))
        all_nodes, all_status = last_lane.nodes, last_lane.status
    elif mpc and mpc_pool is not None:
        # the sets of windows are independent robots: every set runs its K replans on its own (no rendezvous between
        # sets after every replan: a set waits for ITS slowest window only); a step = one replan of every window
        def run_set(Wj):
            with torch.cuda.stream(Wj.stream):
                cnt = torch.zeros((), dtype=torch.int64, device=dev)   # (created on the stream that adds to it)
                for _ in range(args.steps):
                    Wj.replan()
                    cnt.add_((Wj.status == 0).sum())
            Wj.stream.synchronize()
            return int(cnt.item())
        t0 = time.perf_counter()
        solved_inflight = sum(mpc_pool.map(run_set, windows))
        all_nodes = torch.cat([Wj.nodes for Wj in windows])
        all_status = torch.cat([Wj.status for Wj in windows])
        if use_dist:
            all_nodes, all_status = gather_plans(all_nodes, all_status, B * world)
        iters = torch.cat([Wj.iters for Wj in windows])
    else:
        t0 = time.perf_counter()
    thr0 = cgroup_throttled()
    ev_every = max(1, args.events_every)
    for k_ in range(0 if (lanes or (mpc and mpc_pool is not None)) else args.steps):
        sampled = mpc or k_ % ev_every == 0
        if not mpc:
            P.set_kernel_events(sampled)
        ts = time.perf_counter()
        all_nodes, all_status = step()
        leg.add(P, time.perf_counter() - ts, sampled)
    if not mpc:
        P.set_kernel_events(True)
    sync()
    elapsed = time.perf_counter() - t0
    thr1 = cgroup_throttled()
    kkt_s, kkt_n, tot_s, chord_s, chord_n = leg.kkt_s, leg.kkt_n, leg.tot_s, leg.chord_s, leg.chord_n
    if not mpc and not lanes:
        tc, ti = P.totals()
        solved_dev.fill_(tc)
        iters_dev.fill_(ti)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    n_local = solved_dev.to(torch.float64).reshape(1) if solved_inflight is None else torch.tensor([float(solved_inflight)], dtype=torch.float64, device=dev)
    rank_rate = (n_local / elapsed).clone()           # this rank's own plans/s over its own clock
    rate_min, rate_max = rank_rate.clone(), rank_rate.clone()
    allgather_ms = None
    if use_dist:
        if gather_ev is not None and state["timed"] > 0 and not (lanes or (mpc and mpc_pool is not None)):
            allgather_ms = sum(e0.elapsed_time(e1) for e0, e1 in gather_ev[:state["timed"]]) / state["timed"]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n_local)
        dist.all_reduce(rate_min, op=dist.ReduceOp.MIN)
        dist.all_reduce(rate_max, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    n_solved = int(n_local.item())          # converged plans of all ranks over the K timed steps
    total_plans = B * world
    value = n_solved / elapsed
    itn = iters.cpu().numpy()

    out = {
        "metric": "NLP solves/sec (%d-knot SOLO12 %s gait, %g s horizon, converged to %s); CoM L-inf vs TOWR in `parity`" %
                  (d.n_dyn_times - 2, args.gait, cfg.duration, ("%.0e" % cfg.tol).replace("e-0", "e-")),
        "value": round(value, 2), "unit": "plans/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "warmup_extra": warmup_extra, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "timed_region_s": round(elapsed, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "batch=%d/GPU %s goals, %s transcription (%d base polynomials, %d vars, %d "
                        "constraint rows), %s" %
                        (B, {"exp1_flat": "exp_1 flat-ground", "exp5_step": "exp_5 step-climb",
                             "mixed": "mixed exp_1/exp_3/exp_5",
                             "mpc_random": "receding-window replans (hand-over row %.1f s into the newest plan as the reference's stitcher picks it, %s) on randomized exp_5 heightfields: ledge" % (args.advance, "cold start" if args.warm == "none" else "start = the previous plan shifted to the hand-over time")}[args.workload],
                         args.transcription, d.n_base_nodes - 1, d.n_vars, d.n_cons,
                         "walk gait of the reference's golden plans" if args.gait == "walk" else "diagonal-pair trot (config.TROT_UNNORMALISED, unpinned)"),
            "global_batch": total_plans, "plans_timed": total_plans * args.steps, "converged": n_solved,
            "batches": "one seeded batch replayed" if n_sets == 1 else "%d seeded batches, one per step" % n_sets,
            "iterations_max_last_step": int(itn.max()),
            "iterations_mean": round(float(iters_dev.item()) / max(B * args.steps, 1), 3) if (solved_inflight is None or lanes) else None,
            "parallelism": "batch-shard x%d + 1 %s" % (world, "RCCL all-gather" if use_dist else "all-gather (not launched under torch.distributed: single process)"),
            "kkt_unknowns": d.n_unknowns, "kkt_stages": d.n_stages, "front": d.front, "order_rule": d.order_rule, "reduce_base": bool(cfg.reduce_base), "reduce_swing": bool(cfg.reduce_swing), "mu_superlinear": bool(cfg.mu_superlinear),
            "batches_in_flight": args.inflight, "kernel_events": ("off: no per-kernel figures on this line" if (lanes or (mpc and args.inflight > 1)) else "every %d-th timed step" % max(1, args.events_every)),
            "max_iter": cfg.max_iter, "chord_tol": cfg.chord_tol, "chord_max": cfg.chord_max,
            "initial_guess": "towr straight line" if args.init == "straight_line" else
                             "interpolated table of %d nominal plans (solved before the timed region)" % (P.init_table[2].shape[0] * P.init_table[2].shape[1]),
        },
    }
    out["per_rank_plans_per_s"] = {"min": round(float(rate_min.item()), 1), "max": round(float(rate_max.item()), 1)}
    if leg.step_s:
        # where a step's time went, per step: distribution over the timed steps, the first 25 of them, the gaps between the kernels
        out.update(leg.summary(elapsed))
        if thr0 is not None and thr1 is not None:   # CFS throttling of the container during the timed steps (0 = none)
            out["cgroup_throttled_ms"] = round((thr1[1] - thr0[1]) / 1e3, 3)
            out["cgroup_throttled_periods"] = thr1[0] - thr0[0]
        out["launch_pattern"] = "off (--no-pattern): the host reads the counts in front of every iteration" if args.no_pattern else \
            "qtos_plan_submit queues the kernels the handle's last two calls both needed per launch slot; informed launches behind that prefix"
        if use_dist and leg.have_detail:   # a slow rank shows here, a slow collective in allgather_ms
            g = torch.tensor([out["gap_ms_per_step"], -out["gap_ms_per_step"], out["device_ms_per_step"], -out["device_ms_per_step"]], dtype=torch.float64, device=dev)
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            out["per_rank_gap_ms_per_step"] = {"min": round(-float(g[1]), 4), "max": round(float(g[0]), 4)}
            out["per_rank_device_ms_per_step"] = {"min": round(-float(g[3]), 4), "max": round(float(g[2]), 4)}
    if use_dist:
        out["allgather_ms"] = None if allgather_ms is None else round(allgather_ms, 4)
    if mpc:
        out["config"]["replan_hz_per_window"] = round(args.steps / elapsed, 2)
        out["config"]["windows_per_gpu"] = B
        out["config"]["hand_over_s_into_newest_plan"] = args.advance
        out["config"]["episode_resets"] = 0
        out["config"]["converged_fraction"] = round(n_solved / max(total_plans * args.steps, 1), 4)
    headline = args.transcription == "knots100" and args.workload == "exp1_flat" and B == 256
    prof_tag = "" if args.gait == "walk" else "trot_"      # profiles/rNN_pmc_*.json: the walk's command; rNN_trot_pmc_*.json: the trot's
    traffic, traffic_src = args.traffic_bytes, "--traffic-bytes"
    import glob

    def newest(pattern, kernel=None):
        # newest round's file of that name whose counters belong to the kernel that ran here (a profile of another kernel --
        # the trot's, or a kernel that is no longer the default -- says nothing about this launch)
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)),
                       key=lambda f: (os.path.basename(f)[:3], "final" in f, f))   # newest round, its final pass
        for f in reversed(files):
            if kernel is None:
                return f
            try:
                j = json.load(open(f))
                name = j.get("k_kkt_kernel") or (j.get("k_kkt") or {}).get("kernel") or ""
                if not name:   # (rounds 1 - 4: the walk's default kernel, recorded without its name)
                    name = "k_kkt2<128>"
                if kernel.replace(" ", "") in name.replace(" ", "").replace(",false", ""):
                    return f
            except Exception:
                pass
        return None
    kernel_here = kkt_kernel_name(P)
    if traffic is None and headline:
        f = newest("r[0-9][0-9]_%spmc_hbm.json" % prof_tag, kernel_here)
        if f:
            try:
                traffic = json.load(open(f))["k_kkt_traffic_bytes_per_launch"]["fetch_x2"]
                traffic_src = os.path.relpath(f, ROOT)
            except Exception:
                traffic = None
    if kkt_n:
        avg = kkt_s / kkt_n
        alg_bytes = float(B) * d.kkt_algorithmic_bytes
        full_sys = None
        if cfg.reduce_base:
            import dataclasses
            from qtos_amd.capi import analyze
            dfull, _ = analyze(dataclasses.replace(cfg, reduce_base=False))
            full_sys = {"kkt_unknowns": dfull.n_unknowns, "kkt_stages": dfull.n_stages, "bytes_per_launch": float(B) * dfull.kkt_algorithmic_bytes,
                        "achieved_if_priced_with_its_bytes": round(B * dfull.kkt_algorithmic_bytes / avg / 1e9, 2),
                        "frac_if_priced_with_its_bytes": round(B * dfull.kkt_algorithmic_bytes / avg / 1e9 / HBM_PEAK_GBS, 5)}
        achieved = alg_bytes / avg / 1e9
        out["roofline"] = {
            "kernel": kernel_here, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
            "bytes_per_launch": alg_bytes, "avg_launch_ms": round(1e3 * avg, 4), "launches": kkt_n,
            "fp64_tflops": round(B * d.kkt_flops / avg / 1e12, 3), "fp64_peak_tflops": 78.6,
            # the same launch priced with the bytes of the FULL KKT system of the reference's NLP (every acceleration-continuity
            # row with its multiplier: what rounds 1 and 2 factored): the reduced base solves that system's Newton step with
            # 1721 instead of 2885 unknowns, so `achieved` above (the SURVEY.md 8d formula on the stages actually run) counts fewer bytes
            "full_system": full_sys,
            "kkt_share_of_device_time": round(kkt_s / max(tot_s, 1e-12), 3),
            "launches_per_step": round(kkt_n / max(leg.n_ev, 1), 2),
            "events_every": ev_every,
            # k_chord: a solve that re-uses the factor panels of the preceding k_kkt2 launch (no assembly,
            # no factorisation); it reads the panels once more: same algorithmic read bytes, no write
            "chord_launches_per_step": round(chord_n / max(leg.n_ev, 1), 2),
            "chord_avg_launch_ms": round(1e3 * chord_s / chord_n, 4) if chord_n else None,
        }
        f = newest("r[0-9][0-9]_%spmc_sq.json" % prof_tag, kernel_here) if headline else None
        if f:
            try:
                out["roofline"]["counters"] = dict(json.load(open(f))["k_kkt"], source=os.path.relpath(f, ROOT))
            except Exception:
                pass
    if headline and world == 1 and rank == 0 and not args.no_second_gait and not lanes and args.init == "straight_line":
        # BASELINE.json's metric names a trot: the line's `value`.  The reference's committed plans -- what the oracle is pinned on
        # and the `parity` block re-solves -- are the WALK: the same batch size and goals with the other gait's schedule, timed
        # right behind the first leg with the same --steps / --warmup and the same adaptive warm-up.
        other = "walk" if args.gait == "trot" else "trot"
        cfg_t = PlannerConfig.knots100(gait=other, **{k: v for k, v in kw.items() if k != "gait"})
        Pt = Planner(cfg_t, max_batch=B, device=local_rank)
        if args.no_pattern:
            Pt.set_pattern_speculation(False)
        Pt.set_heightfields(terrain[0], terrain[1])
        dt_ = Pt.dims
        nodes_t = torch.empty((B, dt_.n_vars), dtype=torch.float64, device=dev)
        status_t, iters_t, viol_t = torch.empty_like(status), torch.empty_like(iters), torch.empty_like(viol)
        tstate = {"i": 0}

        def leg2_step():
            i = tstate["i"] % n_sets
            tstate["i"] += 1
            rc = Pt.lib.qtos_plan_batch_device(Pt.h, B, start_all[i].data_ptr(), goal_all[i].data_ptr(), None, None,
                                               nodes_t.data_ptr(), status_t.data_ptr(), iters_t.data_ptr(), viol_t.data_ptr(),
                                               C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError("qtos_plan_batch_device (%s) failed: %d" % (other, rc))
        dev_sync = lambda: torch.cuda.synchronize(dev)
        for i in range(args.warmup):
            leg2_step()
        dev_sync()
        extra2 = settle(leg2_step, dev_sync, args.settle, args.settle_tol) if args.settle > 0 else 0
        dev_sync()
        Pt.totals(reset=True)
        leg2 = LegTimes()
        thr2 = cgroup_throttled()
        tt0 = time.perf_counter()
        for i in range(args.steps):
            sampled = i % ev_every == 0
            Pt.set_kernel_events(sampled)
            ts = time.perf_counter()
            leg2_step()
            leg2.add(Pt, time.perf_counter() - ts, sampled)
        dev_sync()
        tel = time.perf_counter() - tt0
        tconv, titer = Pt.totals()
        tavg = leg2.kkt_s / max(leg2.kkt_n, 1)
        tag2 = "" if other == "walk" else "trot_"
        out[other] = {
            "value": round(tconv / tel, 2), "unit": "plans/s", "steps": args.steps, "warmup": args.warmup, "warmup_extra": extra2,
            "ms_per_step": round(1e3 * tel / args.steps, 4),
            "timed_region_s": round(tel, 4), "plans_timed": B * args.steps, "converged": int(tconv),
            "iterations_mean": round(titer / max(B * args.steps, 1), 3),
            "kkt_unknowns": dt_.n_unknowns, "kkt_stages": dt_.n_stages, "front": dt_.front, "order_rule": dt_.order_rule, "n_vars": dt_.n_vars,
            "gait": "walk of the reference's golden plans (config.REFERENCE_WALK_UNNORMALISED: the gait the oracle is pinned on and the `parity` block re-solves)" if other == "walk"
                    else "diagonal-pair trot (config.TROT_UNNORMALISED; not pinned by any reference artefact)",
            "roofline": {"kernel": kkt_kernel_name(Pt), "bound": "hbm", "achieved": round(B * dt_.kkt_algorithmic_bytes / tavg / 1e9, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(B * dt_.kkt_algorithmic_bytes / tavg / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                         "bytes_per_launch": float(B) * dt_.kkt_algorithmic_bytes, "avg_launch_ms": round(1e3 * tavg, 4), "launches": leg2.kkt_n,
                         "fp64_tflops": round(B * dt_.kkt_flops / tavg / 1e12, 3),
                         "chord_avg_launch_ms": round(1e3 * leg2.chord_s / leg2.chord_n, 4) if leg2.chord_n else None},
        }
        out[other].update(leg2.summary(tel))
        thr3 = cgroup_throttled()
        if thr2 is not None and thr3 is not None:
            out[other]["cgroup_throttled_ms"] = round((thr3[1] - thr2[1]) / 1e3, 3)
            out[other]["cgroup_throttled_periods"] = thr3[0] - thr2[0]
        # HBM traffic and SQ counters of this gait's kernel from the committed PMC passes of its own bench command (separate
        # rocprofv3 runs: profiles/rNN_[trot_]pmc_*.json)
        ft = newest("r[0-9][0-9]_%spmc_hbm.json" % tag2, kkt_kernel_name(Pt))
        if ft:
            try:
                out[other]["roofline"]["traffic"] = json.load(open(ft))["k_kkt_traffic_bytes_per_launch"]["fetch_x2"]
                out[other]["roofline"]["traffic_source"] = os.path.relpath(ft, ROOT)
            except Exception:
                pass
        ft = newest("r[0-9][0-9]_%spmc_sq.json" % tag2, kkt_kernel_name(Pt))
        if ft:
            try:
                out[other]["roofline"]["counters"] = dict(json.load(open(ft))["k_kkt"], source=os.path.relpath(ft, ROOT))
            except Exception:
                pass
        Pt.close()
    if parity is not None:
        out["parity"] = parity
    if rank == 0 and world == 1 and args.cpu_sample > 0 and args.workload not in ("mixed", "mpc_random"):
        from oracle.oracle import Oracle, oracle_dict
        O = Oracle(oracle_dict(cfg), height=None if args.workload == "exp1_flat" else terrain[0],
                   hcell=0.1 if args.workload == "exp1_flat" else terrain[1])
        n_s = min(args.cpu_sample, B)
        # the LAST timed batch is still on the device: its first n_s plans are re-solved on the host
        last = (state["i"] - 1) % n_sets if not lanes else (args.warmup + args.steps - 1) % n_sets
        s_np, g_np = sets[last][0], sets[last][1]
        nodes_h = (nodes if not lanes else last_lane.nodes).cpu().numpy()
        qs = [O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g) for s, g in zip(s_np[:n_s], g_np[:n_s])]
        # Thread counts: 1, 32 and every core this process may run on.  The first parallel passes of a process are dominated
        # by what has nothing to do with the solver -- the OpenMP team is created, every thread's malloc arena grows and its
        # pages are touched for the first time (round 3 timed exactly that: 256 threads 3.9 x one thread) --, so every thread
        # count gets two untimed passes first, and the timed pass gives each thread four problems (the batch repeated).
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        phys = physical_cores()
        qs_all = [O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g) for s, g in zip(s_np, g_np)]

        def timed(n_threads, n_problems):
            reps = [qs_all[i % len(qs_all)] for i in range(n_problems)]
            for _ in range(2):
                O.solve_batch(reps[:max(n_threads, 1)], n_threads=n_threads)
            tc = time.perf_counter()
            x_, inf_ = O.solve_batch(reps, n_threads=n_threads)
            el = time.perf_counter() - tc
            return sum(int(i.status == 0) for i in inf_) / el, x_, inf_

        n1 = max(1, n_s // 4)
        v1, _, _ = timed(1, n1)
        # a ladder of thread counts: a container's CPU quota may be far below the CPUs it may run on (round 4: 256 CPUs listed,
        # linear to 16 threads, slower beyond -- the factorisation's own timer grows 25 x at 256 threads: time slicing, not the
        # solver); `value` is the best rung, the whole ladder is on the line
        ladder = {}
        quota = cpu_quota()
        for nt in sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores}):
            ladder[nt] = timed(nt, 4 * nt if nt <= 64 else 2 * nt)[0]
        best = max(ladder, key=ladder.get) if ladder else 1
        vall = ladder.get(best, v1)
        _, xo, infos = timed(best, len(qs_all))
        worst = float(np.abs(xo[:len(qs_all)] - nodes_h).max()) if not lanes else None
        O.release_buffers(cores)   # (the per-thread Jacobians of the ladder: up to 30 MB x cores)
        out["cpu_baseline"] = {
            "value": round(vall, 3), "unit": "plans/s", "cores": best, "kind": "port",
            "kind_note": "port = the CHECKER (oracle/: complex-step Jacobians into a dense matrix per problem), not a tuned CPU solver: a stated baseline, the GPU / CPU ratio says nothing about kernel quality",
            "value_1_thread": round(v1, 3), "scaling_vs_1_thread": round(vall / max(v1, 1e-9), 2),
            "threads_ladder": {str(k): round(v, 1) for k, v in sorted(ladder.items())},
            "cpus_allowed": cores, "physical_cores": phys, "cgroup_cpu_quota": quota,
            "sample": "the %d problems of the last timed batch (repeated: four per thread), oracle/qtos_oracle.c (same algorithm, "
                      "skyline LDL^T), OpenMP over the problems after two untimed passes per thread count; value = the best thread "
                      "count of the ladder (%d threads); %d problems on 1 thread%s" %
                      (len(qs_all), best, n1, "" if worst is None else "; max |gpu - cpu| nodes = %.1e" % worst),
            "bound_note": "one problem per thread, no shared state; beyond the CPU time the container is given (cgroup quota, SMT "
                          "siblings) more threads only slice it -- the solver's own factorisation timer grows with the thread count there",
            "reference_log_plans_per_s": round(REF_LOG_PLANS_PER_S, 2),
            "reference_log_note": "Docker TOWR/Ipopt, logs/towr_log.out:81-82, unknown CPU, 1 thread; not runnable here",
        }
    if rank == 0:
        print(json.dumps(out))
    P.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
