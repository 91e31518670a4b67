#!/usr/bin/env python3
"""Headline benchmark: NLP solves/sec of the batched local planner on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: B planning problems per GPU (seeded synthetic
start/goal on the exp_1 flat heightfield, BASELINE.json configs[1]) solved to convergence on the
device -- initial guess, constraint/Jacobian assembly, KKT factor+solve, line search -- with the
inputs already resident in HBM, followed for N > 1 by the single all-gather that re-assembles the
plan batch on every rank.  `value` = converged plans of all ranks / wall time (max over ranks).

Extra objects on the JSON line:
  roofline     dominant kernel (k_kkt): algorithmic bytes per launch (SURVEY.md 8d formula on the
               planner's actual stage sizes) / average launch duration from HIP events.
  cpu_baseline the CPU oracle (a port of the same algorithm, 1 thread) on a bounded sample of the
               same workload, rank 0, N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
REF_LOG_PLANS_PER_S = 1.0 / 0.745  # logs/towr_log.out:81-82, unknown CPU -- not this metric's baseline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="plans per GPU per step")
    ap.add_argument("--transcription", default="knots100", choices=["knots100", "reference_compat", "knots200"])
    ap.add_argument("--workload", default="exp1_flat", choices=["exp1_flat", "exp5_step", "mixed", "mpc_random"],
                    help="mpc_random = BASELINE configs[4]: every step is one 50 Hz replan of all windows on randomized "
                         "heightfields (next start = the row 20 ms into the current plan, warm start = the current nodes); "
                         "use with --transcription knots200")
    ap.add_argument("--cpu-sample", type=int, default=96, help="plans timed on the CPU oracle (0 = skip)")
    ap.add_argument("--tol", type=float, default=None,
                    help="constraint-violation tolerance (default: the planner's 1e-4; the reference's Ipopt runs at ~1e-3)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight per GPU (each on its own planner handle + HIP stream, driven by its own "
                         "host thread): 2 lets the next batch use the CUs idled by early-converged problems. "
                         "Default 1 = the configuration BASELINE.json names")
    ap.add_argument("--max-iter", type=int, default=None,
                    help="Newton iteration limit per solve (default: the planner's 24). A fixed small budget is the "
                         "usual real-time setting for mpc_random: windows that need more come back with status 1 and "
                         "are not counted as solves")
    ap.add_argument("--init", default="straight_line", choices=["straight_line", "table"],
                    help="starting point of the solves: towr's straight-line guess (the reference's behaviour, default) or "
                         "the interpolation of a table of nominal plans solved once before the timed region "
                         "(Planner.build_init_table: 15 nominal goals, rest start) -- an amortised warm start, reported "
                         "separately from the headline")
    ap.add_argument("--episode", type=int, default=16,
                    help="mpc_random: replans per window before it is replaced by a fresh patch (cold start). The NLP "
                         "has no cost term, so a window replanned from its own 20 ms-ahead state drifts (base height) "
                         "until the start state leaves the range-of-motion box; episodes bound that")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per k_kkt launch from a separate rocprofv3 --pmc pass "
                         "(default: the newest profiles/*_pmc_hbm.json, collected with this same command)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    from qtos_amd.dist import gather_plans
    tol_kw = {} if args.tol is None else {"tol": args.tol}
    mpc = args.workload == "mpc_random"
    if args.max_iter is not None:
        tol_kw["max_iter"] = args.max_iter
    if mpc:
        tol_kw["honor_start_velocity"] = True    # a replan continues the motion it starts in
    cfg = {"knots100": PlannerConfig.knots100, "knots200": PlannerConfig.knots200,
           "reference_compat": PlannerConfig.reference_compat}[args.transcription](**tol_kw)
    B = args.batch
    P = Planner(cfg, max_batch=B, device=local_rank)
    d = P.dims
    terrain = None
    map_id_np = None
    if mpc:                          # BASELINE configs[4] shard: randomized heightfields, long-horizon goals
        maps, cell = workloads.random_terrains()
        P.set_heightfields(maps, cell)
        start_np, goal_np, map_id_np = workloads.mpc_goals(B, seed=5 + rank, terrains=(maps, cell))
    elif args.workload == "mixed":   # BASELINE configs[3] shard: exp_1 / exp_3 / exp_5 patches, one map index per problem
        maps, cell = workloads.mixed_terrains()
        P.set_heightfields(maps, cell)
        start_np, goal_np, map_id_np = workloads.mixed_goals(B, seed=2 + rank, terrains=(maps, cell))
    elif args.workload == "exp5_step":
        terrain = workloads.exp5_terrain()
        P.set_heightfields(terrain[0], terrain[1])
        start_np, goal_np = workloads.step_goals(B, seed=1 + rank, terrain=terrain)
    else:
        hxy, cell = workloads.exp1_terrain()   # 40 x 20 cells of zeros: the terrain path is live
        P.set_heightfields(hxy, cell)
        start_np, goal_np = workloads.flat_goals(B, seed=rank)   # weak scaling: B plans per GPU

    if args.init == "table":
        P.build_init_table()
    # inputs and outputs resident in HBM before the timed region
    start = torch.as_tensor(start_np, dtype=torch.float64, device=dev).contiguous()
    goal = torch.as_tensor(goal_np, dtype=torch.float64, device=dev).contiguous()
    nodes = torch.empty((B, d.n_vars), dtype=torch.float64, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    viol = torch.empty((B,), dtype=torch.float64, device=dev)
    map_id = None if map_id_np is None else torch.as_tensor(map_id_np, dtype=torch.int32, device=dev).contiguous()
    stream = torch.cuda.current_stream(dev)
    # receding window (mpc_random): ping-pong node buffers (solution k is the warm start of k+1), the two
    # CSV rows the next start state is read from, and the number of converged replans
    nodes_prev = torch.empty_like(nodes) if mpc else None
    rows2 = torch.empty((B, 2, 37), dtype=torch.float64, device=dev) if mpc else None
    t0_dev = torch.zeros((B,), dtype=torch.float64, device=dev) if mpc else None
    mpc_state = {"have_warm": False, "solved": torch.zeros((), dtype=torch.int64, device=dev), "k": 0, "cold": 0}
    start0 = start.clone() if mpc else None

    def step():
        nonlocal nodes, nodes_prev
        warm_ptr = None
        if mpc and mpc_state["k"] % max(args.episode, 1) == 0:   # new episode: fresh patches, cold start
            start.copy_(start0)
            mpc_state["have_warm"] = False
            mpc_state["cold"] += 1
        if mpc and mpc_state["have_warm"]:
            nodes, nodes_prev = nodes_prev, nodes
            warm_ptr = nodes_prev.data_ptr()
        rc = P.lib.qtos_plan_batch_device(P.h, B, start.data_ptr(), goal.data_ptr(),
                                          None if map_id is None else map_id.data_ptr(), warm_ptr,
                                          nodes.data_ptr(), status.data_ptr(), iters.data_ptr(),
                                          viol.data_ptr(), C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError("qtos_plan_batch_device failed: %d %s" % (rc, P.lib.qtos_last_error(P.h)))
        if mpc:   # the window moves on by 20 ms: CSV row 1 at 50 Hz, columns 1..24 = the next start vector
            rc = P.lib.qtos_sample_csv_device(P.h, B, nodes.data_ptr(), t0_dev.data_ptr(), C.c_double(50.0), 2,
                                              rows2.data_ptr(), C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError("qtos_sample_csv_device failed: %d" % rc)
            start.copy_(rows2[:, 1, 1:25])
            mpc_state["solved"] += (status == 0).sum()
            mpc_state["have_warm"] = True
            mpc_state["k"] += 1
        if world > 1:
            return gather_plans(nodes, status, B * world)
        return nodes, status

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # optional: several batches in flight (same inputs, separate planner handles / streams / outputs)
    lanes = []
    if args.inflight > 1:
        from concurrent.futures import ThreadPoolExecutor
        for _ in range(args.inflight):
            Pl = Planner(cfg, max_batch=B, device=local_rank)
            if args.workload == "mixed":
                Pl.set_heightfields(maps, cell)
            elif terrain is not None:
                Pl.set_heightfields(terrain[0], terrain[1])
            else:
                Pl.set_heightfields(hxy, cell)
            if args.init == "table":
                Pl.set_init_table(*P.init_table)
            lanes.append(dict(P=Pl, stream=torch.cuda.Stream(dev), nodes=torch.empty_like(nodes),
                              status=torch.empty_like(status), iters=torch.empty_like(iters), viol=torch.empty_like(viol)))
        pool = ThreadPoolExecutor(args.inflight)

        def lane_step(L):
            rc = L["P"].lib.qtos_plan_batch_device(L["P"].h, B, start.data_ptr(), goal.data_ptr(),
                                                   None if map_id is None else map_id.data_ptr(), None,
                                                   L["nodes"].data_ptr(), L["status"].data_ptr(), L["iters"].data_ptr(),
                                                   L["viol"].data_ptr(), C.c_void_p(L["stream"].cuda_stream))
            if rc != 0:
                raise RuntimeError("qtos_plan_batch_device failed: %d" % rc)
            L["stream"].synchronize()
            return int((L["status"] == 0).sum().item())

    for _ in range(args.warmup):
        step()
    sync()
    mpc_state["solved"].zero_()
    mpc_state["cold"] = 0
    kkt_s, kkt_n, tot_s, it_sum = 0.0, 0, 0.0, 0
    solved_inflight = None
    if lanes:
        for L in lanes:
            lane_step(L)
        sync()
        t0 = time.perf_counter()
        futs = [pool.submit(lane_step, lanes[i % len(lanes)]) for i in range(len(lanes))]
        done_steps, solved_inflight, nxt = 0, 0, len(lanes)
        while done_steps < args.steps:          # keep `inflight` batches running until K are done
            f = futs.pop(0)
            solved_inflight += f.result()
            done_steps += 1
            if nxt < args.steps:
                futs.append(pool.submit(lane_step, lanes[nxt % len(lanes)]))
                nxt += 1
        all_nodes, all_status = lanes[0]["nodes"], lanes[0]["status"]
    else:
        t0 = time.perf_counter()
    for _ in range(0 if lanes else args.steps):
        all_nodes, all_status = step()
        tm = P.timing()   # HIP events recorded on the launch stream around every k_kkt launch
        kkt_s += tm["kkt_seconds"]
        kkt_n += tm["kkt_launches"]
        tot_s += tm["total_seconds"]
        it_sum += tm["iterations"]
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    solved = (all_status == 0).sum().to(torch.float64).reshape(1)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    n_solved = int(solved.item())          # after the gather every rank sees the whole batch
    total_plans = B * world
    value = n_solved * args.steps / elapsed
    if solved_inflight is not None:
        value = solved_inflight / elapsed
    if mpc:
        ms = mpc_state["solved"].to(torch.float64).reshape(1)
        if world > 1:
            dist.all_reduce(ms)
        value = float(ms.item()) / elapsed
    st = status.cpu().numpy()
    itn = iters.cpu().numpy()

    out = {
        "metric": "NLP solves/sec (%d-knot SOLO12 gait, %g s horizon, converged to %s)" %
                  (d.n_dyn_times - 2, cfg.duration, ("%.0e" % cfg.tol).replace("e-0", "e-")),
        "value": round(value, 2), "unit": "plans/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "batch=%d/GPU %s goals, %s transcription (%d base polynomials, %d vars, %d "
                        "constraint rows), walk gait of the reference's golden plans" %
                        (B, {"exp1_flat": "exp_1 flat-ground", "exp5_step": "exp_5 step-climb",
                             "mixed": "mixed exp_1/exp_3/exp_5",
                             "mpc_random": "receding-window replans (20 ms shift, warm-started) on randomized exp_5 heightfields: ledge"}[args.workload],
                         args.transcription, d.n_base_nodes - 1, d.n_vars, d.n_cons),
            "global_batch": total_plans, "converged": n_solved, "iterations_max": int(itn.max()),
            "iterations_mean": round(float(itn.mean()), 2), "parallelism": "batch-shard x%d + 1 all-gather" % world,
            "kkt_unknowns": d.n_unknowns, "kkt_stages": d.n_stages, "front": d.front,
            "batches_in_flight": args.inflight, "max_iter": cfg.max_iter,
            "initial_guess": "towr straight line" if args.init == "straight_line" else
                             "interpolated table of %d nominal plans (solved before the timed region)" % (P.init_table[2].shape[0] * P.init_table[2].shape[1]),
        },
    }
    if mpc:
        out["config"]["replan_hz_per_window"] = round(args.steps / elapsed, 2)
        out["config"]["window_shift_s"] = 0.02
        out["config"]["replans_per_episode"] = args.episode
        out["config"]["cold_steps"] = mpc_state["cold"]
        out["config"]["warm_steps"] = args.steps - mpc_state["cold"]
    traffic, traffic_src = args.traffic_bytes, "--traffic-bytes"
    if traffic is None and args.transcription == "knots100" and args.workload == "exp1_flat" and B == 256:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")),
                       key=lambda f: (os.path.basename(f)[:3], "final" in f, f))   # newest round, its final pass
        if files:
            try:
                traffic = json.load(open(files[-1]))["k_kkt_traffic_bytes_per_launch"]["fetch_x2"]
                traffic_src = os.path.relpath(files[-1], ROOT)
            except Exception:
                traffic = None
    if kkt_n:
        avg = kkt_s / kkt_n
        alg_bytes = float(B) * d.kkt_algorithmic_bytes
        achieved = alg_bytes / avg / 1e9
        out["roofline"] = {
            "kernel": "k_kkt", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
            "bytes_per_launch": alg_bytes, "avg_launch_ms": round(1e3 * avg, 4), "launches": kkt_n,
            "fp64_tflops": round(B * d.kkt_flops / avg / 1e12, 3), "fp64_peak_tflops": 78.6,
            "kkt_share_of_device_time": round(kkt_s / max(tot_s, 1e-12), 3),
        }
    if rank == 0 and world == 1 and args.cpu_sample > 0 and args.workload not in ("mixed", "mpc_random"):
        from oracle.oracle import Oracle
        O = Oracle(cfg.oracle_dict(), height=None if terrain is None else terrain[0],
                   hcell=0.1 if terrain is None else terrain[1])
        n_s = min(args.cpu_sample, B)
        nodes_h = nodes.cpu().numpy()
        tc = time.perf_counter()
        ok, worst = 0, 0.0
        for b in range(n_s):
            s, g = start_np[b], goal_np[b]
            xo, info = O.solve(O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g))
            ok += int(info.status == 0)
            worst = max(worst, float(np.abs(xo - nodes_h[b]).max()))
        tc = time.perf_counter() - tc
        out["cpu_baseline"] = {
            "value": round(ok / tc, 3), "unit": "plans/s", "cores": 1, "kind": "port",
            "sample": "first %d problems of the same batch, oracle/qtos_oracle.c (same algorithm, skyline "
                      "LDL^T), 1 thread; max |gpu - cpu| nodes = %.1e" % (n_s, worst),
            "reference_log_plans_per_s": round(REF_LOG_PLANS_PER_S, 2),
            "reference_log_note": "Docker TOWR/Ipopt, logs/towr_log.out:81-82, unknown CPU, 1 thread; not runnable here",
        }
    if rank == 0:
        print(json.dumps(out))
    P.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
