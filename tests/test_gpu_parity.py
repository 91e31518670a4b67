"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Every check goes through the C ABI
(csrc/libqtos_planner.so via capi.py) and compares with the CPU oracle on the same seeded inputs,
with the committed golden plans, or through size-independent properties at the full batch size.

Floating-point tolerances (all double precision):
  constraint values / Jacobian entries  1e-10 abs   (same formulas, different summation order)
  one KKT solve                          5e-6 of the largest entry (achieved 6e-8 .. 1.6e-6 at cond ~5e6, recorded by
                                         the test; LAPACK vs the oracle's LDL^T: 1e-9 .. 1e-8)
  full NLP solve, nodes                  1e-6 abs    (north_star allows 1e-3 m; we hold 1e-6)
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_gv, oracle_problem, start_vector

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def planner(cfg):
    from qtos_amd.capi import Planner
    p = Planner(cfg, max_batch=256)
    yield p
    p.close()


@pytest.fixture(scope="module")
def planner_full(cfg):
    """The planner with every row of the reference's NLP in the KKT system (reduce_base and reduce_swing off): the tests that
    pin the internals -- the node-space Jacobian, one KKT solve against a dense factorisation, the factor panels -- are
    written for that system."""
    import dataclasses
    from qtos_amd.capi import Planner
    p = Planner(dataclasses.replace(cfg, reduce_base=False, reduce_swing=False), max_batch=256)
    yield p
    p.close()


def _random_problems(n, seed, hard=False):
    from qtos_amd import workloads
    start, goal = workloads.flat_goals(n, seed)
    if hard:
        rng = np.random.default_rng(seed + 100)
        goal[:, 0] += rng.uniform(0.0, 0.3, n)
        goal[:, 1] += rng.uniform(-0.15, 0.15, n)
    return start, goal


def _oracle_solve(O, start, goal, opts=None):
    xs, infos = [], []
    for s, g in zip(start, goal):
        q = O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g, (0, 0, 0), (0, 0, 0), 0.0)
        x, info = O.solve(q, opts=opts)
        xs.append(x)
        infos.append((info.status, info.iters, info.inf_pr))
    return np.array(xs), infos


def test_dims_match_reference_log(planner):
    d = planner.dims
    assert (d.n_vars, d.n_free, d.n_eq, d.n_ineq) == (1040, 1005, 706, 1024)
    assert (d.n_ineq_lower, d.n_ineq_both, d.n_ineq_upper) == (112, 816, 96)


def test_constraints_and_jacobian_match_oracle(planner_full, oracle, gv1):
    rng = np.random.default_rng(0)
    inp = gv1["inputs"]
    B = 6
    x = gv1["x"][None] + 0.02 * rng.standard_normal((B, planner_full.n))
    lo, hi = oracle.var_bounds(oracle_problem(oracle, inp))
    fx = lo == hi
    x[:, fx] = lo[fx]
    start = np.repeat(start_vector(inp)[None], B, 0)
    goal = np.repeat(np.array(inp["g"])[None], B, 0)
    g, J = planner_full.debug_eval(start, goal, x)
    rk, vf, _ = planner_full.structure()
    assert np.array_equal(vf == 0, fx)
    for b in range(B):
        go, Jo = oracle.constraints(x[b]), oracle.jacobian(x[b])
        Jo[:, fx] = 0
        Jo[rk == 0] = 0
        assert np.abs(g[b] - go).max() < 1e-10
        assert np.abs(J[b] - Jo).max() < 1e-10
        assert np.array_equal(J[b] != 0, Jo != 0)


def test_terrain_constraints_and_jacobian_match_oracle(gv1):
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.reference_compat(terrain_mode=0, reduce_base=False)   # bilinear: slopes enter the Jacobian; node-space Jacobian: the full system
    hxy, cell = workloads.exp5_terrain()
    O = Oracle(oracle_dict(cfg), height=hxy, hcell=cell)
    P = Planner(cfg, max_batch=4)
    P.set_heightfields(hxy, cell)
    rng = np.random.default_rng(5)
    inp = gv1["inputs"]
    x = gv1["x"][None] + 0.02 * rng.standard_normal((3, P.n))
    x[:, 612:752] += 0.3  # footholds spread over the ledges
    lo, hi = O.var_bounds(oracle_problem(O, inp))
    fx = lo == hi
    x[:, fx] = lo[fx]
    start = np.repeat(start_vector(inp)[None], 3, 0)
    goal = np.repeat(np.array(inp["g"])[None], 3, 0)
    g, J = P.debug_eval(start, goal, x)
    rk, _, _ = P.structure()
    for b in range(3):
        go, Jo = O.constraints(x[b]), O.jacobian(x[b])
        Jo[:, fx] = 0
        Jo[rk == 0] = 0
        assert np.abs(g[b] - go).max() < 1e-10
        assert np.abs(J[b] - Jo).max() < 1e-9
    P.close()


def test_kkt_solve_matches_dense_reference(planner_full, oracle, gv1, cfg):
    """One condensed KKT solve with random barrier weights vs LAPACK and vs the oracle's LDL^T."""
    import ctypes as C
    from oracle.oracle import lib as olib
    rng = np.random.default_rng(1)
    inp = gv1["inputs"]
    B = 4
    x = gv1["x"][None] + 0.01 * rng.standard_normal((B, planner_full.n))
    lo, hi = oracle.var_bounds(oracle_problem(oracle, inp))
    fx = lo == hi
    x[:, fx] = lo[fx]
    start = np.repeat(start_vector(inp)[None], B, 0)
    goal = np.repeat(np.array(inp["g"])[None], B, 0)
    rk, _, _ = planner_full.structure()
    I = rk == 2
    sig = np.zeros((B, planner_full.m))
    w = np.zeros((B, planner_full.m))
    sig[:, I] = 10.0 ** rng.uniform(-3, 3, (B, I.sum()))
    w[:, I] = rng.standard_normal((B, I.sum()))
    dx = planner_full.debug_newton(start, goal, x, sig, w)
    # a-posteriori residual of the same solve (k_residual: K applied from the stream, no use of the factorisation), then one
    # step of iterative refinement through the stored factorisation (k_chord) and its residual
    dx_same, res_plain = planner_full.debug_residual(B, refine=False)
    assert np.array_equal(dx_same, dx)
    dx_ref, res_ref = planner_full.debug_residual(B, refine=True)
    free, E, Ii = np.nonzero(~fx)[0], np.nonzero(rk == 1)[0], np.nonzero(I)[0]
    nf, nE = len(free), len(E)
    achieved = []
    for b in range(B):
        Jo, go = oracle.jacobian(x[b]), oracle.constraints(x[b])
        JE, JI = Jo[np.ix_(E, free)], Jo[np.ix_(Ii, free)]
        K = np.zeros((nf + nE, nf + nE))
        K[:nf, :nf] = cfg.delta_x * np.eye(nf) + JI.T @ (sig[b, Ii][:, None] * JI)
        K[nf:, :nf] = JE
        K[:nf, nf:] = JE.T
        K[nf:, nf:] = -cfg.eps_dual * np.eye(nE)
        rhs = np.concatenate([-JI.T @ w[b, Ii], -go[E]])
        full = np.linalg.solve(K, rhs)
        ref = full[:nf]
        # a reference good beyond double precision's cond * eps: LAPACK's solution refined with residuals in extended precision
        Kl, bl, hp = K.astype(np.longdouble), rhs.astype(np.longdouble), full.astype(np.longdouble)
        for _ in range(3):
            hp = hp + np.linalg.solve(K, (bl - Kl @ hp).astype(np.float64)).astype(np.longdouble)
        hp = hp[:nf].astype(np.float64)
        scale = np.abs(ref).max()
        sol = rhs.copy()
        Kc = np.ascontiguousarray(K)
        assert olib().qo_ldlt_solve_dense(nf + nE, Kc.ctypes.data_as(C.POINTER(C.c_double)),
                                          sol.ctypes.data_as(C.POINTER(C.c_double))) == 0
        # K is indefinite (cond 4e6 .. 6e6 here, barrier weights over six decades).  LAPACK and the oracle's LDL^T agree
        # to 1e-9 .. 1e-8 of the largest entry; the GPU's 16-pivot block elimination multiplies with the explicit inverse
        # of every pivot block and achieves 6e-8 .. 1.6e-6 (profiles/r02_kkt_accuracy.json): gate at
        # 5e-6 relative (2e-5 in round 1).
        cpu_spread = np.abs(sol[:nf] - ref).max()
        tol = 5e-6 * scale if cpu_spread < 2.5e-7 * scale else 20 * cpu_spread
        err_lapack, err_oracle = np.abs(dx[b, free] - ref).max(), np.abs(dx[b, free] - sol[:nf]).max()
        err_hp, err_hp_refined = np.abs(dx[b, free] - hp).max(), np.abs(dx_ref[b, free] - hp).max()
        achieved.append(dict(problem=b, cond=float(np.linalg.cond(K)), scale=float(scale), lapack_vs_oracle=float(cpu_spread / scale),
                             lapack_vs_extended=float(np.abs(ref - hp).max() / scale),
                             gpu_vs_lapack=float(err_lapack / scale), gpu_vs_oracle=float(err_oracle / scale),
                             gpu_vs_extended=float(err_hp / scale), gpu_refined_vs_extended=float(err_hp_refined / scale),
                             residual_rel=float(res_plain[b]), residual_rel_refined=float(res_ref[b])))
        assert err_lapack <= tol
        assert err_oracle <= tol
        assert np.all(dx[b, fx] == 0)
        # one refinement step: to 1e-9 of the largest entry (SURVEY.md section 7-5 asked 1e-10 of a CPU factorisation; LAPACK itself
        # is 1e-9 .. 1e-8 away from the extended-precision solution at this condition), residual down by orders of magnitude
        assert err_hp_refined <= 1e-9 * scale
        assert res_ref[b] <= 1e-3 * res_plain[b] or res_ref[b] <= 1e-13
    # the achieved errors (relative to the largest entry of the solution), for DESIGN.md: printed and, on the GPU box, kept
    print("KKT solve, achieved relative errors:", json.dumps(achieved))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump(achieved, open(os.path.join(out, "kkt_accuracy.json"), "w"), indent=1)


def test_reduced_base_system_gives_the_newton_step_of_the_full_system(cfg):
    """QtosParams.reduce_base: inside the KKT solve the base node values are replaced by the coefficients of a clamped cubic
    B-spline on the same knots (a basis of the C2 splines the acceleration-continuity rows describe): 2885 -> 1721 unknowns,
    181 -> 108 stages on the 100-knot transcription.  QtosParams.reduce_swing (round 5): the x, y, v_x, v_y of every swing's
    mid node are the linear image of the two neighbouring footholds that towr's swing rule makes them: 1721 -> 1593 unknowns,
    108 -> 100 stages.  At a point of both spaces (towr's straight-line guess with the mid nodes on the rule) one Newton step
    of every reduced system equals the step of the full system -- random barrier weights, same right-hand side -- to 1e-7 of
    its largest entry (the full system regularises the multipliers of the eliminated rows with eps_dual = 1e-8, the reduced
    ones have no such multipliers), the reduced systems' own residuals are at rounding level, and the recovered steps keep
    the acceleration continuity and the swing rule."""
    import dataclasses
    from qtos_amd import capi, workloads
    from qtos_amd.config import PlannerConfig
    rng = np.random.default_rng(3)
    B = 2
    s, g = workloads.flat_goals(B, 0)
    for base in (cfg, PlannerConfig.knots100()):
        out, dims = {}, {}
        sig = w = x0 = None
        for rb, sw in ((True, True), (True, False), (False, True), (False, False)):
            P = capi.Planner(dataclasses.replace(base, reduce_base=rb, reduce_swing=sw), max_batch=B)
            dims[(rb, sw)] = (P.dims.n_unknowns, P.dims.n_stages)
            if x0 is None:
                x0 = P.initial_guess(s, g)          # (reduce_swing: the mid nodes are on the swing rule)
            rk, _, _ = P.structure()
            I = rk == 2
            if sig is None:
                sig = np.zeros((B, P.m)); w = np.zeros((B, P.m))
                sig[:, I] = 10.0 ** rng.uniform(-2, 2, (B, I.sum())); w[:, I] = rng.standard_normal((B, I.sum()))
            dx = P.debug_newton(s, g, x0, sig, w)
            _, res = P.debug_residual(B, refine=False)
            dxr, res1 = P.debug_residual(B, refine=True)
            out[(rb, sw)] = dxr
            assert res1.max() < 1e-10
            if rb:
                assert res.max() < 1e-8          # (the full system's first solve: 1e-6, see test_kkt_solve_matches_dense_reference)
            P.close()
        full = out[(False, False)]
        scale = np.abs(full).max()
        for key in ((True, True), (True, False), (False, True)):
            assert dims[key][0] < dims[(False, False)][0] and dims[key][1] < dims[(False, False)][1]
            assert np.abs(out[key] - full).max() < 1e-7 * scale, key
        # the recovered step of the swing mid nodes is the swing rule applied to the footholds' step
        O_ = None
        from oracle.oracle import Oracle, oracle_dict
        O_ = Oracle(oracle_dict(dataclasses.replace(base, reduce_swing=True)))
        for b in range(B):
            assert np.abs(O_.project_swings(out[(True, True)][b]) - out[(True, True)][b]).max() < 1e-12 * max(scale, 1.0)
    assert dims == {(False, False): (2885, 181), (True, False): (1721, 108), (True, True): (1593, 100), (False, True): (2757, 173)}


def test_chord_step_kernel_matches_the_factorising_kernel(planner_full, oracle, gv1, cfg):
    """k_chord (QtosParams.chord_tol: reuse of the stored factorisation with a new right-hand side) solves the system
    of the preceding factorisation to the accuracy of that factorisation; and a solve with chord steps uses one
    factorisation less than one without, in the same number of iterations (the oracle applies the same rule)."""
    import dataclasses
    from qtos_amd.capi import Planner
    rng = np.random.default_rng(3)
    inp = gv1["inputs"]
    B = 4
    x = gv1["x"][None] + 0.01 * rng.standard_normal((B, planner_full.n))
    start = np.repeat(start_vector(inp)[None], B, 0)
    goal = np.repeat(np.array(inp["g"])[None], B, 0)
    rk, _, _ = planner_full.structure()
    I = rk == 2
    sig = np.zeros((B, planner_full.m))
    w = np.zeros((B, planner_full.m))
    sig[:, I] = 10.0 ** rng.uniform(-3, 3, (B, I.sum()))
    w[:, I] = rng.standard_normal((B, I.sum()))
    dx = planner_full.debug_newton(start, goal, x, sig, w)
    dc = planner_full.debug_chord(B)
    assert np.abs(dc - dx).max() <= 1e-7 * np.abs(dx).max()
    # full solves: 3 factorisations + 1 chord step instead of 4 factorisations on the benchmark goals
    s, g = _random_problems(12, seed=11)
    n1, st1, it1, _ = planner_full.plan(s, g)
    t1 = planner_full.timing()
    P0 = Planner(dataclasses.replace(cfg, chord_tol=0.0), max_batch=12)
    n0, st0, it0, _ = P0.plan(s, g)
    t0 = P0.timing()
    P0.close()
    assert (st1 == 0).all() and (st0 == 0).all() and np.array_equal(it1, it0)
    assert t1["chord_launches"] == 1 and t1["kkt_launches"] == t0["kkt_launches"] - 1 and t0["chord_launches"] == 0
    # (the two plans are both feasible to tol but not the same point: the NLP has no cost, the last step decides
    # where on the feasible set the iteration stops; the seeded-batch test pins the chord rule against the oracle)


def test_full_solve_matches_oracle_on_seeded_batch(planner, oracle):
    start, goal = _random_problems(12, seed=11)
    nodes, status, iters, viol = planner.plan(start, goal)
    xo, infos = _oracle_solve(oracle, start, goal)
    assert (status == 0).all() and all(i[0] == 0 for i in infos)
    assert [int(i) for i in iters] == [i[1] for i in infos]
    assert np.abs(nodes - xo).max() < 1e-6
    for b in range(len(start)):
        assert oracle.max_violation(nodes[b]) <= 1e-4 + 1e-9
        tr = planner.trace(b)
        assert tr.shape[0] == iters[b] + 1 and abs(tr[0, 0] - 19.4) < 0.5


def test_harder_goals_match_oracle(planner, oracle):
    start, goal = _random_problems(6, seed=21, hard=True)
    nodes, status, iters, viol = planner.plan(start, goal)
    xo, infos = _oracle_solve(oracle, start, goal)
    for b in range(len(start)):
        assert int(status[b]) == infos[b][0]
        if status[b] == 0:
            assert np.abs(nodes[b] - xo[b]).max() < 1e-5


@pytest.mark.parametrize("name", ["gv1", "gv2"])
def test_golden_inputs_p2_p3(planner, oracle, name):
    gv = load_gv(name)
    inp = gv["inputs"]
    start, goal = start_vector(inp)[None], np.array(inp["g"])[None]
    # P3 cold start: converges from the logged inf_pr 19.4 in O(10) iterations
    nodes, status, iters, viol = planner.plan(start, goal)
    assert status[0] == 0 and iters[0] <= 15 and viol[0] <= 1e-4
    assert float("%.2e" % planner.trace(0)[0, 0]) == 19.4
    xo, info = oracle.solve(oracle_problem(oracle, inp))
    assert np.abs(nodes[0] - xo).max() < 1e-6
    # P2 warm start at the reference's own plan: must stay within 1e-3 m of it
    nodes, status, iters, viol = planner.plan(start, goal, warm=gv["x"][None])
    assert status[0] == 0 and iters[0] <= 2
    d = np.abs(nodes[0] - gv["x"])
    assert d[:612].reshape(-1, 6)[:, :3].max() < 1e-3 and d[612:752].max() < 1e-3
    # sampled CSV rows of the warm-started plan equal the reference's CSV to 1e-3
    rows = planner.sample(nodes, inp["t0"])[0]
    err = np.abs(rows[gv["row_idx"]] - gv["rows"])
    if name == "gv2":
        err[0] = 0
    assert err[:, 0].max() < 1e-9 and err[:, 1:19].max() < 1e-3


def test_gv3_logged_solve_1(planner, oracle):
    """GV3: the inputs of the reference's logged solve #1 (logs/towr_log.out:8-29) and the 126 rows of its
    plan kept in data/traj/towr.csv[:1254] (tests/golden/gv3_partial.npz): GPU = oracle to 1e-6, the sampled
    rows carry the reference's time stamps, and the plan is one of the same family (the NLP has no cost)."""
    from test_oracle_golden import GV3_INPUTS
    d = np.load(os.path.join(GOLDEN, "gv3_partial.npz"))
    rows, idx = d["rows"], d["row_idx"]
    start, goal = start_vector(GV3_INPUTS)[None], np.array(GV3_INPUTS["g"])[None]
    nodes, status, iters, viol = planner.plan(start, goal)
    assert status[0] == 0 and viol[0] <= 1e-4 and float("%.2e" % planner.trace(0)[0, 0]) == 19.4
    xo, info = oracle.solve(oracle_problem(oracle, GV3_INPUTS))
    assert np.abs(nodes[0] - xo).max() < 1e-6 and int(iters[0]) == info.iters
    mine = planner.sample(nodes, 0.0)[0][2502 + idx]
    assert np.abs(mine[:, 0] - rows[:, 0]).max() < 5e-7
    assert np.abs(mine[:, 1:4] - rows[:, 1:4]).max() < 0.15 and np.abs(mine[:, 7:19] - rows[:, 7:19]).max() < 0.3


def test_r_flag_does_not_change_the_plans(gv1):
    """`-r` (probe: 5.0, QTOS/generateHeightField.py:373; default mode: 120 * tiles, scripts/main.py:119):
    accepted, reported back, and without influence on statuses or plans (flags.py); malformed values raise."""
    from qtos_amd.planner import LocalPlanner
    inp = gv1["inputs"]
    base = {'-s': inp["s"], '-s_ang': inp["s_ang"], '-e1': inp["ee"][0], '-e2': inp["ee"][1], '-e3': inp["ee"][2],
            '-e4': inp["ee"][3], '-g': inp["g"]}
    LP = LocalPlanner(max_batch=4)
    st0 = LP.solve_batch([dict(base)], sample=False)
    n0 = LP.last["nodes"].copy()
    st = LP.solve_batch([dict(base, **{'-r': 5.0}), dict(base, **{'-r': 120.0 * 3}), dict(base)], sample=False)
    assert st == [st0[0]] * 3 and LP.last["r"] == [5.0, 360.0, None]
    assert np.array_equal(LP.last["nodes"], np.repeat(n0, 3, 0))
    with pytest.raises(ValueError):
        LP.solve_batch([dict(base, **{'-r': -1.0})], sample=False)
    # mixed horizons in one batch: every call gets the planner of its own -duration
    st = LP.solve_batch([dict(base), dict(base, **{'-duration': 2.5}), dict(base)], sample=True)
    assert st[0] == st0[0] and LP.last["rows"][0].shape == (5001, 37) and LP.last["rows"][1].shape == (2501, 37)
    assert np.array_equal(LP.last["nodes"][0], n0[0]) and np.array_equal(LP.last["nodes"][2], n0[0])
    LP.close()


_GATHER_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch
import torch.distributed as dist
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from qtos_amd.dist import gather_plans
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
P = Planner(PlannerConfig.reference_compat(), max_batch=8)
start, goal = workloads.flat_goals(7, seed=4)
start[3, 0] = np.nan                    # one invalid problem: status 2 must survive the packing
nodes, status, iters, viol = P.plan(start, goal)
tn = torch.as_tensor(nodes, device="cuda")
ts = torch.as_tensor(status, device="cuda")
an, as_ = gather_plans(tn, ts, 7)
torch.cuda.synchronize()
assert an.shape == (7, P.n) and torch.equal(as_, ts) and status[3] == 2 and (np.delete(status, 3) == 0).all()
keep = np.arange(7) != 3
assert np.array_equal(an.cpu().numpy()[keep], nodes[keep])
P.close()
dist.destroy_process_group()
print("GATHER_OK")
"""


def test_gather_plans_on_device_world_1():
    """dist.gather_plans on device tensors with RCCL at world size 1 (the N > 1 path of bench.py; world 2 runs
    with gloo in tests/test_dist_cpu.py): the gathered batch equals the local one, status words intact.  In a
    child process with a time limit: a collective that cannot rendezvous must fail the test, not hang it."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _GATHER_SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "GATHER_OK" in r.stdout, r.stderr[-2000:]


def test_launch_pattern_is_bit_identical_and_survives_a_mismatch():
    """Round 6: qtos_plan_submit queues the LAUNCH PATTERN of the handle (per launch slot the solve kernels its last two calls
    both needed) instead of asking the host in front of every Newton iteration.  (1) Flat walk and trot batches at the BASELINE
    size take kkt, kkt, kkt, chord every time: from the third call on the whole solve is queued at submit time (four slots,
    no launch waits for the host) and the plans, statuses and iteration counts are bit for bit those of the informed loop.
    (2) A batch that does NOT follow the pattern -- the same handle is handed mixed-terrain problems, then warm starts that
    converge at once, then cold ones again -- still gives the informed loop's bits: a problem that finds the wrong solve kernel
    in a slot sits that launch out (k_step) and steps behind a later one; the handle reports that it happened."""
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    for gait in ("walk", "trot"):
        c = PlannerConfig.knots100(gait=gait)
        B = 256
        batches = [workloads.flat_goals(B, seed=300 + i) for i in range(5)]
        Pi = Planner(c, max_batch=B)
        Pi.set_pattern_speculation(False)
        ref = [Pi.plan(s, g) for s, g in batches]
        d = Pi.timing_detail()
        assert d["slots_at_submit"] == 1 and d["informed_launches"] == d["slots"] - 1 == 3     # the premise: 3 + 1 launches, the host in front of three
        Pi.close()
        Pp = Planner(c, max_batch=B)
        got = []
        for k, (s, g) in enumerate(batches):
            got.append(Pp.plan(s, g))
            d = Pp.timing_detail()
            if k >= 1:      # (the first call is trusted as it is: the second already follows its pattern)
                assert d["slots_at_submit"] == 4 and d["informed_launches"] == 0 and d["slots"] == 4, (gait, k, d)
        assert d["pattern_calls"] == 4 and d["pattern_misses"] == 0
        for a, b in zip(ref, got):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), gait
        Pp.close()
    # (2) the pattern breaks: flat problems (kkt, kkt, kkt, chord) teach the handle a pattern that the mixed-terrain batches
    # behind them do not follow (stragglers that factor a fourth time where the pattern launches k_chord only), warm starts
    # that converge at once leave every queued slot empty, and cold batches follow again
    c = PlannerConfig.knots100()
    B = 64
    maps, cell = workloads.mixed_terrains()
    mixed = [workloads.mixed_goals(B, seed=40 + i, terrains=(maps, cell)) for i in range(4)]
    flat = [workloads.flat_goals(B, seed=60 + i) + (np.zeros(B, np.int32),) for i in range(2)]     # (map 0 is exp_1's plane)
    Pi = Planner(c, max_batch=B)
    Pi.set_pattern_speculation(False)
    Pi.set_heightfields(maps, cell)
    Pp = Planner(c, max_batch=B)
    Pp.set_heightfields(maps, cell)
    seq = [(flat[0], False), (flat[1], False), (mixed[0], False), (flat[0], False), (flat[1], False), (mixed[1], False), (mixed[0], True),
           (flat[0], False), (flat[0], False), (mixed[2], False), (mixed[0], True), (mixed[0], True), (mixed[3], False), (mixed[3], False)]
    warm0 = None
    lens, misses = set(), []
    for k, ((s_, g_, m_), w) in enumerate(seq):
        a = Pi.plan(s_, g_, map_id=m_, warm=warm0 if w else None)
        b = Pp.plan(s_, g_, map_id=m_, warm=warm0 if w else None)
        if k == 2:
            warm0 = a[0].copy()               # (the solved plans of mixed[0])
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), (k, w)
        lens.add(int(a[2].max()))
        misses.append(Pp.timing_detail()["pattern_misses"])
    d = Pp.timing_detail()
    assert len(lens) > 2 and min(lens) <= 1   # the premise: calls of different lengths followed each other, the warm ones end behind their first step
    assert d["pattern_calls"] >= 8 and d["pattern_misses"] >= 1, (d, misses)      # ... and at least one found the pattern wrong
    Pi.close()
    Pp.close()


def test_pool_of_handles_equals_one_call_after_the_other(cfg):
    """The asynchronous boundary (qtos_plan_submit / qtos_plan_poll, qtos_amd.pool.PlannerPool): six batches of mixed
    terrain kept in flight on three planner handles by one host thread give bit for bit the plans, statuses and
    iteration counts of the same batches solved one call after the other; blind iterations (queued without the counts of
    unfinished problems) change nothing either; a second submit on a busy handle is refused (-5)."""
    import ctypes as C
    import torch
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.pool import PlannerPool
    B, NB = 64, 6
    maps, cell = workloads.mixed_terrains()
    dev = torch.device("cuda", 0)
    batches = [workloads.mixed_goals(B, seed=20 + i, terrains=(maps, cell)) for i in range(NB)]
    P = Planner(cfg, max_batch=B)
    P.set_heightfields(maps, cell)
    P.set_speculation(1)
    P.set_pattern_speculation(False)                       # the host looks at the counts in front of every iteration
    ref = [P.plan(s, g, map_id=m) for s, g, m in batches]
    P.set_speculation(8)
    again = [P.plan(s, g, map_id=m) for s, g, m in batches]   # from the second call on: as many blind iterations as the last call took
    for a, b in zip(ref, again):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # device-pointer path, ONE handle and ONE stream reused back to back without a stream synchronisation between the calls,
    # blind iterations on: a call that needs fewer iterations than its predecessor leaves launches queued behind its end
    # (first of all a warm start that k_start finds converged), and their late count words must not end the next call early
    # (sequence number in the word, qtos_planner.hip k_post_counts).  Results bit for bit those of the synchronous calls.
    warm0 = ref[0][0]
    ref_w = P.plan(batches[0][0], batches[0][1], map_id=batches[0][2], warm=warm0)
    P.set_speculation(1)
    ref_w1 = P.plan(batches[0][0], batches[0][1], map_id=batches[0][2], warm=warm0)
    assert all(np.array_equal(x, y) for x, y in zip(ref_w, ref_w1))
    assert int(ref_w[2].max()) < min(int(r[2].max()) for r in ref)     # the premise: the calls below differ in their length
    P.set_speculation(8)
    seq = [(0, None), (1, None), (0, warm0), (2, None), (0, warm0), (0, warm0), (3, None), (4, None), (0, warm0), (5, None)]
    dt = [[torch.as_tensor(np.ascontiguousarray(x), device=dev) for x in (s, g, m.astype(np.int32))] for s, g, m in batches]
    dwarm = torch.as_tensor(np.ascontiguousarray(warm0), device=dev)
    outs = [(torch.empty((B, P.n), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
             torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.float64, device=dev)) for _ in seq]
    st2 = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    for (i, w), o in zip(seq, outs):
        P.submit(B, dt[i][0].data_ptr(), dt[i][1].data_ptr(), dt[i][2].data_ptr(), None if w is None else dwarm.data_ptr(),
                 o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st2.cuda_stream)
        P.wait()                                          # (drives the polls; does not wait for the stream)
    st2.synchronize()
    for (i, w), o in zip(seq, outs):
        want = ref[i] if w is None else ref_w
        assert np.array_equal(o[0].cpu().numpy(), want[0]) and np.array_equal(o[1].cpu().numpy(), want[1]) and np.array_equal(o[2].cpu().numpy(), want[2])
    # a busy handle refuses a second call
    t = [torch.as_tensor(x, device=dev) for x in (batches[0][0], batches[0][1], batches[0][2].astype(np.int32))]
    out = (torch.empty((B, P.n), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
           torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.float64, device=dev))
    st = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    P.submit(B, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), None, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), st.cuda_stream)
    rc = P.lib.qtos_plan_submit(P.h, B, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), None, out[0].data_ptr(), out[1].data_ptr(),
                                out[2].data_ptr(), out[3].data_ptr(), C.c_void_p(st.cuda_stream))
    assert rc == -5
    P.wait()
    st.synchronize()
    assert np.array_equal(out[0].cpu().numpy(), ref[0][0]) and np.array_equal(out[1].cpu().numpy(), ref[0][1])
    P.close()
    got = {}

    def done(lane):
        got[lane.tag] = (lane.nodes[:lane.n].cpu().numpy(), lane.status[:lane.n].cpu().numpy(), lane.iters[:lane.n].cpu().numpy())
    pool = PlannerPool(cfg, n_lanes=3, max_batch=B, device=0, heightfields=(maps, cell), on_done=done)
    dev_in = [[torch.as_tensor(np.ascontiguousarray(x), device=dev) for x in (s, g, m.astype(np.int32))] for s, g, m in batches]
    torch.cuda.synchronize()      # the inputs are complete
    for rep in range(2):
        got.clear()
        for i, (s, g, m) in enumerate(dev_in):
            pool.submit(s, g, m, tag=i)
        pool.drain()
        assert sorted(got) == list(range(NB))
        for i in range(NB):
            assert np.array_equal(got[i][0], ref[i][0]) and np.array_equal(got[i][1], ref[i][1]) and np.array_equal(got[i][2], ref[i][2])
    pool.close()


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["QTOS_SWEEP_DS", "QTOS_SPEC_JAC"])
@pytest.mark.parametrize("workload", ["mixed", "trot", "knots200", "duration12"])
def test_work_moved_between_kernels_leaves_the_plans_bit_for_bit(switch, workload):
    """Round 4 moved two pieces of k_step without touching their arithmetic: the slack steps ds = Ji dx + (g - s) are formed by
    the waves that idle in the backward sweep of the KKT kernels (QTOS_SWEEP_DS, kkt2.hpp sweep_backward: same four-lane sums
    in the same order), and the first trial point of a Newton step's line search is evaluated together with its Jacobian
    (QTOS_SPEC_JAC).  With either switched off the planner takes the round-3 path: plans, statuses, iteration counts and
    violations must be the same bits -- on terrain (stragglers, chord steps, rejected chord steps), on the trot (k_kkt3),
    on 200 knots and on a longer horizon (fronts above 128 slots, short stages)."""
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    B = 64
    mid = None
    if workload == "mixed":
        c = PlannerConfig.knots100(); ter = workloads.mixed_terrains(); start, goal, mid = workloads.mixed_goals(B, seed=7, terrains=ter)
    elif workload == "trot":
        c = PlannerConfig.knots100(gait="trot"); ter = workloads.exp1_terrain(); start, goal = workloads.flat_goals(B, 3)
    elif workload == "knots200":
        c = PlannerConfig.knots200(); ter = workloads.random_terrains(); start, goal, mid = workloads.mpc_goals(B, seed=5, terrains=ter)
    else:
        c = PlannerConfig.reference_compat(reduce_base=True, duration=12.0); ter = workloads.exp1_terrain(); start, goal = workloads.flat_goals(B, 4)
    res = {}
    for v in ("0", "1"):
        os.environ[switch] = v
        try:
            P = Planner(c, max_batch=B)
        finally:
            del os.environ[switch]
        P.set_heightfields(ter[0], ter[1])
        res[v] = P.plan(start, goal, map_id=mid)
        P.close()
    assert (res["1"][1] == 0).mean() > 0.9
    assert all(np.array_equal(a, b) for a, b in zip(res["0"], res["1"]))


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["flat", "mixed"])
def test_kronecker_assembly_of_the_range_of_motion_blocks(workload):
    """QTOS_KRON=1 (experiment of round 4, k_kkt2<128, false, true>): every column of a range-of-motion block's Jacobian is
    a static multiple of a column of one of two 3 x 3 matrices (towr range_of_motion_constraint.cc: R(theta)^T (p - r)), so
    the entries of G' S G are products of two static weights with one of 33 sums per block (Symbolic::kron_meta, kernels.hpp
    kron_sums / kron_term) instead of three-term sums of their own.  Same iterates as the default assembly up to rounding
    (a few 1e-8 on the nodes), same iteration counts, and the oracle's plans within the usual tolerance."""
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd import workloads
    from qtos_amd.capi import Planner, build_flags
    from qtos_amd.config import PlannerConfig
    if not build_flags() & 1:
        pytest.skip("the Kronecker assembly lives in the experiment build (scratch/build.sh -DQTOS_EXPERIMENTS), not in the product library")
    cfg = PlannerConfig.knots100()      # (the 128-slot front of the benchmark: the experiment's only kernel)
    B = 64
    mid = None
    if workload == "flat":
        ter = workloads.exp1_terrain(); start, goal = workloads.flat_goals(B, 5)
    else:
        ter = workloads.mixed_terrains(); start, goal, mid = workloads.mixed_goals(B, seed=9, terrains=ter)
    res = {}
    for v in ("0", "1"):
        os.environ["QTOS_KRON"] = v
        try:
            P = Planner(cfg, max_batch=B)
        finally:
            del os.environ["QTOS_KRON"]
        P.set_heightfields(ter[0], ter[1])
        res[v] = P.plan(start, goal, map_id=mid)
        P.close()
    assert (res["1"][1] == res["0"][1]).all() and (res["1"][2] == res["0"][2]).all()
    ok = res["0"][1] == 0
    d = np.abs(res["1"][0][ok] - res["0"][0][ok]).max()
    assert 0.0 < d < 1e-6, d      # (> 0: the other path did run)
    if workload == "flat":
        O = Oracle(oracle_dict(cfg))
        xo, infos = _oracle_solve(O, start[:4], goal[:4])
        assert [int(i) for i in res["1"][2][:4]] == [i[1] for i in infos]
        assert np.abs(res["1"][0][:4] - xo).max() < 1e-6


@pytest.mark.gpu
def test_call_larger_than_the_gpu_is_cut_into_lanes_with_identical_plans(cfg):
    """A call of more problems than the GPU has compute units is served by several lanes of the handle (contiguous parts on
    streams of the planner, each with its own host-driven Newton loop: the late iterations of one part's stragglers run beside
    the other parts' full grids -- the reference's queue of probes, QTOS/generateHeightField.py:375-377, inside ONE call).
    Mixed terrain, 1024 problems: plans, statuses and iteration counts bit for bit those of the same call on one lane
    (QTOS_LANES=1) and of four calls of 256; the asynchronous form (device pointers, submit / wait on a caller's stream, no
    synchronisation in between) gives the same."""
    import torch
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    B = 1024
    maps, cell = workloads.mixed_terrains()
    start, goal, mid = workloads.mixed_goals(B, seed=31, terrains=(maps, cell))
    res = {}
    for lanes in ("4", "1"):
        os.environ["QTOS_LANES"] = lanes
        try:
            P = Planner(cfg, max_batch=B)
        finally:
            del os.environ["QTOS_LANES"]
        P.set_heightfields(maps, cell)
        res[lanes] = P.plan(start, goal, map_id=mid)
        if lanes == "4":
            dev = torch.device("cuda", 0)
            t = [torch.as_tensor(np.ascontiguousarray(x), device=dev) for x in (start, goal, mid.astype(np.int32))]
            out = (torch.empty((B, P.n), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
                   torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.float64, device=dev))
            st = torch.cuda.Stream(dev)
            torch.cuda.synchronize()
            for _ in range(2):     # back to back on one stream
                P.submit(B, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), None, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), st.cuda_stream)
                P.wait()
            st.synchronize()
            res["async"] = (out[0].cpu().numpy(), out[1].cpu().numpy(), out[2].cpu().numpy(), out[3].cpu().numpy())
            parts = [P.plan(start[i:i + 256], goal[i:i + 256], map_id=mid[i:i + 256]) for i in range(0, B, 256)]
            res["parts"] = tuple(np.concatenate([q[k] for q in parts]) for k in range(4))
        P.close()
    assert (res["4"][1] == 0).mean() > 0.95 and len(set(res["4"][2].tolist())) > 1      # stragglers exist
    for other in ("1", "async", "parts"):
        assert all(np.array_equal(a, b) for a, b in zip(res["4"], res[other])), other


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [["--workload", "mixed", "--steps", "10", "--warmup", "2"],
                                   ["--transcription", "knots200", "--workload", "mpc_random", "--steps", "6", "--warmup", "2"]])
def test_bench_on_two_gpus_when_the_node_has_them(flags):
    """The first box with more than one GPU exercises what world size 1 cannot: RCCL between ranks, ragged shards of the
    mixed batch (BASELINE configs[3]) and the gather of the receding windows (configs[4]).  `bench.py --gpus 2` starts its two
    ranks as a child process (torch.distributed.run) and prints ONE JSON line with n_gpus = 2 and the whole job's rate.
    Skipped on a one-GPU box."""
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-sample", "0", "--no-parity", "--no-trot",
                        "--child-timeout", "900"] + flags, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert "RCCL all-gather" in out["config"]["parallelism"]
    pr = out["per_rank_plans_per_s"]
    assert 0 < pr["min"] <= pr["max"]


@pytest.mark.gpu
def test_bench_relaunches_itself_under_torchrun_on_one_gpu():
    """bench.py --force-torchrun: the launcher path of `--gpus N` (bench.py starts `python -m torch.distributed.run`
    as a CHILD before it touches the GPU and exits with the child's code) exercised with one rank: one JSON line,
    RCCL at world size 1, the all-gather timed by HIP events, and a rate within a few per cent of the plain run.
    A request for more GPUs than the node has ends with exit code 2 and a message, before anything is spawned."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--cpu-sample", "0", "--no-parity", "--no-trot"]
    r = subprocess.run(common + ["--force-torchrun", "--child-timeout", "600"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    tr = json.loads(lines[0])
    assert tr["n_gpus"] == 1 and "RCCL all-gather" in tr["config"]["parallelism"]
    assert tr["allgather_ms"] is not None and 0.0 < tr["allgather_ms"] < 5.0
    assert tr["config"]["converged"] == tr["config"]["plans_timed"] == 20 * 256
    assert tr["per_rank_plans_per_s"]["min"] <= tr["per_rank_plans_per_s"]["max"]
    r2 = subprocess.run(common, env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    plain = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][0])
    assert "allgather_ms" not in plain and plain["timed_region_s"] > 0
    assert tr["value"] > 0.85 * plain["value"]         # (measured: within 3 % on an idle box over 100 steps; 20 steps on a shared one)
    import torch
    r3 = subprocess.run(common + ["--gpus", str(torch.cuda.device_count() + 1)], env=env, capture_output=True, text=True, timeout=120)
    assert r3.returncode == 2 and "GPU(s)" in r3.stderr


def test_sampler_matches_oracle_and_reference_csv(planner, oracle, gv1):
    rows = planner.sample(gv1["x"][None], gv1["inputs"]["t0"])[0]
    ro = oracle.sample(gv1["x"], gv1["inputs"]["t0"])
    assert rows.shape == (5001, 37)
    assert np.abs(rows - ro).max() < 1e-12
    err = np.abs(rows[gv1["row_idx"]] - gv1["rows"])
    assert err[:, 1:19].max() < 2e-5 and err[:, 25:].max() < 2e-4


def test_full_batch_properties_and_determinism(planner, cfg):
    """BASELINE configs[1]-sized batch on the reference transcription: size-independent properties."""
    from qtos_amd import workloads
    start, goal = workloads.flat_goals(256, seed=0)
    planner.totals(reset=True)
    n1, s1, i1, v1 = planner.plan(start, goal)
    n2, s2, i2, v2 = planner.plan(start, goal)
    assert np.array_equal(n1, n2) and np.array_equal(s1, s2)      # bitwise reproducible
    # the planner's own running totals (qtos_plan_totals, bench.py's count of converged plans)
    assert planner.totals() == (int((s1 == 0).sum() + (s2 == 0).sum()), int(i1.sum() + i2.sum()))
    assert planner.totals(reset=True)[0] == 512 and planner.totals() == (0, 0)
    assert (s1 == 0).all() and v1.max() <= cfg.tol and i1.max() <= 15
    rk, vf, _ = planner.structure()
    fixed = np.nonzero(vf == 0)[0]
    # fixed variables carry the inputs exactly: start state, goal xy, zero final velocities
    assert np.array_equal(n1[:, 0:3], start[:, 0:3]) and np.array_equal(n1[:, 306:309], start[:, 3:6])
    nb = planner.dims.n_base_nodes - 1
    assert np.array_equal(n1[:, 6 * nb:6 * nb + 2], goal[:, :2])
    assert len(fixed) == 35
    # batch order independence: a permuted batch gives the permuted result
    perm = np.random.default_rng(0).permutation(256)
    n3, _, _, _ = planner.plan(start[perm], goal[perm])
    assert np.array_equal(n3, n1[perm])
    # translation equivariance on flat ground: shifting start and goal in x shifts the plan
    sh = start.copy()
    sh[:, [0, 6, 9, 12, 15]] += 0.5
    gsh = goal.copy()
    gsh[:, 0] += 0.5
    n4, s4, _, _ = planner.plan(sh[:16], gsh[:16])
    xcols = np.zeros(planner.n, bool)
    xcols[np.arange(0, 306, 6)] = True
    for e in range(4):
        off = 612 + 35 * e
        for s in range(5):
            xcols[off + 8 * s] = True
        for s in range(4):
            xcols[off + 8 * s + 3] = True
    diff = n4 - n1[:16]
    assert np.abs(diff[:, xcols] - 0.5).max() < 1e-6 and np.abs(diff[:, ~xcols]).max() < 1e-6


def test_knots100_batch_matches_oracle(oracle):
    """BASELINE configs[1] transcription (100 base polynomials): GPU vs oracle on a seeded sample."""
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.knots100()
    P = Planner(cfg, max_batch=256)
    O = Oracle(oracle_dict(cfg))
    assert P.n == O.n == 1640 and P.m == O.m
    start, goal = workloads.flat_goals(256, seed=0)
    nodes, status, iters, viol = P.plan(start, goal)
    assert (status == 0).all() and viol.max() <= cfg.tol
    # 32 of the 256 problems (every 8th) against the oracle, OpenMP over the problems
    sel = np.arange(0, 256, 8)
    qs = [O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g) for s, g in zip(start[sel], goal[sel])]
    xo, infos = O.solve_batch(qs)
    assert all(i.status == 0 for i in infos)
    assert np.abs(nodes[sel] - xo).max() < 1e-6
    assert [int(i) for i in iters[sel]] == [i.iters for i in infos]
    P.close()


def test_step_terrain_batch(cfg):
    """BASELINE configs[2]: exp_5 heightfield, terrain constraint active: stance feet end up ON the
    terrain, swing apexes above it; checked against the oracle on the same terrain."""
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd import heightfield, workloads
    from qtos_amd.capi import Planner
    hxy, cell = workloads.exp5_terrain()
    start, goal = workloads.step_goals(64, seed=1, terrain=(hxy, cell))
    P = Planner(cfg, max_batch=64)
    P.set_heightfields(hxy, cell)
    nodes, status, iters, viol = P.plan(start, goal)
    ok = status == 0
    assert ok.mean() >= 0.9
    O = Oracle(oracle_dict(cfg), height=hxy, hcell=cell)
    lifted = 0
    for b in np.nonzero(ok)[0][:16]:
        assert O.max_violation(nodes[b]) <= 1e-4 + 1e-9
        for e in range(4):
            off = 612 + 35 * e
            for s in range(1, 5):
                p = nodes[b, off + 8 * s: off + 8 * s + 3]
                h = float(heightfield.height_at(hxy, cell, p[0], p[1], mode=cfg.terrain_mode))
                assert abs(p[2] - h) <= 1e-4
                lifted += h > 0.02
    assert lifted > 0   # some footholds really are on the ledges
    # the terrain makes the problem piecewise smooth: a solve whose iterates never sit near a cell
    # edge follows the oracle exactly, one that does may branch differently (both feasible)
    same = 0
    for b in range(8):
        s, g = start[b], goal[b]
        xo, info = O.solve(O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g, (0, 0, 0), (0, 0, 0), 0.0))
        if info.status == 0 and status[b] == 0 and info.iters == iters[b]:
            same += np.abs(nodes[b] - xo).max() < 1e-5
    assert same == 8   # (4 of 8 before the two-phase solve: free footholds near a cell edge branched differently; 7 the gate of round 2)
    P.close()


def test_local_planner_writes_reference_csv(tmp_path, gv1):
    from qtos_amd import csvio, flags
    from qtos_amd.planner import LocalPlanner
    inp = gv1["inputs"]
    args = {"-g": inp["g"], "-s": inp["s"], "-s_ang": [0, 0, 0], "-e1": inp["ee"][0], "-e2": inp["ee"][1],
            "-e3": inp["ee"][2], "-e4": inp["ee"][3], "-t": 3.756, "-resolution": 0.01, "scripts": {}}
    lp = LocalPlanner(max_batch=4)
    out = tmp_path / "towr.csv"
    assert lp.solve(args, out_csv=str(out)) == 0
    rows = csvio.read_csv(str(out))
    assert rows.shape == (5001, 37)
    assert rows[0, 0] == 3.756 and abs(rows[-1, 0] - 8.756) < 1e-9
    assert np.allclose(rows[0, 1:4], inp["s"]) and np.allclose(rows[0, 7:19], np.ravel(inp["ee"]))
    assert np.allclose(rows[-1, 1:3], inp["g"][:2], atol=1e-6)
    # first line is printed like the reference prints it (%g)
    first = open(out).readline().strip().split(",")
    assert first[0] == "3.756" and first[3] == "0.24"
    # batch form returns one exit status per problem
    assert lp.solve_batch([args, args]) == [0, 0]
    # the argv twin accepts the reference's flag string
    from qtos_amd import main as cli
    out2 = tmp_path / "traj.csv"
    assert cli.main(flags.cmd_args(args).split() + ["--out", str(out2)]) == 0
    assert np.array_equal(csvio.read_csv(str(out2)), rows)
    lp.close()


def test_feasibility_map_batched():
    """SURVEY.md 8f row 2 on the GPU: exp_3's 48 probe patches as ONE batch (the reference runs 48
    docker processes, 32 at a time) -> exit codes -> boolean map that A* can consume."""
    from qtos_amd import feasibility, heightfield
    from qtos_amd.global_planner import PathSolver
    from qtos_amd.planner import LocalPlanner
    tiles = [heightfield.read_tile(os.path.join(GOLDEN, "heightfields", t + ".txt"))
             for t in ("feasibility_test", "feasibility_test_1", "plane")]
    m = heightfield.build_map(tiles, 1)
    lp = LocalPlanner(max_batch=64)
    lp.set_heightfield(heightfield.towr_map(m), heightfield.cell_size(m))
    bm, patches, statuses = feasibility.feasibility_map(lp, m, multi_map_shift=3)
    assert len(patches) == 48 and len(statuses) == 48 and bm.shape == m.shape
    assert set(statuses) <= {0, 1, 2}
    # every failed patch blocks its start and goal cells, every cell blocked belongs to some failure
    for (_, _, s, g), rc in zip(patches, statuses):
        if rc != 0:
            assert bm[s] == 1 and bm[g] == 1
    if all(rc == 0 for rc in statuses):
        assert bm.sum() == 0
    ps = PathSolver(m, [0, 0, 0.24], [4.5, 0, 0.24], 1.0, 0.1, bool_map=bm)
    assert ps.path is None or ps.path[0] == ps.start_idx
    lp.close()


def test_replan_loop_stitches_two_plans(gv1):
    """SURVEY.md 8f row 1 end to end: plan, pick the hand-over row like Combiner._state, re-plan from
    it, splice like Combiner.combine."""
    from qtos_amd.planner import LocalPlanner
    from qtos_amd.stitcher import Stitcher
    inp = gv1["inputs"]
    args = {"-g": [0.52, 0.0, 0.24], "-s": inp["s"], "-s_ang": [0, 0, 0], "-e1": inp["ee"][0],
            "-e2": inp["ee"][1], "-e3": inp["ee"][2], "-e4": inp["ee"][3], "-resolution": 0.01}
    lp = LocalPlanner(max_batch=4)
    assert lp.solve(args, out_csv=None) == 0
    old = lp.last["rows"][0]
    st = Stitcher(lookahead=3750, height_set=(0.0,))
    st.cutoff_idx = 2500
    state = st.state(np.round(old, 6), last_timestep=0.006)
    assert st.legs_in_contact(state)
    args2 = st.plan_args(args, state, runtime=0.006, goal=[0.91, 0.0, 0.24])
    assert lp.solve(args2, out_csv=None) == 0
    new = lp.last["rows"][0]
    assert abs(new[0, 0] - args2["-t"]) < 1e-12
    comb = st.combine(old, new)
    t = comb[:, 0]
    assert np.all(np.diff(t) > 0) and abs(np.diff(t).max() - 0.001) < 1e-9      # seamless time base
    j = np.nonzero(t >= args2["-t"])[0][0]
    assert np.abs(comb[j + 1, 1:19] - comb[j, 1:19]).max() < 2e-3               # positions continuous
    lp.close()


def test_replan_loop_state_machine_follows_the_reference_update_thread():
    """scripts/main.py:_run/_update as qtos_amd.replan.ReplanLoop on exp_1 (flat, goal x = 2.5): plan, wait for f_steps
    rows, stitch, plan ... with Global_Planner.pop() feeding the goals; the stitched plan is one seamless 1 kHz
    trajectory.  The time-shifted warm start (qtos_shift_warm) is an option of the loop: same statuses, and its
    iteration count is recorded next to the cold start's (replan.py: it does not pay on this NLP)."""
    from qtos_amd.global_planner import GlobalPlanner
    from qtos_amd.planner import LocalPlanner
    from qtos_amd.replan import ReplanLoop
    gp = GlobalPlanner(np.zeros((20, 40)), [0, 0, 0.24], [2.5, 0, 0.24], step_size=1.0, resolution=0.1, lookahead=3750)
    lp = LocalPlanner(max_batch=2)
    loop = ReplanLoop(lp, gp, {"-resolution": 0.1}, lookahead=3750, f_steps=2500)
    plan = loop.run(max_plans=4)
    kinds = [e[0] for e in loop.events]
    assert kinds[:7] == ["plan", "plan", "stitch", "plan", "stitch", "plan", "stitch"][:len(kinds[:7])] and kinds.count("plan") == 4
    assert all(s == 0 for s in loop.statuses)
    # goals come from the global planner (the logged -g of the reference's first two solves: logs/towr_log.out:8,140)
    assert abs(loop.events[0][1]) < 1e-12
    t = plan[:, 0]
    assert np.all(np.diff(t) > 0) and abs(np.diff(t).max() - 0.001) < 1e-9            # seamless time base
    assert np.abs(np.diff(plan[:, 1:19], axis=0)).max() < 5e-3                           # positions continuous over the splices
    # second plan started lookahead rows ahead of the clock it was asked at, on an all-feet-down row
    assert loop.st.legs_in_contact({k: v for k, v in zip(("FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT"), np.round(loop.new[0, 7:19], 6).reshape(4, 3).tolist())})
    cold_iters = int(lp.last["iters"][0])
    # the same loop with the time-shifted previous plan as starting point
    gp2 = GlobalPlanner(np.zeros((20, 40)), [0, 0, 0.24], [2.5, 0, 0.24], step_size=1.0, resolution=0.1, lookahead=3750)
    loop2 = ReplanLoop(lp, gp2, {"-resolution": 0.1}, lookahead=3750, f_steps=2500, shifted_warm_start=True)
    loop2.run(max_plans=4)
    assert [e[0] for e in loop2.events] == kinds and all(s == 0 for s in loop2.statuses)
    print("last replan: %d iterations cold, %d from the shifted plan" % (cold_iters, int(lp.last["iters"][0])))
    lp.close()


@pytest.mark.parametrize("order", ["0", "auto"])
@pytest.mark.parametrize("preset", ["knots200", "receding_windows"])
def test_shifted_windows_match_oracle_over_five_replans(preset, order):
    """BASELINE configs[4] loop (qtos_amd.replan.ShiftedWindows, bench.py --workload mpc_random): six consecutive
    replans of four windows on randomized heightfields; every replan starts from an all-feet-down hand-over row of
    the previous plan -- cold for the first four replans (the loop's default), from the time-shifted previous plan
    (qtos_shift_warm) for the last two.  The oracle solves the same problems from the same starting points: same
    statuses, iteration counts and nodes.  Both configurations a user can get: the plain 200-knot transcription and
    PlannerConfig.receding_windows() -- no chord steps, plain barrier update: what bench.py times for configs[4].

    Tolerances (round 6; the round-5 gate of 3e-4 on the shifted replans is gone).  The product solves the Newton step with the
    swing rows and the base's continuity rows ELIMINATED (reduce_swing / reduce_base: no multipliers for them); the oracle
    keeps a multiplier for every row of the reference's NLP, regularised by -eps_dual = -1e-8, so its rows hold to eps x
    multiplier: 1e-8 x O(1) on a cold start, 1e-8 x O(1e3) at a shifted start whose violation is 40 - 250 -- the 1e-5 .. 1e-4
    that round 5 took for amplified rounding is there after the FIRST step (scratch/r6_shift_gap.py).  With the oracle's eps
    on exactly those rows at 1e-13 (oracle_options(match_eliminated=True)) the two agree to 1e-9 .. 3e-9 over 5 - 17
    iterations: gate 1e-7 for every replan.  The plain oracle -- the reference's formulation as it stands -- stays in the
    test at the bound that difference explains: 1e-6 cold, 3e-4 shifted (3 x eps x the largest multiplier seen).

    order: "0" = the elimination order of rounds 1 - 5 (QTOS_ORDER=0), "auto" = the planner's choice, for this transcription round
    6's order with the late force nodes (96 slots instead of 112, -10 % per KKT launch): the same gates.

    Round 6 also fixed the shifted start itself: k_shift_warm read the time KEYS of the elimination order as node times, and
    reduce_swing (round 5) had moved a foothold's key to the end of the swing behind its stance -- the warm start's footholds were
    the NEXT footholds' positions.  With the node times proper the shifted replans take 4 - 8 iterations (5 - 17 before)."""
    import torch
    from oracle.oracle import Oracle, oracle_dict, oracle_options
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    from qtos_amd.replan import ShiftedWindows
    cfg = getattr(PlannerConfig, preset)()
    assert (cfg.chord_tol, cfg.mu_superlinear) == ((0.0, False) if preset == "receding_windows" else (4e-3, True))
    maps, cell = workloads.random_terrains()
    old_order = os.environ.get("QTOS_ORDER")
    if order == "0":
        os.environ["QTOS_ORDER"] = "0"
    else:
        os.environ.pop("QTOS_ORDER", None)
    try:
        P = Planner(cfg, max_batch=4)
    finally:
        if old_order is None:
            os.environ.pop("QTOS_ORDER", None)
        else:
            os.environ["QTOS_ORDER"] = old_order
    assert P.dims.front == (112 if order == "0" else 96)
    P.set_heightfields(maps, cell)
    start, goal, map_id = workloads.mpc_goals(4, seed=5, terrains=(maps, cell))
    W = ShiftedWindows(P, start, goal - start[:, 0:3], map_id, advance=2.5)
    oracles = [Oracle(oracle_dict(cfg), height=maps[m], hcell=cell) for m in map_id]
    from oracle.projection import project_nodes
    var_free = P.structure()[1]
    worst = {"matched": 0.0, "plain_cold": 0.0, "plain_shifted": 0.0}
    for k in range(6):
        W.warm_mode = "shifted" if k >= 4 else "none"
        nodes, status = W.replan()
        torch.cuda.synchronize()
        st, gl = W.start.cpu().numpy(), W.goal.cpu().numpy()
        # (a solve that is given nodes starts from their projection onto the reduced base's spline space: the oracle gets the same)
        # ... through the numpy / scipy restatement of the projection (oracle/projection.py), not the product's own
        warm = project_nodes(W.warm.cpu().numpy(), oracles[0].L, var_free) if k >= 4 else [None] * 4
        it = W.iters.cpu().numpy()
        if k > 0:   # hand-over rows: all four feet carry force, a few hundred rows after `advance`
            assert (W.offset.cpu().numpy() >= 2.5).all() and (W.offset.cpu().numpy() <= 2.9).all()
        for b in range(4):
            O = oracles[b]
            q = O.problem(st[b, 0:3], st[b, 3:6], st[b, 6:18].reshape(4, 3), gl[b])
            xg = nodes[b].cpu().numpy()
            xo, info = O.solve(q, x0=warm[b], opts=oracle_options(cfg, O, match_eliminated=True))
            assert info.status == int(status[b]) == 0 and info.iters == int(it[b])
            worst["matched"] = max(worst["matched"], float(np.abs(xg - xo).max()))
            if k < 4:
                worst["matched_cold"] = max(worst.get("matched_cold", 0.0), float(np.abs(xg - xo).max()))
            xo, info = O.solve(q, x0=warm[b], opts=oracle_options(cfg, O))
            assert info.status == int(status[b]) == 0 and info.iters == int(it[b])
            worst["plain_shifted" if k >= 4 else "plain_cold"] = max(worst["plain_shifted" if k >= 4 else "plain_cold"], float(np.abs(xg - xo).max()))
    print("shifted windows [%s, order %s]: worst |gpu - oracle| %s" % (preset, order, worst))
    assert worst["matched"] < 1e-7 and worst["plain_cold"] < 1e-6 and worst["plain_shifted"] < 3e-4, worst
    P.close()


def test_mixed_terrain_batch_with_map_ids(cfg):
    """BASELINE configs[3] (single-GPU shard of it): exp_1 / exp_3 / exp_5 patches in one batch, a
    heightfield index per problem; every converged plan is feasible on ITS OWN terrain."""
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    maps, cell = workloads.mixed_terrains()
    start, goal, map_id = workloads.mixed_goals(96, seed=2, terrains=(maps, cell))
    P = Planner(cfg, max_batch=96)
    P.set_heightfields(maps, cell)
    nodes, status, iters, viol = P.plan(start, goal, map_id=map_id)
    assert (status == 0).mean() >= 0.85
    for k in range(3):
        assert (status[map_id == k] == 0).mean() >= 0.75
    oracles = [Oracle(oracle_dict(cfg), height=maps[k], hcell=cell) for k in range(3)]
    checked = 0
    for b in np.nonzero(status == 0)[0][:24]:
        assert oracles[map_id[b]].max_violation(nodes[b]) <= 1e-4 + 1e-9
        checked += 1
    assert checked >= 12
    # same problems solved one map at a time give the same plans (map index only selects terrain)
    sel = np.nonzero(map_id == 2)[0][:8]
    P.set_heightfields(maps[2], cell)
    n2, s2, _, _ = P.plan(start[sel], goal[sel])
    assert np.array_equal(s2, status[sel]) and np.array_equal(n2, nodes[sel])
    P.close()


def _batch_vs_oracle(cfg, start, goal, sel, maps=None, cell=None, map_id=None, status=None, iters=None, nodes=None, tol=1e-6, eps_dual=None):
    """The problems `sel` of a batch solved by the oracle (OpenMP over the problems, one oracle per heightfield): returns
    the number of problems whose status and iteration count equal the GPU's and whose nodes agree to 1e-6, and the
    worst node difference among them."""
    import os
    from oracle.oracle import Oracle, oracle_dict, oracle_options
    same, worst = 0, 0.0
    groups = {}
    for b in sel:
        groups.setdefault(0 if map_id is None else int(map_id[b]), []).append(int(b))
    for m, idx in groups.items():
        h = None if maps is None else (maps if np.ndim(maps) == 2 else maps[m])
        O = Oracle(oracle_dict(cfg), height=h, hcell=cell if cell else 0.1)
        qs = [O.problem(start[b][0:3], start[b][3:6], start[b][6:18].reshape(4, 3), goal[b], start[b][18:21], start[b][21:24]) for b in idx]
        opts = oracle_options(cfg, O)
        if eps_dual is not None:
            opts.eps_dual = eps_dual          # (the oracle's regularisation of the multipliers only: the product keeps its own)
        xo, infos = O.solve_batch(qs, n_threads=os.cpu_count() or 1, opts=opts)
        for j, b in enumerate(idx):
            if infos[j].status == int(status[b]) and infos[j].iters == int(iters[b]):
                e = float(np.abs(nodes[b] - xo[j]).max())
                if e < tol:
                    same += 1
                    worst = max(worst, e)
    return same, worst


def test_jammed_problem_stops_like_a_stalled_one():
    """One window in 250 on the randomized heightfields is left no room by the fraction-to-the-boundary rule: step lengths
    0.02, 0.00, 0.00 ... at a violation of 2.5.  It used to sit there until a division overflowed in its ninth iteration
    (status 2) while its batch waited; two steps in a row shorter than PlannerConfig.stall_alpha now stop it like a stalled
    problem -- status 1, best iterate, iteration 5 -- in the product and in the oracle alike."""
    import dataclasses
    from oracle.oracle import Oracle, oracle_dict, oracle_options
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.knots200(chord_tol=0.0)
    assert cfg.stall_alpha == 1e-2
    maps, cell = workloads.random_terrains()
    start, goal, mid = workloads.mpc_goals(256, seed=5, terrains=(maps, cell))
    P = Planner(cfg, max_batch=256)
    P.set_heightfields(maps, cell)
    nodes, status, iters, viol = P.plan(start, goal, map_id=mid)
    jam = np.nonzero(status != 0)[0]
    assert len(jam) >= 1 and (status[jam] == 1).all() and (iters[jam] <= 6).all()   # (no numerical failure, no ninth iteration)
    b = int(jam[0])
    tr = np.asarray(P.trace(b))[:iters[b] + 1]
    assert (tr[-2:, 2] < cfg.stall_alpha).all() and tr[-3, 2] >= cfg.stall_alpha      # the two short steps that stopped it
    assert viol[b] == tr[:, 0].min()                                                   # the best iterate came back
    P.close()
    P0 = Planner(dataclasses.replace(cfg, stall_alpha=0.0), max_batch=256)
    P0.set_heightfields(maps, cell)
    _, status0, iters0, _ = P0.plan(start, goal, map_id=mid)
    P0.close()
    assert status0[b] != 0 and iters0[b] > iters[b] + 2                                # without the rule: iterations later
    ok = status == 0
    assert np.array_equal(status0[ok], status[ok]) and np.array_equal(iters0[ok], iters[ok])
    O = Oracle(oracle_dict(cfg), height=maps[mid[b]], hcell=cell)
    s = start[b]
    xo, info = O.solve(O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), goal[b], (0, 0, 0), (0, 0, 0), 0.0), opts=oracle_options(cfg, O))
    assert (info.status, info.iters) == (1, int(iters[b]))
    assert np.abs(nodes[b] - xo).max() < 1e-5


def test_second_chord_step_finishes_the_trot_batch():
    """Half of a trot batch leaves its chord step at 1.1e-4, ten per cent above the tolerance: a second chord step with the same
    factorisation (PlannerConfig.chord_max = 2: allowed behind a full chord step that cut the violation to a third) finishes
    them -- three factorisations and two chord solves per batch; with one chord step per factorisation the whole batch pays a
    fourth factorisation for those problems.  The problems the first chord step finishes get bit-identical plans."""
    import dataclasses
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    # (under the plain mu <- 0.2 mu of rounds 1 - 4: with Ipopt's update of the barrier parameter, the default since round 5,
    #  every problem of this batch converges behind its first chord step -- test_trot_gait_batch_matches_oracle)
    cfg = PlannerConfig.knots100(gait="trot", mu_superlinear=False)
    assert cfg.chord_max == 2
    start, goal = workloads.flat_goals(128, seed=0)
    out = {}
    for cm in (2, 1):
        P = Planner(dataclasses.replace(cfg, chord_max=cm), max_batch=128)
        nodes, status, iters, viol = P.plan(start, goal)
        t = P.timing()
        P.close()
        assert (status == 0).all() and viol.max() <= cfg.tol
        out[cm] = (nodes, iters, t["kkt_launches"], t["chord_launches"])
    assert out[2][2:] == (3, 2) and out[1][2:] == (4, 1)
    assert np.array_equal(out[2][1], out[1][1]) and set(np.unique(out[2][1])) == {4, 5}
    same = out[2][1] == 4           # (finished by the first chord step: the second never ran)
    assert same.sum() >= 32 and np.array_equal(out[2][0][same], out[1][0][same])
    assert np.abs(out[2][0] - out[1][0]).max() < 5e-2   # (two points of the feasible set, both within the tolerance: forces of ~7 N differ by ~1e-2)


@pytest.mark.parametrize("reduce_base", [False, True])
def test_trot_gait_batch_matches_oracle(reduce_base):
    """The gait BASELINE.json's metric names (diagonal-pair trot, config.TROT_UNNORMALISED; the reference's committed
    plans are the walk), at the benchmark's transcription and batch size: all 256 seeded flat goals converge, and the
    first 32 take the oracle's iterations to the oracle's nodes to 1e-6, with every row of the NLP in the KKT system and with
    the reduced base (the default).  Rounds 3 - 5 held the reduced base to 5e-6 and blamed the oracle's regularisation, then the
    rounding sensitivity of a cost-free NLP; round 6 found the cause -- the elimination order of those rounds loses digits of the
    trot's KKT solve (model.hpp order_rule; under rule 2 the gap is the full system's 5e-8) -- and took the gate back to 1e-6."""
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.knots100(gait="trot", reduce_base=reduce_base)
    B = 256
    P = Planner(cfg, max_batch=B)
    assert (P.dims.n_vars, P.dims.n_stages) == (1880, 113 if reduce_base else 186)   # (reduce_swing, the default: 127 / 200 without)
    start, goal = workloads.flat_goals(B, seed=0)
    nodes, status, iters, viol = P.plan(start, goal)
    P.close()
    assert (status == 0).all() and viol.max() <= cfg.tol
    # (round 5, Ipopt's update of the barrier parameter: the whole batch in four iterations -- it was 4: 58 %, 5: 42 %)
    assert iters.max() <= (4 if reduce_base else 6)
    same, worst = _batch_vs_oracle(cfg, start, goal, range(32), status=status, iters=iters, nodes=nodes, tol=1e-6)
    assert same == 32, (same, worst)


@pytest.mark.gpu
@pytest.mark.parametrize("gait", ["walk", "trot"])
def test_unmirrored_pair_full_swings_against_the_plain_oracle(gait):
    """The round-4 pair stays pinned while the mirrored pair evolves (round-5 verdict): the product with reduce_swing OFF --
    every swing row an equality row with its multiplier, towr's straight-line guess as the starting point as it is -- against
    the oracle WITHOUT the mirror of round 5 (swing_start_on_rule = 0: nothing of the product's elimination is written into the
    checker), at the benchmark's transcription: same statuses, same iteration counts, nodes to 1e-6 (measured 2e-10 on both gaits)."""
    from oracle.oracle import Oracle, oracle_dict, oracle_options
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.knots100(gait=gait, reduce_swing=False)
    B = 16
    O = Oracle(oracle_dict(cfg))
    opts = oracle_options(cfg, O)
    assert not O.swing_start_on_rule and opts.swing_start_on_rule == 0 and opts.eps_dual_swing < 0     # the checker as round 4 left it
    P = Planner(cfg, max_batch=B)
    start, goal = workloads.flat_goals(B, seed=23)
    nodes, status, iters, viol = P.plan(start, goal)
    P.close()
    assert (status == 0).all()
    same, worst = _batch_vs_oracle(cfg, start, goal, range(B), status=status, iters=iters, nodes=nodes, tol=1e-6)
    assert same == B, (same, worst)


@pytest.mark.gpu
def test_trot_gap_to_the_oracle_was_the_elimination_order():
    """Where the 5e-6 of the trot in rounds 3 - 5 came from.  Round 3 blamed the oracle's eps_dual on the multipliers of the
    acceleration-continuity rows, round 5 the sensitivity of a cost-free NLP's iterates to rounding (two KKT kernels of the product
    -- k_kkt2 and k_kkt5: the same system, another slot assignment and order of every sum -- differed by 2.6e-6 on the trot).
    Round 6 measured the KKT solve itself over a spread of transcriptions and found the order of rounds 1 - 5 (rule 0) losing up
    to six digits on trots with a reduced base: the multipliers of the second dynamics knot and of the first junction's
    acceleration rows were eliminated behind one B-spline coefficient per dimension.  With the coefficients one polynomial
    earlier (rule 2, what the planner keeps now) the trot's plans equal the oracle's to 5e-8 -- the gap of the FULL system, i.e.
    the oracle's regularisation and nothing else -- and two kernels of the product agree to 4e-12.  Pinned here: the new order
    at 5e-7 / 1e-9, and the old order (QTOS_ORDER=0) showing the old gap, so that the cause stays on record."""
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    res = {}
    B = 32
    start, goal = workloads.flat_goals(B, seed=0)
    for gait in ("trot", "walk"):
        cfg = PlannerConfig.knots100(gait=gait)
        for order in (None, "0"):
            plans = {}
            for kkt in ("2", "6"):
                os.environ["QTOS_KKT"] = kkt
                if order is not None:
                    os.environ["QTOS_ORDER"] = order
                try:
                    P = Planner(cfg, max_batch=B)
                finally:
                    del os.environ["QTOS_KKT"]
                    os.environ.pop("QTOS_ORDER", None)
                assert P.dims.order_rule == (0 if order == "0" else 2 if gait == "trot" else 1)
                plans[kkt] = P.plan(start, goal)
                P.close()
            (n2, s2, i2, v2), (n6, s6, i6, v6) = plans["2"], plans["6"]
            assert (s2 == 0).all() and (s6 == 0).all() and np.array_equal(i2, i6) and max(v2.max(), v6.max()) <= cfg.tol
            same, worst = _batch_vs_oracle(cfg, start, goal, range(16), status=s2, iters=i2, nodes=n2, tol=1e-4)
            assert same == 16, (gait, order, same)
            res[(gait, order)] = (worst, float(np.abs(n2 - n6).max()))
    (gt, kt), (gt0, kt0) = res[("trot", None)], res[("trot", "0")]
    (gw, kw), (gw0, kw0) = res[("walk", None)], res[("walk", "0")]
    assert gt < 5e-7 and kt < 1e-9 and gw < 1e-6 and kw < 1e-9, res          # the orders the planner keeps (measured 5.1e-8 / 3.6e-12, 1.8e-7 / 3.5e-12)
    assert 3e-7 < gt0 < 5e-6 and kt0 > 100 * kt, res                          # the trot under the order of rounds 1 - 5 (measured 1.1e-6, kernels 2.6e-6 apart)
    assert gw0 < 1e-6 and kw0 < 1e-8, res                                     # the walk never had the problem


@pytest.mark.gpu
def test_exp5_batch_matches_oracle_at_baseline_size():
    """BASELINE configs[2] at its own size: batch 256 of exp_5 step-climb goals on the 100-knot transcription; all
    converge, and the first 32 take the oracle's iterations to the oracle's nodes (1e-6)."""
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.knots100()
    hxy, cell = workloads.exp5_terrain()
    start, goal = workloads.step_goals(256, seed=1, terrain=(hxy, cell))
    P = Planner(cfg, max_batch=256)
    P.set_heightfields(hxy, cell)
    nodes, status, iters, viol = P.plan(start, goal)
    P.close()
    assert (status == 0).all() and viol.max() <= cfg.tol
    same, worst = _batch_vs_oracle(cfg, start, goal, range(32), hxy, cell, None, status, iters, nodes)
    assert same == 32, (same, worst)


@pytest.mark.gpu
def test_mixed_batch_matches_oracle_at_baseline_size():
    """BASELINE configs[3], one GPU's shard at the benchmark's batch size: 256 problems over the exp_1 / exp_3 / exp_5
    patches (a heightfield index per problem), 100-knot transcription.  At least 32 problems PER TERRAIN are compared
    with the oracle: same status, same iteration count, nodes to 1e-6."""
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.knots100()
    maps, cell = workloads.mixed_terrains()
    start, goal, map_id = workloads.mixed_goals(256, seed=2, terrains=(maps, cell))
    P = Planner(cfg, max_batch=256)
    P.set_heightfields(maps, cell)
    nodes, status, iters, viol = P.plan(start, goal, map_id=map_id)
    P.close()
    assert (status == 0).mean() >= 0.99 and viol[status == 0].max() <= cfg.tol
    for m in np.unique(map_id):
        sel = np.nonzero(map_id == m)[0][:32]
        assert len(sel) == 32
        same, worst = _batch_vs_oracle(cfg, start, goal, sel, maps, cell, map_id, status, iters, nodes)
        assert same == 32, (int(m), same, worst)


@pytest.mark.gpu
def test_shifted_windows_walk_for_twenty_replans_at_baseline_size():
    """BASELINE configs[4] as a property test at the benchmark's size: 256 receding windows on the randomized
    heightfields, 20 consecutive replans with the reference's hand-over rule (QTOS/combiner.py:245-296: the first
    all-feet-down row at least 2.5 s into the newest plan).  >= 99 % of all replans converge, every plan is finite,
    every hand-over offset lies in [2.5, 2.9] s, and the windows really move."""
    import torch
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    from qtos_amd.replan import ShiftedWindows
    cfg = PlannerConfig.knots200(chord_tol=0.0)
    B, K = 256, 20
    maps, cell = workloads.random_terrains()
    P = Planner(cfg, max_batch=B)
    P.set_heightfields(maps, cell)
    start, goal, map_id = workloads.mpc_goals(B, seed=5, terrains=(maps, cell))
    W = ShiftedWindows(P, start, goal - start[:, 0:3], map_id, advance=2.5, x_range=(0.0, 2.2))
    x0 = W.start[:, 0].clone()
    conv, moved = 0, torch.zeros(B, dtype=torch.float64, device=W.dev)
    for k in range(K):
        xprev = W.start[:, 0].clone()
        nodes, status = W.replan()
        torch.cuda.synchronize()
        conv += int((status == 0).sum())
        assert bool(torch.isfinite(nodes).all())
        if k > 0:
            off = W.offset.cpu().numpy()
            assert (off >= 2.5).all() and (off <= 2.9).all()
            moved += (W.start[:, 0] - xprev).abs()
    P.close()
    assert conv >= 0.99 * B * K, conv
    assert float(moved.min()) > 0.5        # every window walked (there and back on its 2.2 m map)


def test_edge_cases_infeasible_nan_and_chunking(planner, cfg):
    """Error behaviour at the boundary: an unreachable goal and a NaN start come back as non-zero
    exit statuses (the reference's solver exits non-zero; callers treat that as infeasible,
    QTOS/generateHeightField.py:387-404) without disturbing the other problems of the batch; batches
    larger than the planner's capacity are chunked by LocalPlanner."""
    from qtos_amd import workloads
    from qtos_amd.planner import LocalPlanner
    start, goal = workloads.flat_goals(8, seed=9)
    ref_nodes, ref_status, _, _ = planner.plan(start, goal)
    bad_goal = goal.copy()
    bad_goal[2, 0] += 4.0                      # 4 m in 5 s with 0.07 m leg reach: infeasible
    bad_start = start.copy()
    bad_start[5, 2] = np.nan
    nodes, status, iters, viol = planner.plan(bad_start, bad_goal)
    assert status[2] != 0 and status[5] != 0
    ok = [b for b in range(8) if b not in (2, 5)]
    assert (status[ok] == 0).all() and np.array_equal(nodes[ok], ref_nodes[ok])
    # single problem == the same problem inside a batch
    n1, s1, _, _ = planner.plan(start[3:4], goal[3:4])
    assert s1[0] == 0 and np.array_equal(n1[0], ref_nodes[3])
    # LocalPlanner chunks a 10-problem list over a capacity-4 planner
    lp = LocalPlanner(cfg=cfg, max_batch=4)
    args = [{"-s": s[0:3].tolist(), "-s_ang": [0, 0, 0], "-e1": s[6:9].tolist(), "-e2": s[9:12].tolist(),
             "-e3": s[12:15].tolist(), "-e4": s[15:18].tolist(), "-g": g.tolist()} for s, g in zip(start, goal)]
    st = lp.solve_batch(args + args[:2], sample=False)
    assert st == [0] * 10 and np.array_equal(lp.last["nodes"][:8], ref_nodes)
    lp.close()


@pytest.mark.gpu
def test_stall_detection_returns_best_iterate(cfg):
    """A problem that stops lowering its violation (a foot cycling across a ledge edge of the
    piecewise-constant exp_5 terrain) stops `stall_iters` iterations after its best iterate with
    status 1 and returns that iterate; converged problems are untouched; with the rule switched
    off the same problems run on, most of them to the iteration limit.  The oracle applies the same rule.
    (Footholds left free for the whole solve, `foothold_hold_from` = 0; chord steps off: the rule under test is
    about the cycling iterates of the plain Newton sequence.)"""
    import dataclasses
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    hxy, cell = workloads.exp5_terrain()
    start, goal = workloads.step_goals(256, seed=1, terrain=(hxy, cell))
    cfg = dataclasses.replace(cfg, foothold_hold_from=0, chord_tol=0.0)
    P = Planner(cfg, max_batch=256)
    P.set_heightfields(hxy, cell)
    nodes, status, iters, viol = P.plan(start, goal)
    stuck = np.nonzero(status == 1)[0]
    assert 0 < len(stuck) <= 16 and (status != 2).all()
    traces = {int(b): P.trace(int(b)) for b in stuck}
    P.close()
    for b in stuck:
        tr = traces[int(b)]
        best = int(np.argmin(tr[:, 0]))
        assert iters[b] < cfg.max_iter and iters[b] == best + cfg.stall_iters
        assert abs(viol[b] - tr[best, 0]) <= 1e-12
    O = Oracle(oracle_dict(cfg), height=hxy, hcell=cell)
    assert all(O.max_violation(nodes[b]) <= viol[b] + 1e-9 for b in stuck[:4])
    b = int(stuck[0])
    s, g = start[b], goal[b]
    oo = O.default_options()
    oo.hold_from = 0
    oo.chord_tol = 0.0
    xo, info = O.solve(O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g, (0, 0, 0), (0, 0, 0), 0.0), opts=oo)
    assert info.status == 1 and info.iters < cfg.max_iter
    P0 = Planner(dataclasses.replace(cfg, stall_iters=0), max_batch=256)
    P0.set_heightfields(hxy, cell)
    nodes0, status0, iters0, _ = P0.plan(start, goal)
    P0.close()
    ok = status == 0
    assert (status0[ok] == 0).all() and np.array_equal(nodes0[ok], nodes[ok]) and np.array_equal(iters0[ok], iters[ok])
    # (the stopped problems go on without the rule: to the iteration limit, or -- the barrier parameter stays at its floor,
    #  round 5 -- to a late convergence the batch would have waited for)
    assert (iters0[stuck] > iters[stuck]).all() and (iters0[stuck] == cfg.max_iter).any()


@pytest.mark.gpu
def test_factor_panels_match_block_elimination(planner_full, oracle, gv1, cfg):
    check_factor_panels(planner_full, oracle, gv1, cfg)


def check_factor_panels(planner_full, oracle, gv1, cfg):
    """The factor panels k_kkt leaves in HBM (per stage w = L^-T D^-1 y_F and V = Y D^-1 L^-1) against
    an independent numpy block elimination of the same KKT matrix in the planner_full's elimination order
    (16 pivots per stage, unpivoted LDL^T of the pivot block, explicit L^-1), stage by stage: this
    pins the chain itself, not only the solution it produces.  Tolerance 1e-6 relative to the stage's
    largest entry (the factors reach 1e8 where a multiplier is eliminated)."""
    rng = np.random.default_rng(5)
    inp = gv1["inputs"]
    x = gv1["x"][None] + 0.01 * rng.standard_normal((1, planner_full.n))
    lo, hi = oracle.var_bounds(oracle_problem(oracle, inp))
    fx = lo == hi
    x[:, fx] = lo[fx]
    start, goal = start_vector(inp)[None], np.array(inp["g"])[None]
    rk, _, order = planner_full.structure()
    I = rk == 2
    sig = np.zeros((1, planner_full.m)); w = np.zeros((1, planner_full.m))
    sig[:, I] = 10.0 ** rng.uniform(-3, 3, (1, I.sum())); w[:, I] = rng.standard_normal((1, I.sum()))
    dx = planner_full.debug_newton(start, goal, x, sig, w)
    pan, ps = planner_full.factor(0)
    free, E, Ii = np.nonzero(~fx)[0], np.nonzero(rk == 1)[0], np.nonzero(I)[0]
    nf, nE = len(free), len(E)
    Jo, go = oracle.jacobian(x[0]), oracle.constraints(x[0])
    JE, JI = Jo[np.ix_(E, free)], Jo[np.ix_(Ii, free)]
    K = np.zeros((nf + nE, nf + nE))
    K[:nf, :nf] = cfg.delta_x * np.eye(nf) + JI.T @ (sig[0, Ii][:, None] * JI)
    K[nf:, :nf] = JE; K[:nf, nf:] = JE.T; K[nf:, nf:] = -cfg.eps_dual * np.eye(nE)
    rhs = np.concatenate([-JI.T @ w[0, Ii], -go[E]])
    pos_of_var = {v: i for i, v in enumerate(free)}
    pos_of_row = {r: nf + i for i, r in enumerate(E)}
    # (by position; -1 = the dummy pivots that fill a short stage: unit pivot, no entries, zero right-hand side)
    Np = len(order); NS = Np // 16; N = Np
    real = np.nonzero(order >= 0)[0]
    perm = np.array([pos_of_var[u] if u < planner_full.n else pos_of_row[u - planner_full.n] for u in order[real]])
    S = np.eye(Np); S[np.ix_(real, real)] = K[np.ix_(perm, perm)]
    y = np.zeros(Np); y[real] = rhs[perm]
    slot_of_pos = ps.ravel()
    checked = 0
    for k in range(NS):
        p, r = slice(16 * k, 16 * k + 16), slice(16 * k + 16, Np)
        L = np.eye(16); d = np.zeros(16); Aw = S[p, p].copy()
        for i in range(16):
            d[i] = Aw[i, i]
            L[i + 1:, i] = Aw[i + 1:, i] / d[i]
            Aw[i + 1:, i + 1:] -= np.outer(L[i + 1:, i], Aw[i, i + 1:])
        Li = np.linalg.inv(L)
        Y = S[r, p] @ Li.T
        yF = Li @ y[p]
        S[r, r] -= (Y / d) @ Y.T
        y[r] -= (Y / d) @ yF
        V, wk = (Y / d) @ Li, Li.T @ (yF / d)
        if k % 9 == 0 or k == NS - 1:      # a dozen stages across the chain
            nxt = {}
            for pp in range(16 * (k + 1), N):
                nxt.setdefault(int(slot_of_pos[pp]), pp)          # next occupant of every slot
            rows = np.array(sorted(nxt.values()), dtype=int)
            scale = max(1.0, np.abs(V).max(initial=0.0), np.abs(wk).max())
            if len(rows):
                Vg = pan[k, 1 + slot_of_pos[rows]]
                assert np.abs(Vg - V[rows - 16 * (k + 1)]).max() <= 1e-6 * scale
            assert np.abs(pan[k, 0] - wk).max() <= 1e-6 * scale
            checked += 1
    assert checked >= 10
    ref = np.linalg.solve(K, rhs)[:nf]
    assert np.abs(dx[0, free] - ref).max() <= 2e-5 * np.abs(ref).max()


@pytest.mark.gpu
def test_knots200_receding_window_on_random_heightfields():
    """BASELINE configs[4]: 200 dynamics knots (10 s, two walk cycles), randomized heightfields, replans
    at 50 Hz.  Cold solves and every warm-started replan are checked against the oracle run on the same
    inputs (same start row, same warm nodes): nodes to 1e-5 when both took the same number of
    iterations, CoM / feet of the sampled trajectory to the stated 1e-3 m otherwise."""
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.knots200(honor_start_velocity=True)   # a replan continues the motion it starts in
    B, NCHK = 64, 3
    P = Planner(cfg, max_batch=B)
    assert (P.dims.n_dyn_times, P.dims.front, P.dims.n_stages) == (202, 96, 192)   # (round 6: 96 slots by the order with the late force nodes; 112 by the order of rounds 1 - 5)
    maps, cell = workloads.random_terrains()
    P.set_heightfields(maps, cell)
    start, goal, mid = workloads.mpc_goals(B, terrains=(maps, cell))
    oracles = {}

    def check(chk, nodes, status, iters, start, warm):
        """The oracle on the same inputs.  The ledges make the problem piecewise smooth: a solve whose
        iterates never sit near a cell edge follows the oracle step for step (same iteration count,
        nodes to 1e-5, trajectories to the stated 1e-3 m); one that does may branch differently (both
        feasible), so those only have to be feasible."""
        exact = 0
        for b in chk:
            if b not in oracles:
                oracles[b] = Oracle(oracle_dict(cfg), height=maps[mid[b]], hcell=cell)
            O, s = oracles[b], start[b]
            assert (O.n, O.m) == (P.n, P.m)
            q = O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), goal[b], s[18:21], s[21:24], 0.0)
            xo, info = O.solve(q, x0=None if warm is None else warm[b])
            if status[b] == 0:
                assert O.max_violation(nodes[b]) <= cfg.tol + 1e-9
            if info.status == 0 and status[b] == 0 and info.iters == iters[b]:
                assert np.abs(nodes[b] - xo).max() < 1e-5
                ro = O.sample(xo, hz=100.0)
                rg = P.sample(nodes[b:b + 1], 0.0, hz=100.0)[0]
                assert np.abs(rg[:, 1:4] - ro[:, 1:4]).max() < 1e-3      # CoM
                assert np.abs(rg[:, 7:19] - ro[:, 7:19]).max() < 1e-3    # feet
                exact += 1
        return exact

    nodes, status, iters, viol = P.plan(start, goal, map_id=mid)
    assert (status == 0).mean() >= 0.9 and viol[status == 0].max() <= cfg.tol   # nearest-cell terrain: a few stall on a cell edge
    assert len({int(mid[b]) for b in range(B)}) == 8 and np.ptp(maps, axis=0).max() > 0.01   # the maps differ
    cold_iters = iters.copy()
    from oracle.oracle import Oracle as _O, oracle_dict as _od
    from oracle.projection import project_nodes
    proj_layout, var_free = _O(_od(cfg)).L, P.structure()[1]
    chk = [int(b) for b in np.nonzero((status == 0) & (iters <= 5))[0][:NCHK + 3]]
    assert check(chk, nodes, status, iters, start, None) >= NCHK
    # five replans of the receding window: the next start is the row 20 ms into the current plan, the
    # current nodes are the warm start
    for k in range(5):
        row = P.sample(nodes, 0.0, hz=50.0, n_rows=2)[:, 1]
        nstart = row[:, 1:25].copy()
        assert np.abs(nstart[:, 0:3] - start[:, 0:3]).max() < 0.02     # 20 ms later: barely moved
        warm = nodes
        nodes, status, iters, viol = P.plan(nstart, goal, map_id=mid, warm=warm)
        ok = status == 0
        assert ok.mean() >= 0.9
        assert iters[ok].mean() < cold_iters.mean() - 0.25             # the warm start pays (cold: 4-5 iterations, warm: 3-4)
        # (what the solve starts from: the nodes' projection onto the reduced base's spline space -- handed to the oracle from
        #  the numpy / scipy restatement, oracle/projection.py; test_projection.py holds the product's to it)
        assert check(chk, nodes, status, iters, nstart, project_nodes(warm, proj_layout, var_free)) >= NCHK - 1
        start = nstart
    # (the 20 ms look-ahead loop of round 1 drifts after ~28 replans -- replanning from one's own first 20 ms -- and is not
    #  the reference's loop: the reference's hand-over rule, 2.5 s ahead with all feet down, is exercised over 20 replans
    #  of 256 windows in test_shifted_windows_walk_for_twenty_replans_at_baseline_size)
    P.close()


@pytest.mark.gpu
@pytest.mark.parametrize("reduce_base", [False, True])
@pytest.mark.parametrize("kw,front,heavy", [
    (dict(duration=12.0), 160, True),                                  # default-TOWR mode on a 3-tile experiment: `-duration 4.0 * tiles`
                                                                       # (scripts/main.py:119-120): a 160-slot front, `k_kkt2<160>`
    (dict(duration=20.0), 208, True),                                  # the largest `-duration` the boundary takes (208 slots: thirteen
                                                                       # row tiles, 91 Schur tiles)
    (dict(duration=10.0, dt_base=0.05, dt_dynamic=0.05), 208, True),   # the one-cycle schedule stretched to 10 s at 0.05 s knots (what the
                                                                       # reference's `-duration 10` means for the 100-knot transcription)
    (dict(duration=8.0), 112, True),                                   # `-duration 8`: stance phases of 1.7 s own more
                                                                       # inequality blocks than one stage record holds
    (dict(duration=2.5), 96, False),                                   # `-duration 2.5` (scripts/main.py:119-120)
    (dict(duration=2.5, dt_dynamic=0.2, dt_base=0.2), 96, True),
    (dict(duration=1.5, dt_dynamic=0.25, dt_base=0.25, dt_range_of_motion=0.25), 96, False),
    (dict(duration=1.0, dt_dynamic=0.5, dt_base=0.5, dt_range_of_motion=0.5), 64, False),              # a 64-slot front: `k_kkt<64>`, four
                                                                       # of the eight waves without a panel tile
])
def test_other_horizons_match_oracle(kw, front, heavy, reduce_base):
    """Horizons other than 5 s (the reference's `-duration` flag rescales the gait schedule): other
    front sizes (other `k_kkt<F>` instantiations) and, for long stance phases, stages whose inequality
    blocks spill into continuation records.  GPU vs oracle, same iterates."""
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd import capi, workloads
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.reference_compat(reduce_base=reduce_base, **kw)
    d, _ = capi.analyze(cfg)
    # fronts of the full system with stage boundaries at multiples of 16 unknowns as listed; short stages (round 4,
    # Symbolic::shorten_stages) and the reduced base (half the base unknowns, no continuity multipliers) make them the same or
    # smaller -- 12 s: 144 / 128 instead of 160, 20 s: 208 / 160
    assert d.front <= front and (d.front >= front - 80)   # (round 6, reduced base: the order with the late force nodes where it is smaller: 10 s at 0.05 s knots 128, 20 s 144)
    P = capi.Planner(cfg, max_batch=8)
    O = Oracle(oracle_dict(cfg))
    assert (P.n, P.m) == (O.n, O.m)
    start, goal = workloads.flat_goals(8, seed=11)
    # same average speed up to 8 s; beyond, the one-cycle gait's stride limit caps the distance (the scaled goals are
    # infeasible for the oracle too)
    goal[:, 0] = start[:, 0] + (goal[:, 0] - start[:, 0]) * (cfg.duration / 5.0 if cfg.duration <= 8.0 else 1.0)
    nodes, status, iters, viol = P.plan(start, goal)
    assert (status == 0).all() and viol.max() <= cfg.tol
    from oracle.oracle import oracle_options
    # the oracle with its eps on the rows the product eliminates (swing rows; with the reduced base the continuity rows) at
    # 1e-13: the product has no multipliers for them.  1e-6 up to the 8 s horizon again (round 5 had loosened it to 1e-5 and
    # blamed a drift per iteration: it is eps x multiplier of those rows, scratch/r6_shift_gap.py: 4.6e-7 -> 1.8e-9); the 10 s and
    # 20 s horizons -- 12 to 14 iterations on 200 knots -- keep 1e-5 (5.4e-6 on the full-base 10 s case with the swing rows
    # matched: what is left there is the rounding sensitivity of a long cost-free solve, DESIGN.md section 4's table of floors)
    xo, infos = _oracle_solve(O, start[:4], goal[:4], opts=oracle_options(cfg, O, match_eliminated=True))
    assert [i[0] for i in infos] == [0] * 4
    assert [int(i) for i in iters[:4]] == [i[1] for i in infos]
    assert np.abs(nodes[:4] - xo).max() < (1e-5 if cfg.duration >= 10.0 else 1e-6)
    # ... and the plain oracle (every row of the reference's NLP with its multiplier regularised alike) at the bound that
    # difference explains
    xo, infos = _oracle_solve(O, start[:4], goal[:4], opts=oracle_options(cfg, O))
    assert [int(i) for i in iters[:4]] == [i[1] for i in infos]
    assert np.abs(nodes[:4] - xo).max() < (1e-5 if cfg.duration >= 8.0 else 1e-6)
    P.close()


@pytest.mark.gpu
def test_two_phase_solve_holds_the_footholds():
    """Two-phase solve (default): the first Newton iterations place the feet; once an iterate (the second
    or a later one) is within `foothold_hold_tol` of feasibility the stance footholds stay put and
    the solve finishes as a fixed-foothold problem.  On the exp_5 ledges every
    problem of the batch then converges within a handful of iterations (with free footholds a few
    cycle across a ledge edge until the stall rule stops them), the footholds of iterate 2 are the
    footholds of the solution, the solution is feasible on the true terrain, and the oracle -- same
    rule -- takes the same iterations to the same nodes."""
    import dataclasses
    from oracle.oracle import Oracle, oracle_dict
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.knots100()
    assert (cfg.foothold_hold_from, cfg.foothold_hold_weight, cfg.foothold_hold_tol) == (2, 1e6, 0.25)
    hxy, cell = workloads.exp5_terrain()
    start, goal = workloads.step_goals(256, seed=1, terrain=(hxy, cell))
    P = Planner(cfg, max_batch=256)
    P.set_heightfields(hxy, cell)
    nodes, status, iters, viol = P.plan(start, goal)
    assert (status == 0).all() and iters.max() <= 6 and viol.max() <= cfg.tol
    P2 = Planner(dataclasses.replace(cfg, max_iter=2, stall_iters=0), max_batch=256)
    P2.set_heightfields(hxy, cell)
    nodes2, _, it2, viol2 = P2.plan(start, goal)
    P2.close()
    assert (it2 == 2).all()
    early = viol2 <= cfg.foothold_hold_tol   # held from iterate 2 on
    assert early.sum() >= 200
    d = P.dims
    off = 2 * 6 * d.n_base_nodes            # ee-motion sets follow the two base sets (logs/towr_log.out:99-110)
    for e in range(4):
        for s in range(1, 5):               # stance nodes 1..4 (node 0 is the fixed start stance)
            xy, xy2 = nodes[:, off + 35 * e + 8 * s: off + 35 * e + 8 * s + 2], nodes2[:, off + 35 * e + 8 * s: off + 35 * e + 8 * s + 2]
            assert np.abs(xy - xy2)[early].max() < 1e-5
    O = Oracle(oracle_dict(cfg), height=hxy, hcell=cell)
    same = 0
    for b in range(6):
        assert O.max_violation(nodes[b]) <= cfg.tol + 1e-9
        s, g = start[b], goal[b]
        xo, info = O.solve(O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g, (0, 0, 0), (0, 0, 0), 0.0))
        assert info.status == 0
        if info.iters == iters[b]:
            assert np.abs(nodes[b] - xo).max() < 1e-5
            same += 1
    assert same == 6   # (round 2 gated 5 of 6)
    # flat ground: same iteration count with and without the hold
    Pf = Planner(dataclasses.replace(cfg, foothold_hold_from=0), max_batch=64)
    sf, gf = workloads.flat_goals(64, seed=0)
    _, st_f, it_f, _ = Pf.plan(sf, gf)
    Pf.close()
    P.set_heightfields(np.zeros((40, 20)), 0.1)
    _, st_h, it_h, _ = P.plan(sf, gf)
    assert (st_f == 0).all() and (st_h == 0).all() and np.array_equal(it_f, it_h)
    P.close()


@pytest.mark.gpu
def test_nominal_plan_table_as_starting_point(oracle):
    """Optional starting point for cold solves: the bilinear interpolation of a small table of nominal
    plans (rest start at the origin, 5 x 3 goals, solved once) shifted to the problem's start state.
    On flat ground the interpolated plan is so close that ONE Newton iteration reaches 1e-4 for the
    whole seeded batch (four from towr's straight line); the oracle started from the same point takes
    the same iteration to the same nodes; without a table nothing changes; non-nominal start stances
    and goals outside the grid still converge."""
    from qtos_amd import workloads
    from qtos_amd.capi import Planner
    from qtos_amd.config import PlannerConfig
    cfg = PlannerConfig.reference_compat()
    P = Planner(cfg, max_batch=64)
    start, goal = workloads.flat_goals(64, seed=0)
    cold_nodes, cold_status, cold_iters, _ = P.plan(start, goal)
    guess0 = P.initial_guess(start[:2], goal[:2])
    lo, hi = oracle.var_bounds(oracle.problem(start[0][0:3], start[0][3:6], start[0][6:18].reshape(4, 3), goal[0]))
    # (the starting point of a solve: towr's straight-line guess with -- reduce_swing -- the swing mid nodes on the swing rule)
    assert oracle.swing_start_on_rule
    assert np.abs(guess0[0] - oracle.start_point(oracle.problem(start[0][0:3], start[0][3:6], start[0][6:18].reshape(4, 3), goal[0]))).max() < 1e-12
    dx, dy = P.build_init_table()
    assert (len(dx), len(dy)) == (5, 3) and P.init_table[2].shape == (3, 5, P.n)
    nodes, status, iters, viol = P.plan(start, goal)
    assert (status == 0).all() and viol.max() <= cfg.tol
    assert iters.max() <= 2 and iters.mean() < 0.5 * cold_iters.mean()
    guess = P.initial_guess(start, goal)
    fx = lo == hi
    assert np.abs(guess[0, fx] - lo[fx]).max() == 0.0            # fixed variables carry the problem's own data
    for b in range(3):
        s, g = start[b], goal[b]
        xo, info = oracle.solve(oracle.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g), x0=guess[b])
        assert info.status == 0 and info.iters == iters[b]
        assert np.abs(nodes[b] - xo).max() < 1e-6
    # plans from the table guess are plans of the same problems: close to the cold-start ones
    assert np.abs(P.sample(nodes[:4], 0.0, hz=100.0)[:, :, 1:4] - P.sample(cold_nodes[:4], 0.0, hz=100.0)[:, :, 1:4]).max() < 0.05
    # a start stance that is not the nominal one, a yawed start, a goal outside the grid
    s2, g2 = start[:3].copy(), goal[:3].copy()
    s2[0, 6:18] += np.tile([0.02, -0.015, 0.0], 4)
    s2[1, 5] = 0.1
    g2[2, 0] = s2[2, 0] + 0.9
    n2, st2, it2, v2 = P.plan(s2, g2)
    assert (st2[:2] == 0).all() and it2[:2].max() <= 4
    P.set_init_table()
    n3, st3, it3, _ = P.plan(start, goal)
    assert np.array_equal(n3, cold_nodes) and np.array_equal(it3, cold_iters)
    P.close()
