"""ctypes binding of the HIP planner library (csrc/libqtos_planner.so, ABI in include/qtos_planner.h).

There is no CPU fallback: if the library is missing or no MI355X is visible, loading / creating a
planner raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C quadruped-trajectory-optimization-stack_amd/csrc``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (QTOS_LIB names another build of the library in csrc/: A/B timing of kernel variants, scratch/ab.py)
LIB_PATH = os.path.join(_HERE, "csrc", os.environ.get("QTOS_LIB", "libqtos_planner.so"))
NEE, MAX_PHASES, START_DOUBLES, CSV_COLS = 4, 32, 24, 37


class QtosParams(C.Structure):
    _fields_ = [
        ("n_phases", C.c_int * NEE),
        ("phase_dur", (C.c_double * MAX_PHASES) * NEE),
        ("dt_base", C.c_double), ("dt_dyn", C.c_double), ("dt_rom", C.c_double),
        ("force_polys_per_stance", C.c_int),
        ("mass", C.c_double), ("gravity", C.c_double), ("inertia_b", C.c_double * 9),
        ("nominal_stance", (C.c_double * 3) * NEE), ("max_dev", C.c_double * 3),
        ("mu", C.c_double), ("f_max", C.c_double), ("t_swing_avg", C.c_double),
        ("honor_start_velocity", C.c_int), ("terrain_mode", C.c_int),
        ("max_iter", C.c_int),
        ("tol", C.c_double), ("mu_init", C.c_double), ("mu_min", C.c_double),
        ("delta_x", C.c_double), ("eps_dual", C.c_double), ("slack_push", C.c_double), ("warm_slack_push", C.c_double),
        ("stall_iters", C.c_int), ("hold_from", C.c_int), ("hold_weight", C.c_double), ("hold_tol", C.c_double),
        ("chord_tol", C.c_double),
        ("reduce_base", C.c_int),
        ("chord_max", C.c_int), ("chord_shrink", C.c_double), ("stall_alpha", C.c_double),
        ("reduce_swing", C.c_int),
        ("mu_superlinear", C.c_int),
    ]


class QtosDims(C.Structure):
    _fields_ = [
        ("n_vars", C.c_int), ("n_cons", C.c_int),
        ("n_free", C.c_int), ("n_eq", C.c_int), ("n_ineq", C.c_int),
        ("n_ineq_lower", C.c_int), ("n_ineq_both", C.c_int), ("n_ineq_upper", C.c_int),
        ("n_eq_work", C.c_int), ("n_unknowns", C.c_int),
        ("n_stages", C.c_int), ("pivots", C.c_int), ("front", C.c_int),
        ("n_base_nodes", C.c_int), ("n_dyn_times", C.c_int), ("n_rom_times", C.c_int),
        ("n_rows_csv", C.c_int),
        ("panel_doubles", C.c_longlong), ("g_doubles", C.c_longlong),
        ("kkt_algorithmic_bytes", C.c_longlong), ("kkt_flops", C.c_longlong),
        ("envelope", C.c_longlong), ("max_active", C.c_int), ("order_rule", C.c_int),
        ("duration", C.c_double),
    ]


EXPORTS = [
    "qtos_planner_create", "qtos_planner_destroy", "qtos_planner_dims", "qtos_last_error",
    "qtos_set_heightfields", "qtos_plan_batch", "qtos_plan_batch_device", "qtos_sample_csv",
    "qtos_sample_csv_device", "qtos_last_timing", "qtos_debug_eval", "qtos_debug_newton",
    "qtos_debug_structure", "qtos_debug_trace", "qtos_debug_factor", "qtos_analyze", "qtos_analyze_sweep", "qtos_analyze_kron",
    "qtos_set_init_table", "qtos_debug_initial_guess", "qtos_shift_warm", "qtos_shift_warm_device",
    "qtos_last_timing_chord", "qtos_debug_chord", "qtos_plan_totals",
    "qtos_plan_submit", "qtos_plan_poll", "qtos_plan_wait", "qtos_set_speculation", "qtos_debug_residual", "qtos_project_nodes",
    "qtos_debug_stream_len", "qtos_debug_read_stream", "qtos_debug_read_rhs", "qtos_build_flags", "qtos_kkt_kernel",
    "qtos_last_timing_detail", "qtos_set_pattern_speculation", "qtos_env", "qtos_analyze_two_ended", "qtos_analyze_order", "qtos_set_kernel_events",
    "qtos_write_csv",
]

_lib = None


def load():
    """Load the HIP library; raises OSError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("HIP planner library not built: %s (run __graft_entry__.build()); "
                      "this package has no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  If this library
    # pulled in the system copies first, a later `import torch` would map a second runtime and see no GPU
    # (torch.cuda.is_available() False, RCCL unusable).  Binding to torch's copy -- when torch is installed --
    # makes the import order irrelevant; without torch the system runtime is used.
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        for loc in (spec.submodule_search_locations or []) if spec else []:
            hip = os.path.join(loc, "lib", "libamdhip64.so")
            if os.path.exists(hip):
                C.CDLL(hip, mode=C.RTLD_GLOBAL)
                break
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    _warn_if_two_hip_runtimes()
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    lib.qtos_planner_create.argtypes = [C.POINTER(QtosParams), C.c_int, C.c_int, C.POINTER(vp)]
    lib.qtos_planner_destroy.argtypes = [vp]
    lib.qtos_planner_destroy.restype = None
    lib.qtos_planner_dims.argtypes = [vp, C.POINTER(QtosDims)]
    lib.qtos_last_error.argtypes = [vp]
    lib.qtos_last_error.restype = C.c_char_p
    lib.qtos_set_heightfields.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
    lib.qtos_plan_batch.argtypes = [vp, C.c_int, dp, dp, ip, dp, dp, ip, ip, dp]
    lib.qtos_plan_batch_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.qtos_sample_csv.argtypes = [vp, C.c_int, dp, dp, C.c_double, C.c_int, dp]
    lib.qtos_write_csv.argtypes = [C.c_char_p, dp, C.c_int, C.c_int]
    lib.qtos_sample_csv_device.argtypes = [vp, C.c_int, vp, vp, C.c_double, C.c_int, vp, vp]
    lib.qtos_last_timing.argtypes = [vp, dp, ip, dp, ip]
    lib.qtos_debug_eval.argtypes = [vp, C.c_int, dp, dp, ip, dp, dp, dp]
    lib.qtos_debug_newton.argtypes = [vp, C.c_int, dp, dp, ip, dp, dp, dp, dp]
    lib.qtos_debug_structure.argtypes = [vp, ip, ip, ip]
    lib.qtos_debug_trace.argtypes = [vp, C.c_int, dp]
    lib.qtos_debug_factor.argtypes = [vp, C.c_int, dp, ip]
    lib.qtos_analyze.argtypes = [C.POINTER(QtosParams), C.POINTER(QtosDims), ip, C.c_int]
    lib.qtos_analyze_sweep.argtypes = [C.POINTER(QtosParams), ip, ip, ip, ip, ip, C.c_int]
    lib.qtos_analyze_kron.argtypes = [C.POINTER(QtosParams), ip, ip, ip, dp]
    lib.qtos_set_init_table.argtypes = [vp, C.c_int, dp, C.c_int, dp, dp]
    lib.qtos_debug_initial_guess.argtypes = [vp, C.c_int, dp, dp, ip, dp]
    if hasattr(lib, "qtos_last_timing_chord"):
        lib.qtos_last_timing_chord.argtypes = [vp, dp, ip]
    if hasattr(lib, "qtos_plan_totals"):
        lib.qtos_plan_totals.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int]
        lib.qtos_debug_chord.argtypes = [vp, C.c_int, dp]
    if hasattr(lib, "qtos_project_nodes"):
        lib.qtos_project_nodes.argtypes = [vp, C.c_int, dp, dp]
    if hasattr(lib, "qtos_debug_residual"):
        lib.qtos_debug_residual.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    if hasattr(lib, "qtos_plan_submit"):
        lib.qtos_plan_submit.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.qtos_plan_poll.argtypes = [vp, ip]
        lib.qtos_plan_wait.argtypes = [vp]
        lib.qtos_set_speculation.argtypes = [vp, C.c_int]
    if hasattr(lib, "qtos_last_timing_detail"):   # (round 6)
        lib.qtos_last_timing_detail.argtypes = [vp, dp, C.c_int]
        lib.qtos_set_pattern_speculation.argtypes = [vp, C.c_int]
        if hasattr(lib, "qtos_set_kernel_events"):
            lib.qtos_set_kernel_events.argtypes = [vp, C.c_int]
        lib.qtos_env.argtypes = [vp, C.c_char_p, C.c_int]
    if hasattr(lib, "qtos_analyze_two_ended"):
        lib.qtos_analyze_two_ended.argtypes = [C.POINTER(QtosParams), ip, C.c_int]
    if hasattr(lib, "qtos_analyze_order"):
        lib.qtos_analyze_order.argtypes = [C.POINTER(QtosParams), ip, C.c_int]
    if hasattr(lib, "qtos_shift_warm"):   # (absent from older builds loaded through QTOS_LIB for A/B timing)
        lib.qtos_shift_warm.argtypes = [vp, C.c_int, dp, dp, dp, dp, ip, dp]
        lib.qtos_shift_warm_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    _lib = lib
    return lib


def _warn_if_two_hip_runtimes():
    """The pre-load above gives ONE HIP runtime only if torch's libamdhip64 has the SONAME this library was linked against:
    with a torch wheel built for another ROCm major both copies get mapped and the kernels register with whichever wins
    symbol lookup.  Said loudly instead of failing obscurely later."""
    try:
        paths = {ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln}
        real = {os.path.realpath(q) for q in paths}
        if len(real) > 1:
            import warnings
            warnings.warn("two HIP runtimes are mapped into this process (%s): the planner library and PyTorch do not share "
                          "one libamdhip64; GPU work may fail or see no device" % ", ".join(sorted(real)))
    except OSError:
        pass


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def params_from_config(cfg):
    p = QtosParams()
    for e in range(NEE):
        d = cfg.phase_durations[e]
        if len(d) > MAX_PHASES:
            raise ValueError("at most %d phases per foot" % MAX_PHASES)
        p.n_phases[e] = len(d)
        for k, v in enumerate(d):
            p.phase_dur[e][k] = float(v)
        for k in range(3):
            p.nominal_stance[e][k] = float(cfg.nominal_stance[e][k])
    p.dt_base, p.dt_dyn, p.dt_rom = cfg.dt_base, cfg.dt_dynamic, cfg.dt_range_of_motion
    p.force_polys_per_stance = cfg.force_polys_per_stance
    p.mass, p.gravity = cfg.mass, cfg.gravity
    for k, v in enumerate(np.asarray(cfg.inertia_b, float).reshape(9)):
        p.inertia_b[k] = v
    for k in range(3):
        p.max_dev[k] = cfg.max_deviation[k]
    p.mu, p.f_max, p.t_swing_avg = cfg.friction, cfg.force_limit, cfg.t_swing_avg
    p.honor_start_velocity = int(cfg.honor_start_velocity)
    p.terrain_mode = int(cfg.terrain_mode)
    p.max_iter, p.tol = cfg.max_iter, cfg.tol
    p.mu_init, p.mu_min, p.delta_x, p.eps_dual = cfg.mu_init, cfg.mu_min, cfg.delta_x, cfg.eps_dual
    p.slack_push = cfg.slack_push
    p.warm_slack_push = cfg.warm_slack_push
    p.stall_iters = cfg.stall_iters
    p.hold_from, p.hold_weight, p.hold_tol = cfg.foothold_hold_from, cfg.foothold_hold_weight, cfg.foothold_hold_tol
    p.chord_tol = cfg.chord_tol
    p.reduce_base = int(cfg.reduce_base)
    p.chord_max, p.chord_shrink = int(cfg.chord_max), float(cfg.chord_shrink)
    p.stall_alpha = float(cfg.stall_alpha)
    p.reduce_swing = int(getattr(cfg, "reduce_swing", False))
    p.mu_superlinear = int(getattr(cfg, "mu_superlinear", False))
    return p


def build_flags():
    """Bit 0: the library contains the experiment kernels (QTOS_KKT=3 / 5, QTOS_KRON); bit 1: stamps; bit 2: development build."""
    return int(load().qtos_build_flags())


def analyze(cfg):
    """Host-only structure analysis: (QtosDims, per-stage populated front sizes).  No GPU needed."""
    lib = load()
    p = params_from_config(cfg)
    d = QtosDims()
    act = np.zeros(4096, np.int32)
    rc = lib.qtos_analyze(C.byref(p), C.byref(d), _ip(act), act.size)
    if rc != 0:
        raise ValueError("qtos_analyze failed (%d)" % rc)
    return d, act[:d.n_stages].copy()


def analyze_order(cfg):
    """Host-only: the elimination order by position (solver variable, n_sol + row for a multiplier, -1 = dummy pivot)."""
    lib = load()
    p = params_from_config(cfg)
    a = np.full(1 << 15, -2, np.int32)
    n = lib.qtos_analyze_order(C.byref(p), _ip(a), a.size)
    if n < 0 or n > a.size:
        raise ValueError("qtos_analyze_order failed (%d)" % n)
    return a[:n].copy()


def analyze_two_ended(cfg):
    """Host-only: what a two-ended elimination of this model's KKT matrix would look like (qtos_analyze_two_ended): a dict of the
    chain lengths, fronts, the separator and the LDS a workgroup running both chains would need."""
    lib = load()
    p = params_from_config(cfg)
    a = np.zeros(20, np.int32)
    rc = lib.qtos_analyze_two_ended(C.byref(p), _ip(a), a.size)
    if rc != 0:
        raise ValueError("qtos_analyze_two_ended failed (%d)" % rc)
    keys = ("stages_now", "front_now", "split_stage", "stages_left", "stages_right", "sep_unknowns", "stages_sep", "front_left", "front_right",
            "front_sep", "serial_steps", "peak_left", "peak_right", "lds_now", "lds_panels", "lds_records", "lds_cells", "lds_two_chains_as_is",
            "lds_two_chains_lean", "lds_limit")
    return {k: int(v) for k, v in zip(keys, a)}


def analyze_kron(cfg):
    """Host-only: (inequality blocks, blocks with the Kronecker structure, most in a record, worst relative difference)."""
    lib = load()
    p = params_from_config(cfg)
    a = [np.zeros(1, np.int32) for _ in range(3)]
    w = np.zeros(1)
    rc = lib.qtos_analyze_kron(C.byref(p), _ip(a[0]), _ip(a[1]), _ip(a[2]), _dp(w))
    if rc != 0:
        raise ValueError("qtos_analyze_kron failed (%d)" % rc)
    return int(a[0][0]), int(a[1][0]), int(a[2][0]), float(w[0])


def analyze_sweep(cfg, max_places=1 << 16):
    """Host-only: the helper waves' schedule of the backward sweep as arrays [rounds, 16]: (rows, entries, pos_min, pos_max)."""
    lib = load()
    p = params_from_config(cfg)
    nr = np.zeros(1, np.int32)
    a = [np.full(max_places, -1, np.int32) for _ in range(4)]
    rc = lib.qtos_analyze_sweep(C.byref(p), _ip(nr), _ip(a[0]), _ip(a[1]), _ip(a[2]), _ip(a[3]), max_places)
    if rc != 0:
        raise ValueError("qtos_analyze_sweep failed (%d)" % rc)
    n = int(nr[0])
    return tuple(x[:16 * n].reshape(n, 16).copy() for x in a)


class Planner:
    """Owning wrapper of a QtosPlanner handle."""

    def __init__(self, cfg, max_batch=256, device=0):
        self.lib = load()
        self.cfg = cfg
        self.params = params_from_config(cfg)
        self.h = C.c_void_p()
        rc = self.lib.qtos_planner_create(C.byref(self.params), max_batch, device, C.byref(self.h))
        if rc != 0:
            raise RuntimeError("qtos_planner_create failed (%d): -2 = no HIP device, -3 = out of "
                               "memory, -4 = front too large" % rc)
        self.max_batch, self.device = max_batch, device
        self.init_table = None
        self.dims = QtosDims()
        self.lib.qtos_planner_dims(self.h, C.byref(self.dims))
        self.n, self.m = self.dims.n_vars, self.dims.n_cons

    def close(self):
        if self.h:
            self.lib.qtos_planner_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.qtos_last_error(self.h).decode()))

    def set_heightfields(self, maps, cell, x0=-1.0, y0=-1.0):
        """maps: (n_maps, nx, ny) heights, maps[k][ix][iy] at x = x0 + ix*cell, y = y0 + iy*cell."""
        if maps is None:
            self._chk(self.lib.qtos_set_heightfields(self.h, 0, None, 0, 0, 1.0, 0.0, 0.0), "set_heightfields")
            return
        a = np.ascontiguousarray(maps, dtype=np.float64)
        if a.ndim == 2:
            a = a[None]
        self._chk(self.lib.qtos_set_heightfields(self.h, a.shape[0], _dp(a), a.shape[1], a.shape[2],
                                                 cell, x0, y0), "set_heightfields")

    def plan(self, start, goal, map_id=None, warm=None):
        start = np.ascontiguousarray(start, np.float64).reshape(-1, START_DOUBLES)
        goal = np.ascontiguousarray(goal, np.float64).reshape(-1, 3)
        B = start.shape[0]
        nodes = np.empty((B, self.n))
        status = np.empty(B, np.int32)
        iters = np.empty(B, np.int32)
        viol = np.empty(B)
        mid = None if map_id is None else np.ascontiguousarray(map_id, np.int32)
        wm = None if warm is None else np.ascontiguousarray(warm, np.float64).reshape(B, self.n)
        self._chk(self.lib.qtos_plan_batch(self.h, B, _dp(start), _dp(goal), _ip(mid), _dp(wm),
                                           _dp(nodes), _ip(status), _ip(iters), _dp(viol)), "plan_batch")
        return nodes, status, iters, viol

    # ---- asynchronous form (device pointers): submit / poll / wait, see include/qtos_planner.h ----
    def submit(self, B, d_start, d_goal, d_map_id, d_warm, d_nodes, d_status, d_iters, d_viol, stream):
        """Queue a whole solve on `stream` (raw device pointers / None, stream = hipStream_t as int) and return at once."""
        self._chk(self.lib.qtos_plan_submit(self.h, B, d_start, d_goal, d_map_id, d_warm, d_nodes, d_status, d_iters, d_viol,
                                            C.c_void_p(stream)), "plan_submit")

    def poll(self):
        """True once everything the submitted call needs has been queued (results: synchronise its stream)."""
        done = C.c_int(0)
        self._chk(self.lib.qtos_plan_poll(self.h, C.byref(done)), "plan_poll")
        return bool(done.value)

    def wait(self):
        self._chk(self.lib.qtos_plan_wait(self.h), "plan_wait")

    def set_speculation(self, max_blind_iterations):
        self._chk(self.lib.qtos_set_speculation(self.h, int(max_blind_iterations)), "set_speculation")

    def set_pattern_speculation(self, on):
        """The launch pattern of qtos_plan_submit on / off (off forgets what was learnt)."""
        self._chk(self.lib.qtos_set_pattern_speculation(self.h, int(bool(on))), "set_pattern_speculation")

    def set_kernel_events(self, on):
        """Per-kernel HIP events (what timing() / timing_detail() read) on / off; off = the call's first and last event only."""
        self._chk(self.lib.qtos_set_kernel_events(self.h, int(bool(on))), "set_kernel_events")

    def env(self):
        """The environment switches the handle runs with (read once at creation), as a dict of strings."""
        buf = C.create_string_buffer(512)
        self._chk(min(self.lib.qtos_env(self.h, buf, 512), 0), "env")
        return dict(kv.split("=", 1) for kv in buf.value.decode().split())

    def timing_detail(self):
        """Where the time of the last call went (qtos_last_timing_detail): seconds and launch counts."""
        a = (C.c_double * 14)()
        self._chk(self.lib.qtos_last_timing_detail(self.h, a, 14), "last_timing_detail")
        keys = ("total_seconds", "start_seconds", "solve_seconds", "step_seconds", "gap_seconds", "slots", "slots_at_submit",
                "informed_launches", "pattern_calls", "pattern_misses", "kkt_seconds", "kkt_launches", "chord_seconds", "chord_launches")
        return {k: (a[i] if i < 5 or i in (10, 12) else int(a[i])) for i, k in enumerate(keys)}

    def sample(self, nodes, t0, hz=1000.0, n_rows=None):
        nodes = np.ascontiguousarray(nodes, np.float64).reshape(-1, self.n)
        B = nodes.shape[0]
        t0 = np.ascontiguousarray(np.broadcast_to(np.asarray(t0, np.float64), (B,)))
        if n_rows is None:
            n_rows = int(round(self.dims.duration * hz)) + 1
        rows = np.empty((B, n_rows, CSV_COLS))
        self._chk(self.lib.qtos_sample_csv(self.h, B, _dp(nodes), _dp(t0), hz, n_rows, _dp(rows)), "sample_csv")
        return rows

    def shift_warm(self, nodes_prev, offset, start, goal, map_id=None):
        """Time-shifted warm start of a replan (qtos_shift_warm): previous plan read `offset` seconds later."""
        nodes_prev = np.ascontiguousarray(nodes_prev, np.float64).reshape(-1, self.n)
        B = nodes_prev.shape[0]
        off = np.ascontiguousarray(np.broadcast_to(np.asarray(offset, np.float64), (B,)))
        start = np.ascontiguousarray(start, np.float64).reshape(B, START_DOUBLES)
        goal = np.ascontiguousarray(goal, np.float64).reshape(B, 3)
        mid = None if map_id is None else np.ascontiguousarray(map_id, np.int32)
        out = np.empty((B, self.n))
        self._chk(self.lib.qtos_shift_warm(self.h, B, _dp(nodes_prev), _dp(off), _dp(start), _dp(goal), _ip(mid), _dp(out)), "shift_warm")
        return out

    # ---- optional: nominal-plan table for the starting point of cold solves ----
    def set_init_table(self, dx=None, dy=None, nodes=None):
        """Install (or, without arguments, remove) a table of nominal plans: nodes[j][i] = the plan from the
        rest start at the origin to the goal (dx[i], dy[j])."""
        if dx is None:
            self._chk(self.lib.qtos_set_init_table(self.h, 0, None, 0, None, None), "set_init_table")
            self.init_table = None
            return
        dx = np.ascontiguousarray(dx, np.float64)
        dy = np.ascontiguousarray(dy, np.float64)
        nodes = np.ascontiguousarray(nodes, np.float64).reshape(len(dy), len(dx), self.n)
        self._chk(self.lib.qtos_set_init_table(self.h, len(dx), _dp(dx), len(dy), _dp(dy), _dp(nodes)), "set_init_table")
        self.init_table = (dx, dy, nodes)

    def build_init_table(self, dx=None, dy=None, z=0.24):
        """Solve the nominal problems (rest start at the origin, nominal stance, current heightfields,
        straight-line guess) on a grid of goal displacements and install them as the table.  Default
        grid: 0.1 .. 0.16 m per second of horizon ahead in five steps, -0.1 .. 0.1 m sideways in three."""
        from . import workloads
        T = self.dims.duration
        dx = np.asarray(dx if dx is not None else np.linspace(0.03, 0.15, 5) * T, np.float64)
        dy = np.asarray(dy if dy is not None else [-0.1, 0.0, 0.1], np.float64)
        self.set_init_table()
        start = np.repeat(workloads.rest_start(0.0, 0.0, z)[None], len(dx) * len(dy), 0)
        goal = np.array([[x, y, z] for y in dy for x in dx])
        nodes = np.empty((len(goal), self.n))
        for i in range(0, len(goal), self.max_batch):
            sl = slice(i, i + self.max_batch)
            nodes[sl], status, _, _ = self.plan(start[sl], goal[sl])
            if (status != 0).any():
                raise RuntimeError("a nominal plan of the table did not converge")
        self.set_init_table(dx, dy, nodes)
        return dx, dy

    def initial_guess(self, start, goal, map_id=None):
        """The starting point a solve without `warm` uses for these problems (tests: the oracle is started
        from the same point)."""
        start = np.ascontiguousarray(start, np.float64).reshape(-1, START_DOUBLES)
        goal = np.ascontiguousarray(goal, np.float64).reshape(-1, 3)
        out = np.empty((start.shape[0], self.n))
        mid = None if map_id is None else np.ascontiguousarray(map_id, np.int32)
        self._chk(self.lib.qtos_debug_initial_guess(self.h, start.shape[0], _dp(start), _dp(goal), _ip(mid), _dp(out)), "initial_guess")
        return out

    def kkt_kernel(self):
        """Name of the factor + solve kernel the planner selected (the one rocprofv3 lists), e.g. 'k_kkt5<128>'."""
        buf = C.create_string_buffer(64)
        self.lib.qtos_kkt_kernel.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        self._chk(min(self.lib.qtos_kkt_kernel(self.h, buf, 64), 0), "kkt_kernel")
        return buf.value.decode()

    def timing(self):
        k, t = C.c_double(), C.c_double()
        nl, it = C.c_int(), C.c_int()
        self._chk(self.lib.qtos_last_timing(self.h, C.byref(k), C.byref(nl), C.byref(t), C.byref(it)), "last_timing")
        out = dict(kkt_seconds=k.value, kkt_launches=nl.value, total_seconds=t.value, iterations=it.value)
        if hasattr(self.lib, "qtos_last_timing_chord"):
            c, nc = C.c_double(), C.c_int()
            self._chk(self.lib.qtos_last_timing_chord(self.h, C.byref(c), C.byref(nc)), "last_timing_chord")
            out.update(chord_seconds=c.value, chord_launches=nc.value)
        return out

    def totals(self, reset=False):
        """(problems returned with status 0, Newton iterations) over all plan calls of this handle since the last reset."""
        c, n = C.c_longlong(), C.c_longlong()
        self._chk(self.lib.qtos_plan_totals(self.h, C.byref(c), C.byref(n), int(bool(reset))), "plan_totals")
        return c.value, n.value

    # ---- introspection (parity tests) ----
    def debug_eval(self, start, goal, nodes, map_id=None, jac=True):
        start = np.ascontiguousarray(start, np.float64).reshape(-1, START_DOUBLES)
        goal = np.ascontiguousarray(goal, np.float64).reshape(-1, 3)
        nodes = np.ascontiguousarray(nodes, np.float64).reshape(-1, self.n)
        B = start.shape[0]
        g = np.empty((B, self.m))
        J = np.empty((B, self.m, self.n)) if jac else None
        mid = None if map_id is None else np.ascontiguousarray(map_id, np.int32)
        self._chk(self.lib.qtos_debug_eval(self.h, B, _dp(start), _dp(goal), _ip(mid), _dp(nodes), _dp(g), _dp(J)), "debug_eval")
        return g, J

    def debug_newton(self, start, goal, nodes, sig, w, map_id=None):
        start = np.ascontiguousarray(start, np.float64).reshape(-1, START_DOUBLES)
        goal = np.ascontiguousarray(goal, np.float64).reshape(-1, 3)
        nodes = np.ascontiguousarray(nodes, np.float64).reshape(-1, self.n)
        B = start.shape[0]
        sig = np.ascontiguousarray(sig, np.float64).reshape(B, self.m)
        w = np.ascontiguousarray(w, np.float64).reshape(B, self.m)
        dx = np.empty((B, self.n))
        mid = None if map_id is None else np.ascontiguousarray(map_id, np.int32)
        self._chk(self.lib.qtos_debug_newton(self.h, B, _dp(start), _dp(goal), _ip(mid), _dp(nodes), _dp(sig), _dp(w), _dp(dx)), "debug_newton")
        return dx

    def debug_chord(self, B):
        """dx of the system of the preceding debug_newton call, solved again by k_chord with the stored factorisation."""
        dx = np.empty((B, self.n))
        self._chk(self.lib.qtos_debug_chord(self.h, B, _dp(dx)), "debug_chord")
        return dx

    def project(self, nodes):
        """What given nodes become when a solve starts from them (reduce_base: their projection onto the space of the base's
        B-spline coefficients; otherwise a copy)."""
        nodes = np.ascontiguousarray(nodes, np.float64).reshape(-1, self.n)
        out = np.empty_like(nodes)
        self._chk(self.lib.qtos_project_nodes(self.h, nodes.shape[0], _dp(nodes), _dp(out)), "project_nodes")
        return out

    def debug_residual(self, B, refine=False):
        """(dx, max |b - K x| / max |b| per problem) of the system of the preceding debug_newton call; refine: after one
        step of iterative refinement through the stored factorisation."""
        dx = np.empty((B, self.n))
        res = np.empty(B)
        self._chk(self.lib.qtos_debug_residual(self.h, B, int(bool(refine)), _dp(dx), _dp(res)), "debug_residual")
        return dx, res

    def structure(self):
        rk = np.empty(self.m, np.int32)
        vf = np.empty(self.n, np.int32)
        order = np.empty(self.dims.n_stages * self.dims.pivots, np.int32)    # by position; -1 = a dummy pivot (short stage)
        self.lib.qtos_debug_structure(self.h, _ip(rk), _ip(vf), _ip(order))
        return rk, vf, order

    def factor(self, b):
        """(panels [n_stages, front + 1, 16], pivot slots [n_stages, 16]) of problem b's last KKT solve."""
        d = self.dims
        pan = np.zeros((d.n_stages, d.front + 1, 16))
        ps = np.zeros((d.n_stages, 16), dtype=np.int32)
        self._chk(self.lib.qtos_debug_factor(self.h, b, _dp(pan), _ip(ps)), "qtos_debug_factor")
        # in memory column c of a V row sits at 4 (c & 3) + (c >> 2) (32 contiguous bytes per lane on the store)
        cols = np.array([4 * (c & 3) + (c >> 2) for c in range(16)])
        pan[:, 1:, :] = pan[:, 1:, :][:, :, cols]
        return pan, ps

    def trace(self, b):
        t = np.zeros((self.cfg.max_iter + 1, 4))
        rows = self.lib.qtos_debug_trace(self.h, b, _dp(t))
        return t[:max(rows, 0)]
