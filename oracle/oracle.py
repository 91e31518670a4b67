"""ctypes binding of the CPU oracle (oracle/qtos_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libqtos_oracle.so")
NEE, MAX_PHASES = 4, 32


class QoParams(C.Structure):
    _fields_ = [
        ("n_phases", C.c_int * NEE),
        ("phase_dur", (C.c_double * MAX_PHASES) * NEE),
        ("dt_base", C.c_double), ("dt_dyn", C.c_double), ("dt_rom", C.c_double),
        ("force_polys_per_stance", C.c_int),
        ("mass", C.c_double), ("gravity", C.c_double), ("Ib", C.c_double * 9),
        ("nominal_stance", (C.c_double * 3) * NEE),
        ("max_dev", C.c_double * 3),
        ("mu", C.c_double), ("f_max", C.c_double), ("t_swing_avg", C.c_double),
        ("height", C.POINTER(C.c_double)),
        ("hnx", C.c_int), ("hny", C.c_int),
        ("hcell", C.c_double), ("hx0", C.c_double), ("hy0", C.c_double),
        ("terrain_mode", C.c_int),
    ]


class QoProblem(C.Structure):
    _fields_ = [
        ("s", C.c_double * 3), ("s_ang", C.c_double * 3), ("ee", (C.c_double * 3) * NEE),
        ("s_vel", C.c_double * 3), ("s_ang_vel", C.c_double * 3), ("g", C.c_double * 3),
        ("t0", C.c_double),
    ]


class QoLayout(C.Structure):
    _fields_ = [
        ("n_base_nodes", C.c_int),
        ("off_lin", C.c_int), ("off_ang", C.c_int), ("off_eem", C.c_int * NEE),
        ("off_eef", C.c_int * NEE), ("n_vars", C.c_int),
        ("n_eem", C.c_int * NEE), ("n_eef", C.c_int * NEE),
        ("off_terrain", C.c_int * NEE), ("off_dyn", C.c_int), ("off_acc_lin", C.c_int),
        ("off_acc_ang", C.c_int), ("off_rom", C.c_int * NEE), ("off_force", C.c_int * NEE),
        ("off_swing", C.c_int * NEE), ("n_cons", C.c_int),
        ("n_dyn_times", C.c_int), ("n_rom_times", C.c_int), ("T", C.c_double),
    ]


class QoOptions(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int), ("tol", C.c_double), ("mu_init", C.c_double),
        ("mu_min", C.c_double), ("delta_x", C.c_double), ("eps_dual", C.c_double), ("slack_push", C.c_double), ("warm_slack_push", C.c_double),
        ("warm_start", C.c_int), ("verbose", C.c_int), ("stall_iters", C.c_int),
        ("hold_from", C.c_int), ("hold_weight", C.c_double), ("hold_tol", C.c_double), ("chord_tol", C.c_double),
        ("stall_alpha", C.c_double), ("chord_max", C.c_int), ("chord_shrink", C.c_double), ("swing_start_on_rule", C.c_int),
        ("eps_dual_swing", C.c_double), ("eps_dual_acc", C.c_double),
        ("mu_superlinear", C.c_int),
    ]


class QoInfo(C.Structure):
    _fields_ = [
        ("status", C.c_int), ("iters", C.c_int), ("inf_pr", C.c_double), ("inf_pr0", C.c_double),
        ("mu", C.c_double), ("factor_secs", C.c_double), ("eval_secs", C.c_double),
    ]


def build(force=False):
    src = os.path.join(_HERE, "qtos_oracle.c")
    hdr = os.path.join(_HERE, "qtos_oracle.h")
    if (force or not os.path.exists(_LIB)
            or os.path.getmtime(_LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        _lib.qo_get_layout.argtypes = [C.POINTER(QoParams), C.POINTER(QoLayout)]
        _lib.qo_var_bounds.argtypes = [C.POINTER(QoParams), C.POINTER(QoProblem), dp, dp]
        _lib.qo_con_bounds.argtypes = [C.POINTER(QoParams), dp, dp]
        _lib.qo_initial_guess.argtypes = [C.POINTER(QoParams), C.POINTER(QoProblem), dp]
        _lib.qo_constraints.argtypes = [C.POINTER(QoParams), dp, dp]
        _lib.qo_jacobian.argtypes = [C.POINTER(QoParams), dp, dp]
        _lib.qo_sample.argtypes = [C.POINTER(QoParams), dp, C.c_double, C.c_double, C.c_int, dp]
        _lib.qo_terrain_height.argtypes = [C.POINTER(QoParams), C.c_double, C.c_double]
        _lib.qo_terrain_height.restype = C.c_double
        _lib.qo_max_violation.argtypes = [C.POINTER(QoParams), dp]
        _lib.qo_max_violation.restype = C.c_double
        _lib.qo_default_options.argtypes = [C.POINTER(QoOptions)]
        _lib.qo_solve.argtypes = [C.POINTER(QoParams), C.POINTER(QoProblem),
                                  C.POINTER(QoOptions), dp, C.POINTER(QoInfo)]
        _lib.qo_ldlt_solve_dense.argtypes = [C.c_int, dp, dp]
        _lib.qo_solve_batch.argtypes = [C.POINTER(QoParams), C.c_int, C.POINTER(QoProblem), C.POINTER(QoOptions),
                                        dp, C.POINTER(QoInfo), C.c_int]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def oracle_dict(cfg):
    """A qtos_amd.config.PlannerConfig in the key names Oracle() expects (moved here from the product package:
    test infrastructure only)."""
    return dict(phase_durations=cfg.phase_durations, nominal_stance=cfg.nominal_stance,
                dt_base=cfg.dt_base, dt_dyn=cfg.dt_dynamic, dt_rom=cfg.dt_range_of_motion,
                force_polys_per_stance=cfg.force_polys_per_stance, mass=cfg.mass,
                gravity=cfg.gravity, inertia_b=cfg.inertia_b, max_dev=cfg.max_deviation,
                mu=cfg.friction, f_max=cfg.force_limit, t_swing_avg=cfg.t_swing_avg,
                terrain_mode=cfg.terrain_mode,
                # the product's reduce_swing (nearest-cell terrain only, model.hpp) places a starting point's swing mid nodes on
                # the swing rule: the oracle's solves start from the same point
                swing_start_on_rule=bool(getattr(cfg, "reduce_swing", False)) and cfg.terrain_mode == 1)


def oracle_options(cfg, O, match_eliminated=False):
    """qo_options of the oracle O with the solver settings of a PlannerConfig.  match_eliminated: the multipliers of the rows the
    product's reduce_swing / reduce_base eliminate from its KKT system get a regularisation of 1e-13 instead of eps_dual (the
    product has no multipliers for them: eps = 0), so the two Newton steps agree to rounding instead of to eps x multiplier."""
    o = O.default_options()
    if match_eliminated:
        if bool(getattr(cfg, "reduce_swing", False)) and cfg.terrain_mode == 1:
            o.eps_dual_swing = 1e-13
        if bool(getattr(cfg, "reduce_base", False)):
            o.eps_dual_acc = 1e-13
    o.max_iter, o.tol, o.mu_init, o.mu_min = cfg.max_iter, cfg.tol, cfg.mu_init, cfg.mu_min
    o.delta_x, o.eps_dual, o.slack_push, o.warm_slack_push = cfg.delta_x, cfg.eps_dual, cfg.slack_push, cfg.warm_slack_push
    o.stall_iters, o.hold_from, o.hold_weight, o.hold_tol = (cfg.stall_iters, cfg.foothold_hold_from,
                                                             cfg.foothold_hold_weight, cfg.foothold_hold_tol)
    o.chord_tol, o.chord_max, o.chord_shrink = cfg.chord_tol, cfg.chord_max, cfg.chord_shrink
    o.stall_alpha = cfg.stall_alpha
    o.mu_superlinear = int(bool(getattr(cfg, "mu_superlinear", False)))
    # (the product's reduce_swing applies with nearest-cell terrain only: model.hpp)
    o.swing_start_on_rule = int(bool(getattr(cfg, "reduce_swing", False)) and cfg.terrain_mode == 1)   # (== O.swing_start_on_rule for an Oracle(oracle_dict(cfg)))
    return o


class Oracle:
    """Thin object wrapper: Oracle(cfg_dict).  cfg keys mirror qo_params."""

    def __init__(self, cfg, height=None, hcell=0.1, hx0=-1.0, hy0=-1.0):
        self.p = QoParams()
        p = self.p
        for e in range(NEE):
            d = cfg["phase_durations"][e]
            p.n_phases[e] = len(d)
            for k, v in enumerate(d):
                p.phase_dur[e][k] = float(v)
            for k in range(3):
                p.nominal_stance[e][k] = float(cfg["nominal_stance"][e][k])
        p.dt_base, p.dt_dyn, p.dt_rom = cfg["dt_base"], cfg["dt_dyn"], cfg["dt_rom"]
        p.force_polys_per_stance = cfg.get("force_polys_per_stance", 3)
        p.mass, p.gravity = cfg["mass"], cfg["gravity"]
        for k, v in enumerate(np.asarray(cfg["inertia_b"], float).reshape(9)):
            p.Ib[k] = v
        for k in range(3):
            p.max_dev[k] = cfg["max_dev"][k]
        p.mu, p.f_max, p.t_swing_avg = cfg["mu"], cfg["f_max"], cfg["t_swing_avg"]
        p.terrain_mode = int(cfg.get("terrain_mode", 0))
        self.swing_start_on_rule = bool(cfg.get("swing_start_on_rule", False))
        self._height = None
        if height is not None:
            self._height = np.ascontiguousarray(height, dtype=np.float64)
            p.height = _dp(self._height)
            p.hnx, p.hny = self._height.shape
            p.hcell, p.hx0, p.hy0 = hcell, hx0, hy0
        self.L = QoLayout()
        rc = lib().qo_get_layout(C.byref(p), C.byref(self.L))
        if rc:
            raise ValueError("qo_get_layout failed rc=%d" % rc)
        self.n, self.m = self.L.n_vars, self.L.n_cons

    @staticmethod
    def problem(s, s_ang, ee, g, s_vel=(0, 0, 0), s_ang_vel=(0, 0, 0), t0=0.0):
        q = QoProblem()
        for k in range(3):
            q.s[k], q.s_ang[k], q.g[k] = s[k], s_ang[k], g[k]
            q.s_vel[k], q.s_ang_vel[k] = s_vel[k], s_ang_vel[k]
            for e in range(NEE):
                q.ee[e][k] = ee[e][k]
        q.t0 = t0
        return q

    def var_bounds(self, q):
        lo, hi = np.empty(self.n), np.empty(self.n)
        lib().qo_var_bounds(C.byref(self.p), C.byref(q), _dp(lo), _dp(hi))
        return lo, hi

    def con_bounds(self):
        lo, hi = np.empty(self.m), np.empty(self.m)
        lib().qo_con_bounds(C.byref(self.p), _dp(lo), _dp(hi))
        return lo, hi

    def initial_guess(self, q):
        x = np.empty(self.n)
        lib().qo_initial_guess(C.byref(self.p), C.byref(q), _dp(x))
        return x

    def constraints(self, x):
        x = np.ascontiguousarray(x, float)
        g = np.empty(self.m)
        lib().qo_constraints(C.byref(self.p), _dp(x), _dp(g))
        return g

    def jacobian(self, x):
        x = np.ascontiguousarray(x, float)
        J = np.empty((self.m, self.n))
        lib().qo_jacobian(C.byref(self.p), _dp(x), _dp(J))
        return J

    def sample(self, x, t0=0.0, hz=1000.0, n_rows=None):
        x = np.ascontiguousarray(x, float)
        if n_rows is None:
            n_rows = int(round(self.L.T * hz)) + 1
        rows = np.empty((n_rows, 37))
        lib().qo_sample(C.byref(self.p), _dp(x), t0, hz, n_rows, _dp(rows))
        return rows

    def max_violation(self, x):
        x = np.ascontiguousarray(x, float)
        return lib().qo_max_violation(C.byref(self.p), _dp(x))

    def terrain_height(self, x, y):
        return lib().qo_terrain_height(C.byref(self.p), x, y)

    def default_options(self):
        o = QoOptions()
        lib().qo_default_options(C.byref(o))
        o.swing_start_on_rule = int(self.swing_start_on_rule)
        return o

    def start_point(self, q):
        """towr's straight-line guess as a solve starts from it: with swing_start_on_rule the swing mid nodes on the swing rule."""
        x = self.initial_guess(q)
        if self.swing_start_on_rule:
            lib().qo_project_swings.argtypes = [C.POINTER(QoParams), C.POINTER(C.c_double)]
            if lib().qo_project_swings(C.byref(self.p), _dp(x)):
                raise RuntimeError("qo_project_swings failed")
        return x

    def project_swings(self, x):
        """A copy of x with the swing mid nodes on the swing rule (what a warm start becomes under swing_start_on_rule)."""
        y = np.array(x, dtype=np.float64, copy=True)
        lib().qo_project_swings.argtypes = [C.POINTER(QoParams), C.POINTER(C.c_double)]
        lib().qo_project_swings(C.byref(self.p), _dp(y))
        return y

    def solve(self, q, x0=None, opts=None):
        o = opts or self.default_options()
        x = np.empty(self.n)
        if x0 is not None:
            x[:] = x0
            o.warm_start = 1
        info = QoInfo()
        lib().qo_solve(C.byref(self.p), C.byref(q), C.byref(o), _dp(x), C.byref(info))
        return x, info

    def solve_batch(self, problems, n_threads=0, opts=None):
        """Independent cold solves of a list of problems, OpenMP over the problems (qo_solve_batch)."""
        o = opts or self.default_options()
        n = len(problems)
        qs = (QoProblem * n)(*problems)
        infos = (QoInfo * n)()
        x = np.empty((n, self.n))
        rc = lib().qo_solve_batch(C.byref(self.p), n, qs, C.byref(o), _dp(x), infos, int(n_threads))
        if rc != 0:
            raise RuntimeError("qo_solve_batch failed")
        return x, list(infos)

    @staticmethod
    def release_buffers(n_threads=0):
        """Give back the per-thread dense Jacobians the solver keeps between batches (30 MB each on the 100-knot problem)."""
        lib().qo_release_buffers.argtypes = [C.c_int]
        lib().qo_release_buffers.restype = None
        lib().qo_release_buffers(int(n_threads))

