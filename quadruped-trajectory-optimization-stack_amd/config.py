"""Planner configuration: SOLO12 single-rigid-body model, gait schedule, transcription.

Defaults are the ``reference_compat`` set: the constants that make the reference's committed
trajectories (test/data/traj/gait.csv, data/traj/towr.csv) satisfy the NLP's constraints
(checked in tests/test_oracle_golden.py).  The solver that produced
them is not in the reference tree, so values that no artefact pins are named and flagged below.
"""
from dataclasses import dataclass, field
from typing import List

EE_NAMES = ("FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT")  # QTOS/utils.py:13, flag order -e1..-e4

# Un-normalised per-foot phase durations (stance, swing, ..., stance); every list sums to 8.12 and
# is scaled to the plan duration.  Reproduces all 32 contact switch times of both golden plans at
# 1 ms resolution.  Order FL, FR, HL, HR.
REFERENCE_WALK_UNNORMALISED = (
    (0.8, 0.3, 1.7, 0.3, 1.7, 0.3, 1.45, 0.51, 1.06),
    (1.8, 0.3, 1.7, 0.3, 1.7, 0.3, 1.21, 0.51, 0.30),
    (0.3, 0.3, 1.7, 0.3, 1.7, 0.3, 1.70, 0.38, 1.44),
    (1.3, 0.3, 1.7, 0.3, 1.7, 0.3, 1.33, 0.51, 0.68),
)

# Diagonal-pair trot (FL+HR, FR+HL), same stand / swing / stance proportions as the walk above.
# Not pinned by any reference artefact (the committed gait is the walk); provided because
# BASELINE.json names a trot.
TROT_UNNORMALISED = (
    (0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.6),
    (0.6, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3),
    (0.6, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3),
    (0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.6),
)

# Inertia tensor as the reference solver evidently used it: data/config/solo12.yml:13-18 lists
# (ixx, ixy, ixz, iyy, iyz, izz) and the values went into an (Ixx, Iyy, Izz, Ixy, Ixz, Iyz)
# constructor.  With this tensor the angular dynamics residual on gait.csv is 5e-5 N m; with the
# physical diagonal tensor it is 0.48 N m.
INERTIA_REFERENCE_COMPAT = ((0.00578574, -0.01938108, 0.0),
                            (-0.01938108, 0.0, -0.02476124),
                            (0.0, -0.02476124, 0.0))
INERTIA_PHYSICAL = ((0.00578574, 0.0, 0.0), (0.0, 0.01938108, 0.0), (0.0, 0.0, 0.02476124))


def scaled_phases(unnormalised, duration):
    total = sum(unnormalised[0])
    return [[d * duration / total for d in foot] for foot in unnormalised]


@dataclass
class PlannerConfig:
    duration: float = 5.0                  # plan horizon [s]
    gait: str = "walk"                     # "walk" (reference) or "trot"
    dt_base: float = 0.1                   # base polynomial duration (upstream towr default)
    dt_dynamic: float = 0.1                # dynamics collocation spacing
    dt_range_of_motion: float = 0.08       # range-of-motion spacing
    force_polys_per_stance: int = 3
    mass: float = 3.0                      # pinned by gait.csv: 2.99994 +- 0.0012 kg
    gravity: float = 9.80665
    inertia_b: tuple = INERTIA_REFERENCE_COMPAT
    # Range-of-motion box: NOT pinned (no bound is active in the golden plans); this is the
    # smallest round box containing the golden envelope.
    nominal_stance: tuple = ((0.20, 0.17, -0.26), (0.20, -0.17, -0.26),
                             (-0.20, 0.17, -0.26), (-0.20, -0.17, -0.26))
    max_deviation: tuple = (0.07, 0.07, 0.09)
    friction: float = 0.5                  # upstream default; golden max |ft|/fn = 0.367
    force_limit: float = 1000.0            # upstream default; golden max fz = 18.3 N
    t_swing_avg: float = 0.3               # pinned: golden swing mid-node velocity ratio
    # The reference solver ignores s_vel / s_ang_vel (towr.csv rows 1255.. restart from rest);
    # set True to start from the hand-over velocities instead.
    honor_start_velocity: bool = False
    # Heightfield lookup between grid points: 0 = bilinear (continuous, exact slope in the Jacobian),
    # 1 = nearest cell (ledges stay flat, steps are jumps).  The fork's choice is unknown; the
    # reference upsamples its tiles by replication (QTOS/generateHeightField.py:39-56), i.e. its
    # terrain IS piecewise constant, and on the exp_5 steps nearest-cell converges in 4-7
    # iterations where the 9 mm bilinear ramps (slope 2.7) stall a third of the solves.
    terrain_mode: int = 1
    hz: float = 1000.0                     # CSV sampling rate (scripts/run.py consumes 1 kHz rows)
    # solver
    max_iter: int = 24
    tol: float = 1e-4
    mu_init: float = 0.1
    mu_min: float = 1e-9
    delta_x: float = 1e-2
    eps_dual: float = 1e-8
    slack_push: float = 0.2            # cold-start slack push (fraction of the bound range)
    warm_slack_push: float = 0.01      # slack push of a solve started from given nodes (Ipopt's bound_push)
    stall_iters: int = 5               # stop after this many iterations without a new lowest violation (0 = off)
    # A problem whose step length stays below stall_alpha for two iterations in a row is jammed against its bounds (one window in
    # 250 on the randomized heightfields: alpha 0.02, 0.00, 0.00 ... at a violation of 2.5 until a division overflows in the
    # ninth iteration -- and its set of windows waits for it): it stops like a stalled one, status 1, best iterate.  0 = off.
    stall_alpha: float = 1e-2
    # Two-phase solve: the first Newton iterations place the feet; once an iterate (number >=
    # `foothold_hold_from`) has a constraint violation <= `foothold_hold_tol`, the stance footholds are
    # held where they are (proximal weight on their x, y) and the rest of the solve is a
    # fixed-foothold problem.  On piecewise-constant terrain a free foothold that drifts over
    # a ledge edge makes Newton's method cycle (the height it must meet changes under it): with the
    # hold the exp_5 batch converges 256/256 in <= 5 iterations instead of 250/256 in <= 13; flat
    # ground is unaffected (4 iterations either way).  0 = never hold.
    foothold_hold_from: int = 2
    foothold_hold_weight: float = 1e6
    foothold_hold_tol: float = 0.25
    # Chord step: an iterate with violation <= chord_tol reached by a full step of a freshly factored KKT system is
    # followed by one step that reuses that factorisation with the new right-hand side (k_chord).  The flat batch goes
    # 27 -> 6.5 -> 0.11 -> 5.7e-4 -> (chord) 3.9e-5: three factorisations instead of four.  0 = off.  (4e-3 with a second
    # chord step in reserve, below: a chord step takes the violation down to 6-22 % -- 1e-3 sent the problems of a batch that
    # arrive at 1.0e-3 ... 4e-3 to a factorisation the whole batch then waited for: knots200 -16 %, exp_5 -12 % per batch.)
    chord_tol: float = 4e-3
    # A full chord step that brought the violation down to chord_shrink of what it was (and to chord_tol) may be followed by
    # another one with the same factorisation, chord_max in a row: the trot batch goes 0.5 -> 5e-4 -> (chord) 1.1e-4 ->
    # (chord) 2.5e-5 instead of paying a fourth factorisation for the last 10 % above the tolerance.
    chord_max: int = 2
    chord_shrink: float = 1.0 / 3.0
    # Reduced base: inside the KKT solve the base node values are replaced by the coefficients of a clamped cubic B-spline on
    # the same knots -- a basis of exactly the C2 splines the acceleration-continuity rows describe --: no multipliers for
    # those rows, half the base unknowns (2885 -> 1721 unknowns, 181 -> 108 stages on the 100-knot transcription), the same
    # Newton step.  False: every row of the reference's NLP has its multiplier (what the internals' tests pin).
    reduce_base: bool = True
    # Reduced swings (round 5): towr's swing rule -- the x, y of a swing's mid node = centre of the neighbouring footholds, its
    # v_x, v_y = their distance / t_swing_avg -- has constant coefficients like the base's continuity rows: inside the KKT solve
    # the four mid-node variables of every swing are their linear image of the two footholds (8 unknowns per swing leave the
    # system: 108 -> 100 stages on the 100-knot walk, 127 -> 113 on the trot); iterate, results and CSV keep the mid nodes.  The
    # rows hold for every iterate only if they hold for the first: the starting point's mid nodes are placed on the rule (towr's
    # straight-line guess has the plan's mean velocity there).  Nearest-cell terrain only.  False: every swing row keeps its
    # multiplier and towr's guess is the starting point as it is (the reference's logged 19.4 at iteration 0).
    reduce_swing: bool = True
    # Barrier parameter: Ipopt's monotone update mu <- max(tol, min(0.2 mu, mu^1.5)) behind a step longer than 0.3 (its
    # mu_linear_decrease_factor / mu_superlinear_decrease_power defaults -- what the reference's solver runs with) instead of the
    # plain mu <- 0.2 mu of rounds 1 - 4: 0.1 -> 0.02 -> 2.8e-3 -> 1.5e-4 -> 1e-4 (the floor: Ipopt's is a tenth of its own tolerance,
    # 1e-3 in the reference's runs) instead of ... 4e-3 -> 8e-4 -> 1.6e-4 -> ... -> 1e-9.  The
    # third iterate is better centred: every problem of a trot batch converges in four iterations (it was 4: 58 %, 5: 42 %, i.e.
    # five for the batch: +10 % plans/s), flat walk and terrains unchanged (round 5, profiles/r05_experiments/mu_rule.log).
    mu_superlinear: bool = True
    phase_durations: List[List[float]] = field(default=None)

    def __post_init__(self):
        if self.phase_durations is None:
            table = REFERENCE_WALK_UNNORMALISED if self.gait == "walk" else TROT_UNNORMALISED
            self.phase_durations = scaled_phases(table, self.duration)

    @classmethod
    def reference_compat(cls, **kw):
        """51 base nodes, dynamics every 0.1 s: the NLP of logs/towr_log.out (1040 vars, 1730 rows)."""
        return cls(**kw)

    @classmethod
    def knots100(cls, **kw):
        """BASELINE.json configs[1..3]: 100 base polynomials / dynamics knots over the 5 s horizon."""
        kw.setdefault("dt_base", 0.05)
        kw.setdefault("dt_dynamic", 0.05)
        return cls(**kw)

    @classmethod
    def knots200(cls, **kw):
        """BASELINE.json configs[4]: 200 dynamics knots = a 10 s horizon at the knots100 spacing with the
        walk schedule run twice (the final stance of the first cycle merges with the first stance of
        the second: 17 phases per foot).  Scaling the one-cycle table to 10 s instead doubles every
        phase and with it the elimination front (208 slots, over the 128 `k_kkt` is built for)."""
        kw.setdefault("duration", 10.0)
        kw.setdefault("dt_base", 0.05)
        kw.setdefault("dt_dynamic", 0.05)
        if "phase_durations" not in kw:
            table = [list(f[:-1]) + [f[-1] + f[0]] + list(f[1:]) for f in REFERENCE_WALK_UNNORMALISED]
            kw["phase_durations"] = scaled_phases(table, kw["duration"])
        return cls(**kw)

    @classmethod
    def receding_windows(cls, **kw):
        """BASELINE.json configs[4] as the product runs it (`replan.ShiftedWindows`, `bench.py --workload mpc_random`): the 200-knot
        transcription with the two solver settings that belong to REPLANNED windows -- the windows of a set are at different points
        of their solves in every iteration, so a chord step saves no launch (the batch still factors for the others) and a
        discarded one costs an iteration: `chord_tol = 0`; and without a chord step to finish, Ipopt's superlinear decrease of the
        barrier parameter has nothing to gain and lengthens the tail (the slowest of 256 windows takes 9.0 instead of 8.45
        factorisations per replan): `mu_superlinear = False`.  ONE place: the bench, the tests
        (`test_shifted_windows_match_oracle_over_five_replans[receding_windows]`) and library users get the same configuration;
        `knots200()` stays the plain 200-knot transcription with the defaults of every other workload."""
        kw.setdefault("chord_tol", 0.0)
        kw.setdefault("mu_superlinear", False)
        return cls.knots200(**kw)
