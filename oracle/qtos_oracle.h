/*
 * qtos_oracle.h -- CPU restatement (plain C99, double precision) of the NLP that QTOS's
 * Docker-backed TOWR/Ipopt local planner solves.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path (the HIP library under
 * quadruped-trajectory-optimization-stack_amd/csrc) never links, imports or calls it.
 *
 * PARITY STATUS: the reference's solver source is absent from /root/reference (solver/ is an
 * empty submodule: github.com/Alexyskoutnev/towr_solo12, fork of ethz-adrl/towr v1.4, unpinned
 * HEAD; ifopt v2.0; Ipopt 3.11.9 + MUMPS -- Dockerfile:12-45, logs/towr_log.out:3,37,88).  The
 * restatement follows the published TOWR v1.4 formulation and is pinned against the artefacts the
 * reference commits for this path:
 *   - NLP dimensions / bound split / set order        logs/towr_log.out:40-52,99-129
 *   - GV1 test/data/traj/gait.csv, GV2 data/traj/towr.csv rows 1254.. (constraint residuals vanish
 *     on the golden trajectories; see tests/test_oracle_golden.py and SURVEY.md 8c P1)
 * The solver *iterates* of Ipopt are not reproducible (no objective, L-BFGS Hessian, unpinned
 * code), so solution parity follows protocol P1-P4 of SURVEY.md 8c, not digit-for-digit equality.
 */
#ifndef QTOS_ORACLE_H
#define QTOS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define QO_NEE 4
#define QO_MAX_PHASES 32
#define QO_MAX_POLYS 260
#define QO_MAX_NODES (QO_MAX_POLYS + 1)

/* Model + transcription parameters (upstream towr Parameters / RobotModel, plus the fork's
 * data-derived SOLO12 constants: SURVEY.md 0.5, 8a-7, 8a-8). */
typedef struct {
  int n_phases[QO_NEE];                     /* odd; every foot starts and ends in stance      */
  double phase_dur[QO_NEE][QO_MAX_PHASES];  /* seconds, sums to T for every foot              */
  double dt_base;                           /* duration_base_polynomial_ (0.1)                */
  double dt_dyn;                            /* dt_constraint_dynamic_ (0.1)                   */
  double dt_rom;                            /* dt_constraint_range_of_motion_ (0.08)          */
  int force_polys_per_stance;               /* 3                                              */
  double mass, gravity, Ib[9];              /* single rigid body                              */
  double nominal_stance[QO_NEE][3];         /* in base frame                                  */
  double max_dev[3];                        /* range-of-motion box half-widths                */
  double mu, f_max, t_swing_avg;            /* friction, normal-force limit, swing ref time   */
  /* terrain: height[ix*hny+iy] at x = hx0 + ix*hcell, y = hy0 + iy*hcell, bilinear, clamped;
   * NULL = flat ground at z = 0 */
  const double *height;
  int hnx, hny;
  double hcell, hx0, hy0;
  int terrain_mode; /* 0 bilinear (C0, exact slope), 1 nearest cell (piecewise constant, zero slope) */
} qo_params;

/* One planning problem = the reference's solver flags (QTOS/utils.py:26 _flags). */
typedef struct {
  double s[3];         /* -s      CoM start                       */
  double s_ang[3];     /* -s_ang  Euler roll,pitch,yaw            */
  double ee[QO_NEE][3];/* -e1..-e4 world foot positions FL,FR,HL,HR */
  double s_vel[3];     /* s_vel                                   */
  double s_ang_vel[3]; /* s_ang_vel (Euler rates)                 */
  double g[3];         /* -g      goal (x,y used; z only seeds)   */
  double t0;           /* -t      time stamp of first CSV row     */
} qo_problem;

typedef struct {
  int n_base_nodes;
  int off_lin, off_ang, off_eem[QO_NEE], off_eef[QO_NEE], n_vars;
  int n_eem[QO_NEE], n_eef[QO_NEE];
  int off_terrain[QO_NEE], off_dyn, off_acc_lin, off_acc_ang, off_rom[QO_NEE], off_force[QO_NEE],
      off_swing[QO_NEE], n_cons;
  int n_dyn_times, n_rom_times;
  double T;
} qo_layout;

/* ---- structure ------------------------------------------------------------------------------ */
int qo_get_layout(const qo_params *p, qo_layout *L);
/* variable bounds; fixed variables have lo == hi (logs/towr_log.out:44: 1040 -> 1005) */
int qo_var_bounds(const qo_params *p, const qo_problem *q, double *lo, double *hi);
/* constraint bounds, +-1e20 = infinite */
int qo_con_bounds(const qo_params *p, double *lo, double *hi);
/* upstream-style initial guess (linear interpolation by node index, fz = m g / 4) */
int qo_initial_guess(const qo_params *p, const qo_problem *q, double *x);

/* ---- functions ------------------------------------------------------------------------------ */
int qo_constraints(const qo_params *p, const double *x, double *g);
/* dense row-major Jacobian n_cons x n_vars */
int qo_jacobian(const qo_params *p, const double *x, double *J);
/* 1 kHz-style sampling: rows[n_rows][37], row k at t = k/hz, time column = t0 + k/hz */
int qo_sample(const qo_params *p, const double *x, double t0, double hz, int n_rows, double *rows);
double qo_terrain_height(const qo_params *p, double x, double y);

/* ---- solver --------------------------------------------------------------------------------- */
typedef struct {
  int max_iter;        /* 40                                                   */
  double tol;          /* primal feasibility (inf-norm) to declare status 0    */
  double mu_init, mu_min;
  double delta_x;      /* proximal weight (W = delta I)                        */
  double eps_dual;     /* quasi-definite regularisation on the equality block  */
  double slack_push;   /* cold-start slack push as a fraction of the bound range */
  double warm_slack_push; /* the same when warm_start is set (Ipopt's 0.01) */
  int warm_start;      /* 1: x_io holds the starting point                     */
  int verbose;
  int stall_iters;     /* stop (status 1, best iterate returned) after this many iterations
                          without a new lowest violation; 0 = never                */
  int hold_from;       /* once an iterate (number >= hold_from) has violation <= hold_tol the stance
                          footholds (x, y) carry the proximal weight hold_weight instead of
                          delta_x for the rest of the solve; 0 = never              */
  double hold_weight, hold_tol;
  double chord_tol;    /* an iterate with violation <= chord_tol that was reached by a full step of a freshly
                          factored system is followed by a chord step: the same factorisation, the right-hand
                          side of the new iterate; 0 = every iteration factors */
  double stall_alpha;  /* a problem whose step length stays below stall_alpha for two iterations in a row is jammed against
                          its bounds (the fraction-to-the-boundary rule leaves it no room: it would sit there until a
                          division overflows): it stops like a stalled one, status 1, best iterate returned; 0 = never */
  int chord_max;       /* chord steps in a row with one factorisation: a further one follows a full chord step that
                          brought the violation down to chord_shrink times what it was (and to chord_tol)      */
  double chord_shrink;
  int swing_start_on_rule; /* 1: the starting point's swing mid nodes (x, y, v_x, v_y) are placed on the swing rule before the
                          first evaluation -- the product's reduce_swing takes those rows out of its KKT system, so they must
                          hold at its first iterate; from there this solver's Newton steps keep them (linear rows) and the two
                          take the same path.  0 = towr's straight-line guess as it is (the logged 19.4 at iteration 0) */
  double eps_dual_swing, eps_dual_acc;
                       /* the regularisation -eps of the multipliers of the swing rows / of the base's acceleration-continuity rows;
                          < 0 (default): eps_dual, like every other equality row.  The product's reduce_swing / reduce_base solve
                          the Newton step with those rows ELIMINATED -- their multipliers do not exist, i.e. eps = 0 for them --, so
                          its step differs from this solver's by eps x multiplier on those rows: 1e-8 x O(1) on a cold start,
                          1e-8 x O(1e3) = 1e-5 at a time-shifted warm start, whose violation is 40 - 250 (round 6,
                          scratch/r6_shift_gap.py: the gap is there after the FIRST step, it is not rounding).  A test that wants to
                          compare the two at rounding level hands this solver a tiny eps for exactly those rows. */
  int mu_superlinear;  /* 1: mu <- max(tol, mu_min, min(0.2 mu, mu sqrt(mu))) behind a step longer than 0.3 (Ipopt's monotone update:
                          mu_linear_decrease_factor 0.2, mu_superlinear_decrease_power 1.5); 0: mu <- max(mu_min, 0.2 mu) */
} qo_options;

typedef struct {
  int status;          /* 0 converged, 1 max-iter, 2 numerical failure         */
  int iters;
  double inf_pr;       /* final max constraint violation (unscaled)            */
  double inf_pr0;      /* violation at the starting point                      */
  double mu;
  double factor_secs, eval_secs;
} qo_info;

void qo_default_options(qo_options *o);
int qo_solve(const qo_params *p, const qo_problem *q, const qo_options *o, double *x_io,
             qo_info *info);
/* n_problems independent solves, OpenMP over the problems (n_threads <= 0: the runtime's default) */
int qo_solve_batch(const qo_params *p, int n_problems, const qo_problem *q, const qo_options *o,
                   double *x_io, qo_info *info, int n_threads);
/* frees the per-thread Jacobian buffers qo_solve keeps between calls and restores the allocator's thresholds */
void qo_release_buffers(int n_threads);
/* the swing mid nodes of x placed on the swing rule (what swing_start_on_rule does to a starting point) */
int qo_project_swings(const qo_params *p, double *x);
/* max violation of all 1730 rows (equalities and two-sided bounds) + violated fixed vars */
double qo_max_violation(const qo_params *p, const double *x);

/* Linear-algebra building block exposed for KKT parity tests: skyline LDL^T of a symmetric
 * quasi-definite matrix given as dense lower triangle (n x n row-major), solve in place. */
int qo_ldlt_solve_dense(int n, const double *A, double *b);

#ifdef __cplusplus
}
#endif
#endif
