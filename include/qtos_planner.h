/*
 * qtos_planner.h -- C ABI of the MI355X-native batched local planner.
 *
 * Drop-in boundary.  In the reference the local planner is a process:
 *     subprocess.run("docker exec <id> ./main " + cmd_args(args))
 *         scripts/main.py:49-50, 90-91, 125-126; scripts/run.py:294-295;
 *         QTOS/generateHeightField.py:385-386   (one NLP per call, 32 callers at once)
 * with the flag set of QTOS/utils.py:26 (_flags), the terrain pushed beforehand as a text file
 * (QTOS/utils.py:21-22) and the result fetched as build/traj.csv (QTOS/utils.py:16,19).
 * Every entry point below names the piece of that process ABI it replaces.  Plain pointers and
 * sizes only; no C++ or torch types; no global state; one planner handle is used by one thread
 * at a time (use one handle per (device, stream)).
 *
 * All floating point is IEEE double (the reference solver is double precision throughout).
 */
#ifndef QTOS_PLANNER_H
#define QTOS_PLANNER_H

#ifdef __cplusplus
extern "C" {
#endif

#define QTOS_NEE 4
#define QTOS_MAX_PHASES 32
#define QTOS_START_DOUBLES 24 /* CoM 3, Euler 3, feet FL FR HL HR 12, lin vel 3, Euler rates 3 */
#define QTOS_CSV_COLS 37

/* Model + transcription + solver parameters.  Replaces the constants compiled into the
 * reference's ./main (towr Parameters / RobotModel of the towr_solo12 fork; values recovered from
 * the committed artefacts) and the Ipopt options. */
typedef struct QtosParams {
  int n_phases[QTOS_NEE];                       /* odd: stance, swing, ..., stance            */
  double phase_dur[QTOS_NEE][QTOS_MAX_PHASES];  /* [s]; each foot sums to the plan duration   */
  double dt_base, dt_dyn, dt_rom;               /* base poly / dynamics / range-of-motion dt  */
  int force_polys_per_stance;
  double mass, gravity, inertia_b[9];
  double nominal_stance[QTOS_NEE][3], max_dev[3];
  double mu, f_max, t_swing_avg;
  int honor_start_velocity; /* 0 = reference behaviour (plans start at rest), 1 = use s_vel */
  int terrain_mode;         /* 0 bilinear heightfield (exact slope), 1 nearest cell (flat ledges) */
  int max_iter;
  double tol, mu_init, mu_min, delta_x, eps_dual;
  double slack_push; /* cold-start slack push, fraction of the bound range (0.2) */
  double warm_slack_push; /* the same for a solve that is given `warm` nodes (or a table guess): Ipopt's 0.01 keeps a
                             feasible warm start where it is; a time-shifted previous plan is only a guess and
                             does better with a larger push; 0 = 0.01 */
  int stall_iters;   /* stop a problem (status 1, best iterate returned) after this many iterations
                        without a new lowest violation; 0 = only the iteration limit stops it */
  int hold_from;     /* two-phase solve: once an iterate (number >= hold_from) has brought the constraint
                        violation down to hold_tol, the stance footholds stay where they are (their x, y
                        get the proximal weight hold_weight instead of delta_x): the first iterations
                        place the feet, the rest of the solve is a fixed-foothold problem; 0 = never */
  double hold_weight, hold_tol;
  double chord_tol;  /* an iterate with violation <= chord_tol that was reached by a full step (alpha = 1) of a
                        freshly factored KKT system is followed by a chord step: the stored factorisation is
                        reused with the right-hand side of the new iterate (k_chord: forward + backward sweep over
                        the factor panels, about a fifth of a factorisation); 0 = every iteration factors */
  int reduce_base;   /* 1: inside the KKT solve the base node values are replaced by the coefficients of a clamped cubic
                        B-spline on the same knots (a basis of the C2 splines the acceleration-continuity rows describe):
                        no multipliers for those rows, half the base unknowns, the same Newton step; 0: every row of the
                        reference's NLP has its multiplier (the formulation the internals' tests pin) */
  int chord_max;     /* chord steps in a row with one factorisation (0 = 1): a further one follows a full chord step that
                        brought the violation down to chord_shrink times what it was (and to chord_tol) -- a solve that a
                        chord step leaves just above the tolerance finishes with a second one instead of a factorisation */
  double chord_shrink; /* (0 = 1/3) */
  double stall_alpha;  /* a problem whose step length stays below stall_alpha for two iterations in a row is jammed against its
                          bounds (it would sit there until a division overflows, and its batch with it): it stops like a
                          stalled one -- status 1, best iterate returned; 0 = never */
  int reduce_swing;    /* 1: the swing rule (towr SwingConstraint: the x, y of a swing's mid node = centre of the neighbouring
                          footholds, its v_x, v_y = their distance / t_swing_avg -- constant coefficients) leaves the KKT system:
                          inside the solve the four mid-node variables of every swing are their linear image of the two footholds
                          (no variables, no multipliers for them: 8 unknowns per swing), exactly as reduce_base treats the base's
                          continuity rows.  The iterate, results and CSV keep the mid nodes.  The rows then hold for every
                          iterate only if they hold for the first: the starting point's mid nodes are placed on the rule
                          (towr's straight-line guess has another v_xy there).  Nearest-cell terrain only (terrain_mode 1);
                          0 = every swing row keeps its multiplier */
  int mu_superlinear;  /* 1: the barrier parameter follows Ipopt's monotone update, mu <- max(tol, mu_min, min(0.2 mu, mu^1.5)) behind a
                          step longer than 0.3 (Ipopt's mu_linear_decrease_factor 0.2 and mu_superlinear_decrease_power 1.5, the
                          defaults the reference's solver runs with; mu^1.5 formed as mu * sqrt(mu)): from mu = 0.02 on the
                          superlinear term is the smaller one.  The floor -- Ipopt's is a tenth of ITS tolerance, 1e-3 in the
                          reference's runs: the same 1e-4 as this planner's tol -- keeps the KKT systems of late iterations
                          as well scaled as those of the fourth (with tol / 10 the GPU <-> oracle gaps of the long horizons
                          doubled).  0: mu <- max(mu_min, 0.2 mu) (rounds 1 - 4) */
} QtosParams;

typedef struct QtosDims {
  int n_vars, n_cons;               /* 1040 / 1730 for the reference transcription           */
  int n_free, n_eq, n_ineq;         /* 1005 / 706 / 1024 (logs/towr_log.out:44-52)            */
  int n_ineq_lower, n_ineq_both, n_ineq_upper; /* 112 / 816 / 96                              */
  int n_eq_work;                    /* equality rows after dropping constant/duplicate rows   */
  int n_unknowns;                   /* unknowns of the KKT system the planner solves (with reduce_base: coefficients in place
                                       of base node values); dummy pivots of short stages are NOT counted: arrays by
                                       position have n_stages * pivots entries */
  int n_stages, pivots, front;      /* chain of n_stages fronts, `pivots` eliminated per stage */
  int n_base_nodes, n_dyn_times, n_rom_times, n_rows_csv;
  long long panel_doubles;          /* factor panel storage per problem                       */
  long long g_doubles;              /* Jacobian block storage per problem                     */
  long long kkt_algorithmic_bytes;  /* w*[sum_k (p+c_k)*p + 2M], SURVEY.md 8d formula          */
  long long kkt_flops;              /* 2*sum_k p*(p+c_k)^2                                     */
  long long envelope;               /* skyline size of K in the elimination order             */
  int max_active;                   /* largest front actually populated                       */
  int order_rule;                   /* time keys of the elimination order the analysis kept: 0 = rounds 1 - 5 (full-base systems),
                                       1 = late force nodes, 2 = early coefficients (round 6; csrc/model.hpp
                                       HostModel::order_rule, QTOS_ORDER)                                                     */
  double duration;
} QtosDims;

typedef struct QtosPlanner QtosPlanner;

/* Build the planner for one transcription (symbolic KKT analysis + device workspaces for up to
 * max_batch problems on HIP device `device`).  Replaces starting the `towr` container
 * (QTOS/utils.py:686-692 DockerInfo).  Returns 0, or <0: -1 bad parameters, -2 no HIP device /
 * HIP error, -3 out of memory, -4 front too large for LDS. */
int qtos_planner_create(const QtosParams *params, int max_batch, int device, QtosPlanner **out);
void qtos_planner_destroy(QtosPlanner *p);
int qtos_planner_dims(const QtosPlanner *p, QtosDims *dims);
const char *qtos_last_error(const QtosPlanner *p);
/* Host-only structure analysis (no GPU needed): the dimensions a planner built from `params`
 * would have; stage_active (may be NULL) receives the populated front size of each stage. */
int qtos_analyze(const QtosParams *params, QtosDims *dims, int *stage_active, int max_stages);
/* Host-only: the schedule by which the idle waves of the KKT kernels' backward sweep form the slack steps ds = Ji dx
 * (round 4; replaces a pass of the line-search kernel).  Per place of a round (16 places per round; round i runs while
 * the sweep solves stage n_stages - 1 - i): constraint row (-1: empty), entries of the row, smallest / largest position
 * of its columns in elimination order.  Arrays may be NULL.  0, or -4: no schedule for this transcription, -5: the packed
 * copy of the column positions disagrees with the list. */
int qtos_analyze_sweep(const QtosParams *params, int *n_rounds, int *rows, int *entries, int *pos_min, int *pos_max, int max_places);
/* Host-only: the elimination order by position, as qtos_debug_structure reports it for a planner (a solver variable's index,
 * n_sol + row for a multiplier -- n_sol = n_vars without reduce_base --, -1 for a dummy pivot): up to max_positions entries into
 * `order`; returns the number of positions (n_stages * pivots) or < 0.  tests/test_order_stability.py eliminates the KKT matrix
 * in this order with numpy, without pivoting, as the chain of fronts does. */
int qtos_analyze_order(const QtosParams *params, int *order, int max_positions);
/* Host-only (round 6, analysis only -- no kernel follows this order yet): what a TWO-ENDED elimination of this model's KKT matrix
 * would look like -- a chain from t = 0 forward, a chain from t = T backward, the unknowns alive across the split time (the
 * separator) last -- and the LDS a workgroup that runs both chains would need.  out (n_out >= 20 ints):
 *   [0] stages today  [1] front today  [2] split stage  [3] stages of chain L  [4] of chain R  [5] separator unknowns
 *   [6] separator stages  [7] front of L  [8] of R  [9] of the separator  [10] serial steps = max([3], [4]) + [6]
 *   [11] populated peak of L  [12] of R;  LDS bytes: [13] today's kernel, of which [14] panels, [15] record buffers, [16] cells;
 *   [17] two chains with today's layouts  [18] two chains with a lean layout (two panels + the blanked copy, one dynamic-only
 *   record buffer per chain, gather tables from L2)  [19] the limit (160 KB - 256 B).  DESIGN.md section 5. */
int qtos_analyze_two_ended(const QtosParams *params, int *out, int n_out);
/* Host-only: the Kronecker structure of the range-of-motion blocks (experiment QTOS_KRON of round 4): inequality blocks in all,
 * blocks with the structure, the most of them in one stage record, and the largest relative difference between an entry formed
 * through a block's 33 sums and the direct three-term sum on random data. */
int qtos_analyze_kron(const QtosParams *params, int *n_blocks, int *n_kron, int *max_in_record, double *worst);

/* Terrain side channel.  Replaces `docker cp towr_heightfield.txt <id>:...`
 * (QTOS/utils.py:21-22; scripts/main.py:77-78; QTOS/generateHeightField.py:276-279).
 * n_maps height grids of identical shape, height[map][ix*hny+iy] at x = x0+ix*cell,
 * y = y0+iy*cell (the file's row = x index, column = y index: QTOS/generateHeightField.py:568,
 * 598-605).  n_maps = 0 restores flat ground. */
int qtos_set_heightfields(QtosPlanner *p, int n_maps, const double *height, int hnx, int hny,
                          double cell, double x0, double y0);

/* Optional table of nominal plans for the starting point of cold solves (no reference counterpart: the
 * reference's solver always starts from towr's straight-line guess, and so does this planner
 * unless a table is set).  nodes[(j * ndx + i) * n_vars ..] = a solved plan from the rest start at the
 * origin (nominal stance) to the goal (dx[i], dy[j]); grids strictly increasing.  A solve without
 * `warm` then starts from the bilinear interpolation of the table over its goal displacement,
 * shifted to its own start state, and is treated like a warm start.  ndx = 0 removes the table.
 * Host pointers. */
int qtos_set_init_table(QtosPlanner *p, int ndx, const double *dx, int ndy, const double *dy,
                        const double *nodes);

/* One batched solve = B invocations of `./main -g .. -s .. -s_ang .. -e1..-e4 .. [s_vel ..]
 * [s_ang_vel ..]`.  Host-pointer form: copies in, solves on the GPU, copies out.
 *   start   B x 24   CoM, Euler, feet FL FR HL HR (world), CoM velocity, Euler rates
 *                    (exactly columns 1..24 of a CSV row, QTOS/combiner.py:267-274)
 *   goal    B x 3    -g
 *   map_id  B        heightfield index per problem, NULL = map 0
 *   warm    B x n_vars or NULL: starting nodes (receding-horizon warm start)
 *   nodes_out  B x n_vars  solution in the reference NLP's variable order
 *                          (logs/towr_log.out:99-110)
 *   status_out B   0 = solved (the reference's exit code / "status -> 0"), 1 = iteration limit,
 *                  2 = numerical failure (non-finite inputs; or a non-finite step, nodes_out is
 *                  then the best finite iterate)
 *   iters_out B, viol_out B (max constraint violation), either may be NULL
 * Returns 0 or a negative error code; never throws. */
int qtos_plan_batch(QtosPlanner *p, int B, const double *start, const double *goal,
                    const int *map_id, const double *warm, double *nodes_out, int *status_out,
                    int *iters_out, double *viol_out);

/* Same, all pointers in device memory of the planner's device, work queued on `stream` (a hipStream_t passed
 * as void*, NULL = default stream) -- the asynchronous form of the boundary, in three entry points that replace the
 * reference's pool of `docker exec ./main` workers pulling probes from a queue (QTOS/generateHeightField.py:344-352,
 * 375-377; scripts/main.py:49-50 for the single call):
 *
 *   qtos_plan_submit   queues the initial guess and the LAUNCH PATTERN of the handle -- per launch slot the solve kernels
 *                      its last two calls both needed there (a flat walk or trot batch: factorisation x 3, chord solve,
 *                      every time) -- and returns at once: the whole solve is in the queue, the host is not part of the
 *                      Newton loop (round 6).  The first call of a handle queues its first iteration only.  A problem's
 *                      result is written to the output buffers by the kernel that finishes it -- converged, stalled,
 *                      failed or out of iterations --: there is no export step.  A problem that finds the wrong solve
 *                      kernel in a slot of the pattern sits that launch out and takes its step behind a later one: the
 *                      plans are bit for bit those of the informed loop, whatever was queued.
 *                      (qtos_set_speculation(n > 1), rounds 3 - 5: n "blind" iterations instead -- both solve kernels in
 *                      every slot; switches the pattern off.)
 *   qtos_plan_poll     non-blocking: reads the counts of unfinished problems the iterations sent back; a batch that
 *                      needs more iterations gets them queued one by one (only the kernels with work);
 *                      *done = 1 once the counts say that every problem is finished (the results are then in the
 *                      output buffers: the counts travel behind the kernel that wrote them).
 *   qtos_plan_wait     polls until done.
 *
 * qtos_plan_batch_device = submit + wait: it returns when the last iteration has reported that no problem is left; work
 * queued on `stream` afterwards sees the results, and so does the host after synchronising `stream`.  For a batch whose problems all finish within the blind
 * slots of the pattern the host does nothing between the submit and the end but wait for one word.  One planner handle serves one call at a time (a second submit before the first is done returns -5);
 * handles are independent: several handles on their own streams keep several batches in flight from ONE host thread
 * (qtos_amd.pool.PlannerPool: submit to a free handle, poll the others) -- a batch that waits for its slowest problem
 * then shares the GPU with the next ones.  This is the form bench.py times.
 *
 * Environment.  Read ONCE, by qtos_planner_create (and by the host-only qtos_analyze* calls for themselves), in one place
 * (csrc/env.hpp); the planner keeps what it found and qtos_env() hands it back.  Diagnostics and measured alternatives, the
 * defaults are the measured optimum:
 *   QTOS_KKT=2 | 4 | 6       force the factor + solve kernel: k_kkt2 / k_kkt3 MODE 1 / k_kkt5 (default: k_kkt3 MODE 1 for fronts of up
 *                            to 112 slots, k_kkt2 above; see qtos_kkt_kernel below and DESIGN.md section 5).  3 and 5 (k_kkt3 MODE 0,
 *                            k_kkt4) exist in experiment builds only (qtos_build_flags bit 0) and mean "default" elsewhere
 *   QTOS_LANES=n             a call of more problems than the GPU has compute units is cut into up to n (<= 4) contiguous parts,
 *                            each with its own host-driven loop on a stream of the planner; bit-identical plans; default 1
 *                            (measured slower than one lock-step loop at 1024 problems per call)
 *   QTOS_SHORT_STAGES=1 | 0  stage boundaries by dynamic programming: 1 = for every front size even at 2 % more stages, 0 = never;
 *                            unset = for fronts above 128 slots by that rule and for smaller fronts only where a 16-slot group
 *                            comes off the front at NO extra stage (walk 128 -> 112 slots, trot 112 -> 96).
 *                            QTOS_NO_SHORT_STAGES=1 is the older spelling of 0
 *   QTOS_SPEC_PATTERN=0      qtos_plan_submit queues the first iteration only and the host reads the counts in front of every
 *                            further one (default 1: the launch pattern below; qtos_set_pattern_speculation does the same per handle)
 *   QTOS_PLACE=1..4          slot placement rule of the analysis (0 = the measured best: a group that hosts the stage's siblings)
 *   QTOS_ORDER=0 | 1 | 2     time keys of the elimination order: 0 = rounds 1 - 5 (force nodes at their node time, B-spline coefficients
 *                            in the middle of their support), 1 = round 6's order with the late force nodes (csrc/model.hpp
 *                            HostModel::order_rule: the 100-knot walk and the 200-knot transcription fit 96 slots instead of 112),
 *                            2 = rule 1 without the late force nodes (coefficients one polynomial earlier, first-knot guard).
 *                            Unset: on a reduced base rules 2 and 1 are analysed and the smaller front is kept (rule 0 loses up to
 *                            six digits of a KKT solve on short trot horizons and is not in the automatic choice there); full-base
 *                            systems keep rule 0
 *   QTOS_KRON=1              experiment builds only (128-slot fronts, k_kkt2): the range-of-motion blocks are assembled through
 *                            their Kronecker structure -- 33 sums per block and one product of static weights per entry instead
 *                            of a three-term sum per entry; plans equal to rounding (1e-8), -0.4 % per launch
 *   QTOS_SWEEP_DS=0          k_step forms the slack steps ds = Ji dx + (g - s) itself (default 1: three waves that idle in the
 *                            backward sweep of the KKT kernels form them, block by block behind the stage that solves the
 *                            block's earliest column; bit-identical plans)
 *   QTOS_SPEC_JAC=0          k_step evaluates the first trial point of the line search without its Jacobian and linearises in a
 *                            second pass (default 1: one pass behind a Newton step; bit-identical plans) */
int qtos_plan_batch_device(QtosPlanner *p, int B, const double *d_start, const double *d_goal,
                           const int *d_map_id, const double *d_warm, double *d_nodes_out,
                           int *d_status_out, int *d_iters_out, double *d_viol_out, void *stream);

/* Time-shifted warm start for a receding-horizon replan (the reference re-plans from the
 * row `lookahead` steps ahead in the plan being executed, QTOS/combiner.py:245-296, and restarts the gait
 * schedule with every plan): warm_out[b] = the previous plan nodes_prev[b] read at offset[b] + (node time) for
 * every variable of the new plan whose shifted time still lies inside the previous horizon, towr's
 * straight-line guess of the new problem (start[b] -> goal[b]) beyond it, and the new start / goal in the fixed
 * variables.  Pass warm_out as `warm` of qtos_plan_batch*.  offset in seconds (>= 0). */
int qtos_shift_warm(QtosPlanner *p, int B, const double *nodes_prev, const double *offset,
                    const double *start, const double *goal, const int *map_id, double *warm_out);
int qtos_shift_warm_device(QtosPlanner *p, int B, const double *d_nodes_prev, const double *d_offset,
                           const double *d_start, const double *d_goal, const int *d_map_id,
                           double *d_warm_out, void *stream);

/* 1 kHz sampling of solution nodes into the reference's CSV row layout (37 columns,
 * QTOS/utils.py:107-148; producer build/traj.csv).  rows_out is B x n_rows x 37, row k is at
 * local time k/hz and carries time stamp t0[b] + k/hz.  Host pointers. */
int qtos_sample_csv(QtosPlanner *p, int B, const double *nodes, const double *t0, double hz,
                    int n_rows, double *rows_out);
int qtos_sample_csv_device(QtosPlanner *p, int B, const double *d_nodes, const double *d_t0,
                           double hz, int n_rows, double *d_rows_out, void *stream);

/* The plan as the text file the reference copies out of its container (`docker cp <id>:.../build/traj.csv ./data/traj/towr.csv`,
 * scripts/main.py:90-92; consumers scripts/run.py:129-137, QTOS/combiner.py:263-274): rows is n_rows x 37 (one plan of
 * qtos_sample_csv), every number printed as the solver's C++ stream prints it (default precision 6 = printf "%g"), comma
 * separated, no header.  Host only, needs no planner and no GPU.  n_threads <= 0: chosen from n_rows (at most 8).
 * Returns 0, -1 bad arguments, -2 the file cannot be opened, -3 a short write. */
int qtos_write_csv(const char *path, const double *rows, int n_rows, int n_threads);

int qtos_plan_submit(QtosPlanner *p, int B, const double *d_start, const double *d_goal,
                     const int *d_map_id, const double *d_warm, double *d_nodes_out,
                     int *d_status_out, int *d_iters_out, double *d_viol_out, void *stream);
int qtos_plan_poll(QtosPlanner *p, int *done);
int qtos_plan_wait(QtosPlanner *p);
/* Rounds 3 - 5: upper limit of the iterations qtos_plan_submit queues "blind" -- BOTH solve kernels per iteration, as many
 * iterations as the previous call needed (measured slower than the informed loop: a blind iteration pays for launches
 * without work).  Default 1 = off; a limit above 1 replaces the launch pattern for this handle. */
int qtos_set_speculation(QtosPlanner *p, int max_blind_iterations);
/* The launch pattern of qtos_plan_submit (above) on / off for this handle (default on; off also forgets what was learnt:
 * the next call queues its first iteration and the host reads the counts in front of every further launch, as in rounds 1 - 5). */
int qtos_set_pattern_speculation(QtosPlanner *p, int on);
/* The environment switches this handle runs with, as text ("QTOS_KKT=0 QTOS_LANES=1 ..."): at most n - 1 characters and a
 * terminating zero into buf; returns the length of the full text.  Read once by qtos_planner_create (csrc/env.hpp). */
int qtos_env(const QtosPlanner *p, char *buf, int n);

/* Seconds spent in the KKT kernels / all kernels during the last qtos_plan_batch* call, from HIP
 * events on the launch stream (valid after the stream has been synchronised), and the number of
 * KKT launches.  Used by bench.py for the roofline figure. */
int qtos_last_timing(QtosPlanner *p, double *kkt_seconds, int *kkt_launches, double *total_seconds,
                     int *iterations);
/* Where the time of the last call went (same events, lane 0), seconds: out[0] first kernel -> end of the call, out[1] initial
 * guess, out[2] solve kernels, out[3] line-search / linearisation kernels with their counts, out[4] GAPS (a slot's last event
 * -> the next slot's first: the host reading the counts and launching; zero between slots queued at submit time), then counts:
 * out[5] launch slots with work, out[6] slots queued at submit time, out[7] launches that waited for the host, out[8] calls of
 * the handle that followed a launch pattern so far, out[9] of those, calls in which a problem sat a slot out.  n_out >= 10; with
 * n_out >= 14 also out[10] seconds / out[11] launches of the factorising kernel and out[12] / out[13] of k_chord (what
 * qtos_last_timing and qtos_last_timing_chord report: one call instead of three inside a timed loop). */
int qtos_last_timing_detail(QtosPlanner *p, double *out, int n_out);
/* The HIP events the three timing entry points read are recorded around every solve kernel and behind every launch slot of a
 * call (default on: bench.py's roofline figure is measured from them).  on = 0 keeps the call's first and last event only -- a
 * caller that does not read kernel times saves the event packets between its kernels; the timing entry points then return -1. */
int qtos_set_kernel_events(QtosPlanner *p, int on);
/* The same for the chord-step launches (k_chord, QtosParams.chord_tol) of the last call. */
int qtos_last_timing_chord(QtosPlanner *p, double *chord_seconds, int *chord_launches);
/* Running totals over all qtos_plan_batch* calls of the handle since the last reset: problems returned with
 * status 0 and Newton iterations spent, tallied on the device at the end of every call (no read-back, no extra
 * work for the caller between batches).  Synchronises the stream of the last call.  reset != 0 clears them. */
int qtos_plan_totals(QtosPlanner *p, long long *converged, long long *iterations, int reset);

/* ---- introspection for the parity tests (host pointers) ------------------------------------ */
/* constraint values (B x n_cons) and, if J_out != NULL, the dense Jacobian (B x n_cons x n_vars,
 * columns of fixed variables zero, dropped rows zero) at the given nodes */
int qtos_debug_eval(QtosPlanner *p, int B, const double *start, const double *goal,
                    const int *map_id, const double *nodes, double *g_out, double *J_out);
/* one condensed KKT solve per problem with caller-supplied barrier weights:
 *   [delta I + Ji' diag(sig) Ji, Je'; Je, -eps I] [dx; y] = [-Ji' w; -g_e]
 * sig, w: B x n_cons (entries of inequality rows are used); dx_out: B x n_vars */
int qtos_debug_newton(QtosPlanner *p, int B, const double *start, const double *goal,
                      const int *map_id, const double *nodes, const double *sig, const double *w,
                      double *dx_out);
/* the same system once more through the chord-step kernel: the factorisation the preceding qtos_debug_newton
 * call left on the device + the right-hand side in elimination order (parity of k_chord with k_kkt2) */
int qtos_debug_chord(QtosPlanner *p, int B, double *dx_out);
/* QtosParams.reduce_base: a solve that is given nodes (`warm`) starts from their projection onto the space of the
 * B-spline coefficients (nodes of a C2 spline stay what they are; the reference's plans, whose acceleration continuity
 * holds to their CSV precision, move by that much).  This is that projection (host pointers, B x n_vars); without
 * reduce_base a copy. */
int qtos_project_nodes(QtosPlanner *p, int B, const double *nodes, double *nodes_out);
/* diagnostics (scratch/reduced_base_sweep.py, host-side emulations of the chain): the stage stream of problem b as the last
 * linearisation left it (qtos_debug_stream_len doubles), and the vector behind the last qtos_debug_residual call, by
 * POSITION of the elimination order: n_stages * pivots doubles (QtosDims) -- the positions count the dummy pivots that
 * fill short stages, QtosDims.n_unknowns does not */
int qtos_debug_stream_len(const QtosPlanner *p);
int qtos_debug_read_stream(QtosPlanner *p, int b, double *out);
int qtos_debug_read_rhs(QtosPlanner *p, int b, double *out);
/* a-posteriori residual of the system the preceding qtos_debug_newton call solved: res_rel_out[b] = max |b - K x| /
 * max |b|, K applied from the problem's stream without the factorisation (k_residual).  refine != 0: first one step of
 * iterative refinement through the stored factorisation (r = b - K x, K e = r by k_chord, x += e); dx_out (B x n_vars,
 * may be NULL): the (refined) solution.  SURVEY.md section 7-5: accuracy of the KKT solve vs a CPU factorisation. */
int qtos_debug_residual(QtosPlanner *p, int B, int refine, double *dx_out, double *res_rel_out);
/* working-set description: row_kind[n_cons] (0 dropped, 1 equality, 2 inequality), var_free[n_vars] (0/1), and the
 * elimination order BY POSITION: order[n_stages * pivots] (QtosDims) = var index, n_vars + row for a multiplier, or -1
 * for a dummy pivot (short stages, the fill of the last stage, the all-dummy stage pair mode may append): size the
 * buffer by n_stages * pivots, NOT by n_unknowns */
int qtos_debug_structure(const QtosPlanner *p, int *row_kind, int *var_free, int *order);
/* factor panels of problem b after the last KKT solve (n_stages x (front + 1) x 16 doubles: per stage
 * w = L^-T D^-1 y_F (16) then V = Y D^-1 L^-1 by front slot, column c of a row stored at
 * 4 (c & 3) + (c >> 2)) and the pivot slots (n_stages x 16) */
int qtos_debug_factor(QtosPlanner *p, int b, double *panel_out, int *piv_slot_out);
/* the starting point a solve without `warm` would use (straight-line guess or table guess), B x n_vars */
int qtos_debug_initial_guess(QtosPlanner *p, int B, const double *start, const double *goal,
                             const int *map_id, double *nodes_out);
/* per-iteration trace of the last plan call for problem b: rows of (viol, theta, alpha, mu),
 * at most max_iter rows; returns the number of rows */
int qtos_debug_trace(QtosPlanner *p, int b, double *trace_out);
/* What this build of the library contains: bit 0 = the kernels that were built, measured and lost (k_kkt3 MODE 0, k_kkt4,
 * the Kronecker assembly: -DQTOS_EXPERIMENTS, scratch/build.sh), bit 1 = per-wave cycle stamps (-DQTOS_STAMPS),
 * bit 2 = a development build with the benchmark's fronts only (-DQTOS_DEV_F128).  The product library returns 0. */
int qtos_build_flags(void);
/* The factor + solve kernel qtos_planner_create selected for this planner, e.g. "k_kkt2<128>", "k_kkt3<112, 1>",
 * "k_kkt5<128>" (the name rocprofv3 lists it under, without the namespace and the trailing template defaults): at most
 * n - 1 characters and a terminating zero into buf; returns the length of the full name.
 * Selection (environment variable QTOS_KKT, read at creation): unset = k_kkt3 MODE 1 for fronts of at most 112 slots,
 * k_kkt2 above; 2 = k_kkt2; 4 = k_kkt3 MODE 1 (fronts up to 128); 6 = k_kkt5 -- two 16-pivot stages per set of barriers,
 * pair-mode analysis (fronts 96 .. 144 without continuation records; otherwise the default).  Every choice solves the same
 * KKT systems; plans of different kernels differ by rounding (2e-8 on the walk, up to 5e-6 on the trot). */
int qtos_kkt_kernel(const QtosPlanner *p, char *buf, int n);

#ifdef __cplusplus
}
#endif
#endif
