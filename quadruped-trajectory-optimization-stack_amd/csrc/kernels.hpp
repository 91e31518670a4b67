// kernels.hpp -- gfx950 device code of the batched planner.  One workgroup per planning problem.
//
//   k_start : initial guess (or warm start), constraint values, slack / barrier initialisation,
//             first linearisation (per-instance dense Jacobian blocks G, barrier weights)
//   k_kkt2 / k_chord (kkt2.hpp): fused front assembly + block LDL^T (Schur-complement chain, 16 pivots per
//             stage, assembled entries in LDS cells, Schur updates in MFMA accumulator registers) +
//             forward/backward substitution  -> Newton step dx; a solve with the stored factorisation
//   k_step  : slack/dual steps, fraction-to-the-boundary, backtracking on the l1 infeasibility,
//             state update, convergence test, next linearisation
//   k_sample: 1 kHz spline sampling into the 37-column CSV row layout
//
// Formulas: towr v1.4 (the reference's solver is a fork of it, Dockerfile:45):
//   euler_converter.cc (R, M, Mdot), single_rigid_body_dynamics.cc (Newton-Euler violation),
//   range_of_motion_constraint.cc, terrain_constraint.cc, force_constraint.cc, height_map.cc.
#pragma once
#include <hip/hip_runtime.h>

#include "model.hpp"
#include "symbolic.hpp"

namespace qtos {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i4_t __attribute__((ext_vector_type(4)));

// a terrain row for the evaluation kernels: TerrInst with everything that depends on the symbolic analysis resolved on the
// host -- stream positions of its three Jacobian entries (equality block of a stance row: Symbolic::eq_pos; inequality
// block: its offset; -1 = not in the KKT system) and of the pivot diagonals of the foot node's x, y (two-phase solve, -1 = none)
struct TerrDev {
  int vx, vy, vz, row;
  int px, py, pz;
  int d0, d1;
  int pad;
  double ex, ey;   // static part of those diagonals beyond delta_x (reduced swings: the proximal term of the replaced mid nodes)
};

struct DevPlan {
  int n_vars, n_cons, n_stages, front;
  // solver variables (QtosParams.reduce_base): ids < n_vars = the model's variables, n_vars .. n_sol - 1 = B-spline coefficients
  // that replace the base node values inside the KKT solve; the step in node space is recovered as dx[rec_var[i]] =
  // sum_a rec_w[4i + a] * dx[rec_col[4i + a]] (k_recover_dx).  Without reduce_base n_sol = n_vars, n_rec = 0.
  int n_sol, n_rec;
  const int *rec_var, *rec_col;
  const double *rec_w;
  // projection of a warm start onto the space of the coefficients (model.hpp: pc_*, pz_*); n_coef = 0 without reduce_base
  int n_coef, n_pz;
  // reduced swings (QtosParams.reduce_swing): a starting point's mid-node x, y, v_x, v_y are placed on the swing rule:
  // x[psw_var[i]] = psw_w[2 i] x[psw_src[2 i]] + psw_w[2 i + 1] x[psw_src[2 i + 1]]; n_psw = 0 without
  int n_psw;
  const int *psw_var, *psw_src;
  const double *psw_w;
  const int *pc_var, *pz_var, *pz_col;
  const double *pc_w, *pz_w;
  int n_dyn, n_rom, n_terr, n_force, n_lin, n_blocks;
  const DynInst *dyn;
  const RomInst *rom;
  // spline inputs of the dynamics knots / range-of-motion instances, one record per (instance, input vector, component) in the
  // order of the pre-pass: the four variables and the four Hermite weights of VecIn, flat (vec_prepass)
  const i4_t *pre_dyn_var, *pre_rom_var;
  const d2_t *pre_dyn_wa, *pre_dyn_wb, *pre_rom_wa, *pre_rom_wb;
  const TerrDev *terr;         // (TerrInst with its stream positions resolved on the host)
  const ForceInst *force;
  const LinRow *lin;
  // iterate-dependent Jacobian entries of the dynamics / range-of-motion columns as linear forms over
  // the local Jacobians in LDS, in stream order (model.hpp: LinTerm1 / LinTerm3)
  const PackedTerm1 *dyn_t1, *rom_t1;
  const PackedTerm3 *dyn_t3;
  const double *lin_coef;      // the distinct coefficients of the three lists
  int n_lin_coef, coef_in_lds; // the evaluation kernels copy the table to LDS when it fits next to their scratch
  const int *dyn_t1_off, *dyn_t3_off;   // first entry of every knot chunk (+ end)
  int n_rom_t1;
  int rom_chunk;               // range-of-motion instances evaluated per pass (LDS scratch bound)
  const int *rom_t1_off;       // first entry of every range-of-motion chunk (+ end)
  int dyn_chunk;               // dynamics knots evaluated per pass of eval_all (LDS scratch bound)
  // optional table of nominal plans (rest start at the origin, goals on a dx x dy grid) for the initial
  // guess: table[(j * tab_ndx + i) * n_vars + v]; null = towr's straight-line guess
  const double *table, *tab_dx, *tab_dy;
  int tab_ndx, tab_ndy;
  int off_lin, off_ang, off_eem[NEE];   // first variable of the base / foot motion node sets
  const int *cont;             // continuation records of heavy stages: {srec offset, ints, stream offset, doubles} each
  int n_cont;                  // how many there are in the whole plan (0 for the standard transcriptions)
  int spec_jac;                // k_step: Jacobian evaluated with the first trial point of the line search (QTOS_SPEC_JAC, default 1)
  int hold_from;               // two-phase solve: stance footholds are held once an iterate >= hold_from has violation <= hold_tol (0: never)
  double hold_weight, hold_tol;
  const unsigned *amask;       // n_stages x 8: rows of the factor panel that are stored / read back (Symbolic::amask)
  const unsigned *amask2;      // the same without the slots of the next stage's pivots (Symbolic::amask2)
  const int *nxt_pack;         // n_stages x 4: slots of the next stage's pivots in this stage's panel, 255 = none (Symbolic::nxt_pack)
  const unsigned short *ctab;  // n_stages x (front/16) x 64 x 4: cell of every entry of a stage's pivot columns (Symbolic::ctab)
  int n_cells;                 // cells of the assembled entries: [0] zero, [1 + slot] right-hand side, then the entries
  // chord step (QtosParams.chord_tol): right-hand side of the KKT system in elimination order, formed by k_step:
  // unknown p is a multiplier (rhs_ptr[p+1] - rhs_ptr[p] == 1, rhs_gpos < 0: rhs = -g[rhs_row]) or a variable
  // (rhs = -sum G[rhs_gpos] * w[rhs_row] over the inequality rows that contain it)
  double chord_tol, chord_shrink;
  int chord_max;               // chord steps in a row with one factorisation (QtosParams.chord_max)
  int n_unknowns;
  const int *rhs_ptr, *rhs_gpos, *rhs_row;
  int n_rhs_ent, rhs_chunk;    // entries of the three lists; entries per pass through the LDS scratch of k_step (a multiple of ET)
  const int *kx_ptr, *kx_col, *kx_pos;   // equality part of K by unknown position (Symbolic::kx_ptr; k_residual)
  const int *rtab;             // n_stages x 16: cell of the assembled right-hand side of every pivot (k_kkt2, Symbolic::rtab)
  const Block *blocks;
  const int *block_cols;
  const IqRow *iq_rows;   // rows of the inequality blocks (stream offsets)
  int n_iq_rows;
  // the same rows as rounds of 16 for the helper waves of the backward sweep (QTOS_SWEEP_DS, default on): round i runs in
  // step i of the chain; k_step then reads ds instead of forming it
  const SwTask *sw_tasks;
  const int *sw_cpos, *sw_c16;   // sw_c16: per block 20 ints = for lane q of a row's quad the positions of entries q, q + 4, ... (8 x 16 bit), then of the up to three entries behind the whole groups
  int sw_steps, sw_on;
  // compact row lists of the working set (the row loops of k_step run over these, branch-free):
  // inequality rows with their bounds, equality rows
  const int *iq_idx, *eq_idx;
  const double *iq_lo, *iq_hi;
  int n_iq, n_eqw;
  const double *g_static;
  const int *piv_slot, *piv_unknown;
  const double *piv_diag;
  const StageDesc *stages;
  const EqEntry *eq_entries;
  const EqRhs *eq_rhs;
  const IqBlock *iq_blocks;
  const short *iq_slots;
  int max_stage_g;
  const int *srec, *srec_off, *pack_src, *drec_off;  // packed records, one per stage or -- Symbolic::pair_mode -- per pair of stages (symbolic.hpp)
  const int *diag_pos;         // per position: stream position of its pivot diagonal (Symbolic::diag_pos)
  const unsigned *pair_groups; // per pair: occupied 16-slot groups (Symbolic::pair_groups, k_kkt5)
  const int *eq_pos, *rhs_pos, *sig_pos, *w_pos;     // direct-write maps into the stream
  int max_srec, max_drec, stream_len;
  const double *con_lo, *con_hi;
  const int *row_kind;
  const InitDesc *init;
  const double *var_time;      // node time of every variable (k_shift_warm)
  double mass, gravity, Ib[9], mu_fric, f_max, T;
  double nominal[NEE][3];
  double tol, mu_init, mu_min, delta_x, eps_dual, slack_push, warm_slack_push;
  int mu_superlinear;   // QtosParams.mu_superlinear: Ipopt's monotone update of the barrier parameter
  int max_iter, stall_iters;
  double stall_alpha;          // QtosParams.stall_alpha
  const double *height;
  int n_maps, hnx, hny, terrain_mode;
  double hcell, hx0, hy0;
  long long g_doubles, panel_stride;  // per problem
  int kron_lds_off;            // k_kkt2<F, CONT, true>: byte offset of the Kronecker scratch (33 doubles per block of a record) in its LDS
};

struct DevWork {
  const double *start, *goal, *warm;
  const int *map_id;
  double *x, *g, *gt, *s, *zl, *zu, *ds, *dzl, *dzu, *sig, *w, *G, *panel, *dx, *stream;
  double *mu, *viol, *trace;
  double *best_viol, *xbest;   // stall detection: lowest violation seen and the iterate that had it
  int *held;                   // two-phase solve: 1 once the problem's footholds are held
  int *best_it;
  int *status, *iters, *done, *n_active;   // n_active[0]: unfinished problems, [1]: of those, flagged for a chord step, [2]: (1 << 20) - the earliest launch slot a
                                           // problem sat out (k_step; 0 = none), [3]: launch slots of the call in which some problem took a step
  int *chord;                  // per problem: the next KKT solve reuses the stored factorisation (k_chord)
  int *chord_run;              // per problem: chord steps taken with the stored factorisation
  int *jam;                    // per problem: steps in a row shorter than stall_alpha
  double *rhs;                 // per problem n_unknowns: right-hand side for that solve
  double *minv;                // per problem n_stages x 256: inverse of every pivot block (written by k_kkt2)
  double *sol;                 // per problem n_stages x 16: the solution of the last KKT solve by unknown position (variables AND multipliers)
  double *sol0, *dx0, *ur;     // iterative refinement (k_residual / k_refine_add): first solution, and sig * (Ji dx) by constraint row
  // where a finished problem leaves its result (the caller's buffers of this call; status / iters / viol may be null) and the
  // handle's running totals {converged problems, iterations}: written by the kernel that finishes the problem (export_problem)
  double *nodes_out, *viol_out;
  int *status_out, *iters_out;
  unsigned long long *totals;
};

// ---- tiny forward-mode dual (one tangent) for the rotation-dependent Jacobians ---------------
struct D1 {
  double v, d;
};
__device__ inline D1 operator+(D1 a, D1 b) { return {a.v + b.v, a.d + b.d}; }
__device__ inline D1 operator-(D1 a, D1 b) { return {a.v - b.v, a.d - b.d}; }
__device__ inline D1 operator-(D1 a) { return {-a.v, -a.d}; }
__device__ inline D1 operator*(D1 a, D1 b) { return {a.v * b.v, a.v * b.d + a.d * b.v}; }
__device__ inline D1 operator*(double a, D1 b) { return {a * b.v, a * b.d}; }
__device__ inline D1 operator*(D1 a, double b) { return {a.v * b, a.d * b}; }
__device__ inline void sincos_t(double x, double &s, double &c) { sincos(x, &s, &c); }
__device__ inline void sincos_t(D1 x, D1 &s, D1 &c) {
  double sv, cv;
  sincos(x.v, &sv, &cv);
  s = {sv, cv * x.d};
  c = {cv, -sv * x.d};
}
// sines / cosines of the three Euler angles of one instance, computed once and shared by the value
// pass and all forward-mode passes (sincos is by far the most expensive operation in them)
struct Trig {
  double s[3], c[3];
};
__device__ inline Trig trig_of(const double th[3]) {
  Trig t;
#pragma unroll
  for (int i = 0; i < 3; ++i) sincos(th[i], &t.s[i], &t.c[i]);
  return t;
}
__device__ inline void sincos_c(double, double sv, double cv, double &s, double &c) { s = sv; c = cv; }
__device__ inline void sincos_c(D1 x, double sv, double cv, D1 &s, D1 &c) {
  s = {sv, cv * x.d};
  c = {cv, -sv * x.d};
}
__device__ inline double mk(double, double v) { return v; }
__device__ inline D1 mk(D1, double v) { return {v, 0.0}; }

// R = Rz(yaw) Ry(pitch) Rx(roll), th = (roll, pitch, yaw)
template <class T>
__device__ inline void rotation(const T th[3], const Trig &tg, T R[9]) {
  T sx, cx, sy, cy, sz, cz;
  sincos_c(th[0], tg.s[0], tg.c[0], sx, cx);
  sincos_c(th[1], tg.s[1], tg.c[1], sy, cy);
  sincos_c(th[2], tg.s[2], tg.c[2], sz, cz);
  R[0] = cy * cz; R[1] = cz * sx * sy - cx * sz; R[2] = sx * sz + cx * cz * sy;
  R[3] = cy * sz; R[4] = cx * cz + sx * sy * sz; R[5] = cx * sy * sz - cz * sx;
  R[6] = -sy;     R[7] = cy * sx;                R[8] = cx * cy;
}

// I_w wd + w x (I_w w) with I_w = R Ib R^T, w = M(th) thd, wd = Mdot thd + M thdd
template <class T>
__device__ inline void dyn_angular(const double *Ib, const T th[3], const T thd[3], const T thdd[3],
                                   const Trig &tg, T out[3]) {
  T sy, cy, sz, cz;
  sincos_c(th[1], tg.s[1], tg.c[1], sy, cy);
  sincos_c(th[2], tg.s[2], tg.c[2], sz, cz);
  const T yd = thd[1], zd = thd[2];
  // w = M thd
  T w[3], wd[3];
  w[0] = cy * cz * thd[0] - sz * thd[1];
  w[1] = cy * sz * thd[0] + cz * thd[1];
  w[2] = thd[2] - sy * thd[0];
  // wd = Mdot thd + M thdd
  T m00 = -(cz * sy * yd) - cy * sz * zd, m01 = -(cz * zd);
  T m10 = cy * cz * zd - sy * sz * yd, m11 = -(sz * zd);
  T m20 = -(cy * yd);
  wd[0] = m00 * thd[0] + m01 * thd[1] + cy * cz * thdd[0] - sz * thdd[1];
  wd[1] = m10 * thd[0] + m11 * thd[1] + cy * sz * thdd[0] + cz * thdd[1];
  wd[2] = m20 * thd[0] + thdd[2] - sy * thdd[0];
  T R[9];
  rotation(th, tg, R);
  // body-frame vectors u = R^T w, ud = R^T wd ; I_w v = R (Ib (R^T v))
  T u[3], ud[3];
  for (int i = 0; i < 3; ++i) {
    u[i] = R[i] * w[0] + R[3 + i] * w[1] + R[6 + i] * w[2];
    ud[i] = R[i] * wd[0] + R[3 + i] * wd[1] + R[6 + i] * wd[2];
  }
  T Iu[3], Iud[3];
  for (int i = 0; i < 3; ++i) {
    Iu[i] = Ib[3 * i] * u[0] + Ib[3 * i + 1] * u[1] + Ib[3 * i + 2] * u[2];
    Iud[i] = Ib[3 * i] * ud[0] + Ib[3 * i + 1] * ud[1] + Ib[3 * i + 2] * ud[2];
  }
  T Iww[3], Iwd[3];
  for (int i = 0; i < 3; ++i) {
    Iww[i] = R[3 * i] * Iu[0] + R[3 * i + 1] * Iu[1] + R[3 * i + 2] * Iu[2];
    Iwd[i] = R[3 * i] * Iud[0] + R[3 * i + 1] * Iud[1] + R[3 * i + 2] * Iud[2];
  }
  out[0] = Iwd[0] + w[1] * Iww[2] - w[2] * Iww[1];
  out[1] = Iwd[1] + w[2] * Iww[0] - w[0] * Iww[2];
  out[2] = Iwd[2] + w[0] * Iww[1] - w[1] * Iww[0];
}

__device__ inline void vec_eval(const VecIn &in, const double *x, double out[3]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    double acc = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int v = in.var[3 * a + d];
      if (v >= 0) acc += in.w[a] * x[v];
    }
    out[d] = acc;
  }
}

// Pre-pass of the spline inputs: one work item per (instance, input vector, component); the items' variables and weights come
// from flat tables in item order (DevPlan::pre_*: three 16-byte loads per item, consecutive items consecutive -- the VecIn
// records inside the instance descriptors cost five times the cache-line visits).  Same summation order as vec_eval.
__device__ inline void vec_prepass(const i4_t *__restrict__ V, const d2_t *__restrict__ Wa, const d2_t *__restrict__ Wb, int total,
                                   const double *x, double *out) {
  const int nt = blockDim.x;
#ifndef PREPASS_UNROLL
#define PREPASS_UNROLL 4
#endif
  constexpr int UN = PREPASS_UNROLL;   // work items per thread and round: all table reads of a round are in flight together (two rounds
                          // for the 100 knots of the benchmark -- 4 k items on 512 threads; measured: 2: +0 %, 4: best, 6: +2 %, 8: +0.8 %
                          // of the evaluation kernels' time)
  for (int base = threadIdx.x; base < total; base += UN * nt) {
    i4_t var[UN];
    d2_t wa[UN], wb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = min(base + u * nt, total - 1);
      var[u] = V[idx]; wa[u] = Wa[idx]; wb[u] = Wb[idx];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const double w[4] = {wa[u][0], wa[u][1], wb[u][0], wb[u][1]};
      double acc = 0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (var[u][a] >= 0) acc += w[a] * x[var[u][a]];
      if (base + u * nt < total) out[base + u * nt] = acc;
    }
  }
}

struct Terr {
  double h, hx, hy, hxy;
};
// bilinear height, clamped at the border (zero slope outside the map)
__device__ inline Terr terrain_at(const DevPlan &P, int map, double x, double y) {
  Terr t = {0, 0, 0, 0};
  if (!P.height || P.n_maps <= 0) return t;
  map = min(max(map, 0), P.n_maps - 1);   // a stale / out-of-range map id must not read outside the maps
  const double *H = P.height + (size_t)map * P.hnx * P.hny;
  double fx = (x - P.hx0) / P.hcell, fy = (y - P.hy0) / P.hcell;
  const double mx = P.hnx - 1, my = P.hny - 1;
  if (P.terrain_mode == 1) {  // nearest cell: ledges stay flat, steps are jumps
    int jx = (int)floor(fx + 0.5), jy = (int)floor(fy + 0.5);
    jx = min(max(jx, 0), P.hnx - 1);
    jy = min(max(jy, 0), P.hny - 1);
    t.h = H[jx * P.hny + jy];
    return t;
  }
  bool cx = false, cy = false;
  if (fx <= 0) { fx = 0; cx = true; }
  if (fx >= mx) { fx = mx; cx = true; }
  if (fy <= 0) { fy = 0; cy = true; }
  if (fy >= my) { fy = my; cy = true; }
  int ix = (int)floor(fx), iy = (int)floor(fy);
  ix = min(ix, P.hnx - 2); iy = min(iy, P.hny - 2);
  ix = max(ix, 0); iy = max(iy, 0);
  const double u = fx - ix, v = fy - iy;
  const int ix1 = P.hnx > 1 ? ix + 1 : ix, iy1 = P.hny > 1 ? iy + 1 : iy;
  const double h00 = H[ix * P.hny + iy], h10 = H[ix1 * P.hny + iy], h01 = H[ix * P.hny + iy1],
               h11 = H[ix1 * P.hny + iy1];
  t.h = h00 * (1 - u) * (1 - v) + h10 * u * (1 - v) + h01 * (1 - u) * v + h11 * u * v;
  const double c = P.hcell;
  t.hx = cx ? 0 : ((h10 - h00) * (1 - v) + (h11 - h01) * v) / c;
  t.hy = cy ? 0 : ((h01 - h00) * (1 - u) + (h11 - h10) * u) / c;
  t.hxy = (cx || cy) ? 0 : (h11 - h10 - h01 + h00) / (c * c);
  return t;
}
// normalised normal / tangent1 / tangent2 and their x, y derivatives (height_map.cc)
__device__ inline void terrain_basis(const Terr &t, int which, double b[3], double bx[3], double by[3]) {
  double v[3], vx[3] = {0, 0, 0}, vy[3] = {0, 0, 0};
  if (which == 0) { v[0] = -t.hx; v[1] = -t.hy; v[2] = 1; vx[1] = -t.hxy; vy[0] = -t.hxy; }
  else if (which == 1) { v[0] = 1; v[1] = 0; v[2] = t.hx; vy[2] = t.hxy; }
  else { v[0] = 0; v[1] = 1; v[2] = t.hy; vx[2] = t.hxy; }
  const double nn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  for (int i = 0; i < 3; ++i) b[i] = v[i] / nn;
  const double px = b[0] * vx[0] + b[1] * vx[1] + b[2] * vx[2];
  const double py = b[0] * vy[0] + b[1] * vy[1] + b[2] * vy[2];
  for (int i = 0; i < 3; ++i) { bx[i] = (vx[i] - b[i] * px) / nn; by[i] = (vy[i] - b[i] * py) / nn; }
}

// ---- per-instance evaluation ------------------------------------------------------------------
// Dynamics / range-of-motion blocks are linearised in two phases so that every Jacobian entry is
// written exactly once and all threads of the workgroup share the work:
//   phase A (thread per instance): constraint values + the small local Jacobians -> LDS
//   phase B (thread per block column): the 6 (3) entries of that column from the local Jacobians
//                                      and the column's combined Hermite weights (LinTerm1 / LinTerm3)

// Newton-Euler violation of one dynamics knot (values; with JAC also the force / lever-arm part of
// the local Jacobian data).  The nine forward-mode passes for the Euler-angle columns are separate
// work items (eval_dyn_pass) so that ten threads share one knot.
template <bool JAC>
__device__ inline void eval_dyn(const DevPlan &P, const DynInst &I, const double *vin, double *g, double *loc,
                                const double th[3], const double thd[3], const double thdd[3], const Trig &tg) {
  // vin: the 13 spline inputs of the knot evaluated by the pre-pass: r a th thd thdd p[0..3] f[0..3]
  const double r[3] = {vin[0], vin[1], vin[2]}, a[3] = {vin[3], vin[4], vin[5]};
  double ga[3], gl[3];
  dyn_angular<double>(P.Ib, th, thd, thdd, tg, ga);
  gl[0] = P.mass * a[0]; gl[1] = P.mass * a[1]; gl[2] = P.mass * a[2] + P.mass * P.gravity;
  const bool jac = JAC && I.in_kkt;
  double sf[3] = {0, 0, 0};
#pragma unroll
  for (int e = 0; e < NEE; ++e) {
    const double *pe = vin + 15 + 3 * e, *f = vin + 27 + 3 * e;
    const double d[3] = {r[0] - pe[0], r[1] - pe[1], r[2] - pe[2]};
    // tau_sum += f x (r - p)
    ga[0] -= f[1] * d[2] - f[2] * d[1];
    ga[1] -= f[2] * d[0] - f[0] * d[2];
    ga[2] -= f[0] * d[1] - f[1] * d[0];
    gl[0] -= f[0]; gl[1] -= f[1]; gl[2] -= f[2];
    if (jac) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { sf[i] += f[i]; loc[30 + 3 * e + i] = f[i]; loc[42 + 3 * e + i] = d[i]; }
    }
  }
  if (jac) { loc[27] = sf[0]; loc[28] = sf[1]; loc[29] = sf[2]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { g[I.row0 + i] = ga[i]; g[I.row0 + 3 + i] = gl[i]; }
}
// three forward-mode passes: d(angular rows) / d(theta (WHAT = 0), theta-dot (1), theta-ddot (2)).
// WHAT is a compile-time constant so that the zero tangents fold away.
template <int WHAT>
__device__ inline void eval_dyn_pass(const DevPlan &P, const DynInst &I, double *loc, const double th[3],
                                     const double thd[3], const double thdd[3], const Trig &tg) {
  if (!I.in_kkt) return;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    D1 t0[3], t1[3], t2[3], o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      t0[i] = {th[i], (WHAT == 0 && i == j) ? 1.0 : 0.0};
      t1[i] = {thd[i], (WHAT == 1 && i == j) ? 1.0 : 0.0};
      t2[i] = {thdd[i], (WHAT == 2 && i == j) ? 1.0 : 0.0};
    }
    dyn_angular<D1>(P.Ib, t0, t1, t2, tg, o);
#pragma unroll
    for (int i = 0; i < 3; ++i) loc[9 * WHAT + 3 * i + j] = o[i].d;
  }
}

#ifndef TERM1_UNROLL
#define TERM1_UNROLL 24
#endif
#ifndef TERM3_UNROLL
#define TERM3_UNROLL 8
#endif
// G[pos] = a * loc[off] for the entries [i0, i1) of a PackedTerm1 list, TERM1_UNROLL per thread and round (all
// descriptor reads of a round in flight together); consecutive threads write consecutive positions.  coef = the table
// of coefficients in LDS.
__device__ inline void write_terms1(const PackedTerm1 *T, int i0, int i1, const double *loc, const double *coef, double *G) {
  const int nt = blockDim.x;
  constexpr int UN = TERM1_UNROLL;   // descriptor reads in flight per thread: the lists are read at memory latency
  for (int i = i0 + threadIdx.x; i < i1; i += UN * nt) {
    PackedTerm1 t[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) t[u] = T[min(i + u * nt, i1 - 1)];
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (i + u * nt < i1) G[t[u].pos] = coef[t[u].ai] * loc[t[u].off];
  }
}
__device__ inline void write_terms3(const PackedTerm3 *T, int i0, int i1, const double *loc, const double *coef, double *G) {
  const int nt = blockDim.x;
  constexpr int UN = TERM3_UNROLL;
  for (int i = i0 + threadIdx.x; i < i1; i += UN * nt) {
    PackedTerm3 t[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) t[u] = T[min(i + u * nt, i1 - 1)];
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (i + u * nt < i1)
        G[t[u].pos] = loc[t[u].off[0]] * coef[t[u].ai[0]] + loc[t[u].off[1]] * coef[t[u].ai[1]] + loc[t[u].off[2]] * coef[t[u].ai[2]];
  }
}

template <bool JAC>
__device__ inline void eval_rom(const DevPlan &P, const RomInst &I, const double *vin, double *g, double *loc) {
  // vin: r th p evaluated by the pre-pass
  const double *r = vin, *th = vin + 3, *pe = vin + 6;
  const double d[3] = {pe[0] - r[0], pe[1] - r[1], pe[2] - r[2]};
  double R[9];
  const Trig tg = trig_of(th);
  rotation<double>(th, tg, R);
#pragma unroll
  for (int i = 0; i < 3; ++i) g[I.row0 + i] = R[i] * d[0] + R[3 + i] * d[1] + R[6 + i] * d[2];
  if (JAC) {
#pragma unroll
    for (int i = 0; i < 9; ++i) loc[i] = R[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      D1 t0[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) t0[i] = {th[i], i == j ? 1.0 : 0.0};
      D1 Rd[9];
      rotation<D1>(t0, tg, Rd);
#pragma unroll
      for (int i = 0; i < 3; ++i) loc[9 + 3 * i + j] = Rd[i].d * d[0] + Rd[3 + i].d * d[1] + Rd[6 + i].d * d[2];
    }
  }
}

template <bool JAC>
__device__ inline void eval_terr(const DevPlan &P, const TerrDev &I, int map, const double *x, double *g, double *Gp, int hold = -1) {
  const Terr t = terrain_at(P, map, x[I.vx], x[I.vy]);
  g[I.row] = x[I.vz] - t.h;
  if (JAC) {
    // two-phase solve (QtosParams.hold_from): the proximal weight of a stance foothold's x, y for the
    // KKT system of this iterate -- delta_x while the feet are being placed, hold_weight afterwards
    // (hold = -1, the introspection calls: delta_x)
    const double wgt = hold > 0 ? P.hold_weight : P.delta_x;
    if (I.d0 >= 0) Gp[I.d0] = wgt + I.ex;
    if (I.d1 >= 0) Gp[I.d1] = wgt + I.ey;
    if (I.px >= 0) Gp[I.px] = -t.hx;
    if (I.py >= 0) Gp[I.py] = -t.hy;
    if (I.pz >= 0) Gp[I.pz] = 1.0;
  }
}

template <bool JAC>
__device__ inline void eval_force(const DevPlan &P, const ForceInst &I, int map, const double *x, double *g, double *Gp) {
  const double f[3] = {x[I.vf[0]], x[I.vf[1]], x[I.vf[2]]};
  const Terr t = terrain_at(P, map, x[I.vsx], x[I.vsy]);
  double b[3][3], bx[3][3], by[3][3];
  for (int w = 0; w < 3; ++w) terrain_basis(t, w, b[w], bx[w], by[w]);
  const double mu = P.mu_fric;
  const double ct[5][3] = {{1, 0, 0}, {-mu, 1, 0}, {mu, 1, 0}, {-mu, 0, 1}, {mu, 0, 1}};
  const int nc = I.ncol;
  double *G = JAC ? Gp + I.goff : nullptr;
  for (int row = 0; row < 5; ++row) {
    double v[3], vx[3], vy[3];
    for (int i = 0; i < 3; ++i) {
      v[i] = ct[row][0] * b[0][i] + ct[row][1] * b[1][i] + ct[row][2] * b[2][i];
      vx[i] = ct[row][0] * bx[0][i] + ct[row][1] * bx[1][i] + ct[row][2] * bx[2][i];
      vy[i] = ct[row][0] * by[0][i] + ct[row][1] * by[1][i] + ct[row][2] * by[2][i];
    }
    g[I.row0 + row] = f[0] * v[0] + f[1] * v[1] + f[2] * v[2];
    if (JAC) {
      for (int i = 0; i < 3; ++i)
        if (I.cf[i] >= 0) G[row * nc + I.cf[i]] = v[i];
      if (I.csx >= 0) G[row * nc + I.csx] = f[0] * vx[0] + f[1] * vx[1] + f[2] * vx[2];
      if (I.csy >= 0) G[row * nc + I.csy] = f[0] * vy[0] + f[1] * vy[1] + f[2] * vy[2];
    }
  }
}

// all constraint rows of one problem, by the whole workgroup; `loc` = LDS scratch of
// max(DYN_LOC * n_dyn, ROM_LOC * n_rom) doubles (only used when JAC)
// lds: n_vars doubles for a staged copy of the nodes (every spline evaluation then reads LDS instead
// of chasing indices through global memory), followed by the local Jacobian data of the dynamics /
// range-of-motion instances
__device__ __forceinline__ int eval_loc_offset(int n_vars) { return (n_vars + 1) & ~1; }
constexpr int DYN_VIN = 39, ROM_VIN = 9;   // pre-evaluated spline inputs per instance
// The barriers inside eval_all order LDS traffic only (the nodes, the pre-evaluated inputs, the local Jacobians); what an evaluation
// writes to memory (g, the stream) is read behind the caller's __syncthreads().  A barrier that also drains the memory counter would
// put every section behind the stores of the one before it.
__device__ __forceinline__ void wg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifndef QTOS_EVAL_LDSBAR
#define QTOS_EVAL_LDSBAR 1
#endif
#if QTOS_EVAL_LDSBAR
#define EVAL_BARRIER() wg_lds_barrier()
#else
#define EVAL_BARRIER() __syncthreads()
#endif
template <bool JAC>
__device__ inline void eval_all(const DevPlan &P, int map, const double *xg, double *g, double *G, double *lds, double *dbg = nullptr,
                                int hold = -1) {
  const int tid = threadIdx.x, nt = blockDim.x;
#ifdef QTOS_STAMPS
  unsigned long long et0 = 0; int ei = 0;
#define ESTAMP() do { __syncthreads(); if (tid == 0 && dbg) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); dbg[ei++] = (double)(t_ - et0); et0 = t_; } } while (0)
  if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(et0) :: "memory");
#else
#define ESTAMP() do {} while (0)
#endif
  double *x = lds, *loc = lds + eval_loc_offset(P.n_vars);
  double *vin = loc + max(DYN_LOC * P.dyn_chunk, ROM_LOC * P.rom_chunk);
  double *coef_lds = vin + max(DYN_VIN * P.dyn_chunk, ROM_VIN * P.rom_chunk);   // coefficients of the entry lists (JAC only)
  const double *coef = P.coef_in_lds ? coef_lds : P.lin_coef;
  if (xg)   // (null: the caller has left the nodes in lds[0 .. n_vars) already)
    for (int v = tid; v < P.n_vars; v += nt) x[v] = xg[v];
  if (JAC && P.coef_in_lds)
    for (int v = tid; v < P.n_lin_coef; v += nt) coef_lds[v] = P.lin_coef[v];
  EVAL_BARRIER();
  // the dynamics knots go through the LDS scratch in chunks of P.dyn_chunk (one chunk up to 128 knots)
  for (int c0 = 0, ch = 0; c0 < P.n_dyn; c0 += P.dyn_chunk, ++ch) {
    const int cnt = min(P.dyn_chunk, P.n_dyn - c0);
    if (c0) EVAL_BARRIER();   // the previous chunk is done with vin / loc
    vec_prepass(P.pre_dyn_var + c0 * DYN_VIN, P.pre_dyn_wa + c0 * DYN_VIN, P.pre_dyn_wb + c0 * DYN_VIN, cnt * DYN_VIN, x, vin);
    EVAL_BARRIER();
    ESTAMP();
    if (JAC) {
      // four work items per knot -- the value pass and the three groups of forward-mode passes --, a whole number of
      // waves per kind (a chunk of 128 knots is one item per thread)
      const int cpad = (cnt + 63) & ~63;
      for (int it = tid; it < 4 * cpad; it += nt) {
        const int what = it / cpad, i = it - what * cpad;
        if (i >= cnt) continue;
        double *li = loc + (size_t)i * DYN_LOC;
        const DynInst &I = P.dyn[c0 + i];
        const double *vi = vin + (size_t)i * DYN_VIN, *th = vi + 6, *thd = vi + 9, *thdd = vi + 12;
        const Trig tg = trig_of(th);
        if (what == 0) eval_dyn<true>(P, I, vi, g, li, th, thd, thdd, tg);
        else if (what == 1) eval_dyn_pass<0>(P, I, li, th, thd, thdd, tg);
        else if (what == 2) eval_dyn_pass<1>(P, I, li, th, thd, thdd, tg);
        else eval_dyn_pass<2>(P, I, li, th, thd, thdd, tg);
      }
      EVAL_BARRIER();
      ESTAMP();
      write_terms1(P.dyn_t1, P.dyn_t1_off[ch], P.dyn_t1_off[ch + 1], loc, coef, G);
      write_terms3(P.dyn_t3, P.dyn_t3_off[ch], P.dyn_t3_off[ch + 1], loc, coef, G);
      EVAL_BARRIER();
      ESTAMP();
    } else {
      for (int i = tid; i < cnt; i += nt) {
        const DynInst &I = P.dyn[c0 + i];
        const double *vi = vin + (size_t)i * DYN_VIN, *th = vi + 6, *thd = vi + 9, *thdd = vi + 12;
        eval_dyn<false>(P, I, vi, g, nullptr, th, thd, thdd, trig_of(th));
      }
      ESTAMP();
    }
  }
  for (int c0 = 0, ch = 0; c0 < P.n_rom; c0 += P.rom_chunk, ++ch) {
    const int cnt = min(P.rom_chunk, P.n_rom - c0);
    EVAL_BARRIER();   // the dynamics knots / the previous chunk are done with vin and loc
    vec_prepass(P.pre_rom_var + c0 * ROM_VIN, P.pre_rom_wa + c0 * ROM_VIN, P.pre_rom_wb + c0 * ROM_VIN, cnt * ROM_VIN, x, vin);
    EVAL_BARRIER();
    for (int i = tid; i < cnt; i += nt) eval_rom<JAC>(P, P.rom[c0 + i], vin + (size_t)i * ROM_VIN, g, JAC ? loc + (size_t)i * ROM_LOC : nullptr);
    if (!JAC) ESTAMP();
    if (JAC) {
      EVAL_BARRIER();
      ESTAMP();
      write_terms1(P.rom_t1, P.rom_t1_off[ch], P.rom_t1_off[ch + 1], loc, coef, G);
      ESTAMP();
    }
  }
  // force, terrain and constant-coefficient rows: the descriptors of a thread's items are read before the first of them is
  // used (three loops in a row are three memory round trips in a row: the stores of one may alias the loads of the
  // next); force rows from the last thread down, terrain rows below them, two linear rows per thread from the first up
  {
    const int rt = nt - 1 - tid;
    for (int base = 0; base < max(max(P.n_force + P.n_terr, 1), (P.n_lin + 1) / 2); base += nt) {
      const int fi = base + rt, ti = base + rt - P.n_force, l0 = 2 * (base + tid), l1 = l0 + 1;
      const bool hf = fi < P.n_force, ht = ti >= 0 && ti < P.n_terr, h0 = l0 < P.n_lin, h1 = l1 < P.n_lin;
      ForceInst FI;
      TerrDev TI;
      LinRow L0, L1;
      if (hf) FI = P.force[fi];
      if (ht) TI = P.terr[ti];
      if (h0) L0 = P.lin[l0];
      if (h1) L1 = P.lin[l1];
      if (hf) eval_force<JAC>(P, FI, map, x, g, G);
      if (ht) eval_terr<JAC>(P, TI, map, x, g, G, hold);
      auto lin_row = [&](const LinRow &L) __attribute__((always_inline)) {   // (the row sits in registers: constant indices only)
        double xv[8], acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = x[k < L.n ? L.var[k] : 0];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < L.n) acc += L.coef[k] * xv[k];
        g[L.row] = acc;
      };
      if (h0) lin_row(L0);
      if (h1) lin_row(L1);
    }
  }
  ESTAMP();
}

#ifndef QTOS_ET
#define QTOS_ET 512
#endif
constexpr int ET = QTOS_ET;   // threads of the evaluation kernels (k_start, k_step): one workgroup per problem
// ---- workgroup reductions (fixed tree => bitwise reproducible) --------------------------------
// The tree: a[t] = op(a[t], a[t + s]) for s = nt/2 ... 1, the result in a[0] (rounds 1 - 4 ran it level by level through LDS:
// twelve workgroup barriers per reduction, each behind the kernel's outstanding memory traffic, six reductions per k_step
// launch).  The same tree, the same bits, with ONE trip through LDS: the levels that pair waves (s >= 64) are lane-wise --
// every wave reads the ET / 64 values of its lane and combines them in the tree's order --, the levels inside a wave pair lane t
// with lane t + s: rows 0, 1 with rows 2, 3 and row 0 with row 1 by the gfx950 lane swaps, then 8, 4, 2, 1 lanes inside a row
// by DPP rotations (lane t < s reads lane t + s: all the tree needs).  Both barriers wait for LDS only.
template <int OP>
__device__ __forceinline__ double wg_op(double a, double b) { return OP == 0 ? a + b : (OP == 1 ? fmax(a, b) : fmin(a, b)); }
template <int CTRL>
__device__ __forceinline__ double wg_dpp(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false));
}
// NT = the workgroup's thread count (the evaluation kernels' ET): scratch holds NT doubles, and EVERY lane of every wave must
// be active (the lane swaps and readfirstlane below read across the whole wave) -- no calls from divergent control flow.
template <int OP, int NT = ET>  // 0 sum, 1 max, 2 min
__device__ inline double wg_reduce(double v, double *scratch) {
  static_assert(NT % 64 == 0 && (NT / 64 & (NT / 64 - 1)) == 0, "wg_reduce: a power-of-two number of full waves");
  constexpr int NW = NT / 64;
#ifdef QTOS_CHECKS   // (development builds: a kernel launched with another thread count would read stale scratch)
  if (blockDim.x != NT) __builtin_trap();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  wg_lds_barrier();             // the previous reduction's readers are done with the scratch
  scratch[tid] = v;
  wg_lds_barrier();
  double a[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) a[w] = scratch[64 * w + lane];
#pragma unroll
  for (int s = NW / 2; s > 0; s >>= 1)
#pragma unroll
    for (int w = 0; w < s; ++w) a[w] = wg_op<OP>(a[w], a[w + s]);
  double r = a[0];
  {
    int lo = __double2loint(r), hi = __double2hiint(r);
    auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);   // [0]: rows (0, 1, 0, 1), [1]: rows (2, 3, 2, 3)
    auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    r = wg_op<OP>(__hiloint2double(h[0], l[0]), __hiloint2double(h[1], l[1]));
    lo = __double2loint(r); hi = __double2hiint(r);
    auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // [0]: the even rows of the source, [1]: the odd ones
    auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    r = wg_op<OP>(__hiloint2double(h2[0], l2[0]), __hiloint2double(h2[1], l2[1]));
  }
  r = wg_op<OP>(r, wg_dpp<0x128>(r));   // row_ror:8  (lane i reads lane i - n mod 16: i + 8)
  r = wg_op<OP>(r, wg_dpp<0x12C>(r));   // row_ror:12 (i + 4)
  r = wg_op<OP>(r, wg_dpp<0x12E>(r));   // row_ror:14 (i + 2)
  r = wg_op<OP>(r, wg_dpp<0x12F>(r));   // row_ror:15 (i + 1)
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(r)), __builtin_amdgcn_readfirstlane(__double2loint(r)));
}

// barrier weights of every inequality row: sig = zl/(s-l) + zu/(u-s), w = sig (g - s) - mu/(s-l) + mu/(u-s)
__device__ inline void barrier_terms(const DevPlan &P, const double *__restrict__ g, const double *__restrict__ s,
                                     const double *__restrict__ zl, const double *__restrict__ zu, double mu,
                                     double *__restrict__ sig, double *__restrict__ w, double *__restrict__ stream) {
#pragma unroll 4
  for (int i = threadIdx.x; i < P.n_eqw; i += blockDim.x) {
    const int r = P.eq_idx[i];
    stream[P.rhs_pos[r]] = -g[r];
  }
#pragma unroll 4
  for (int i = threadIdx.x; i < P.n_iq; i += blockDim.x) {
    const int r = P.iq_idx[i];
    const double l = P.iq_lo[i], u = P.iq_hi[i];
    const bool hl = l > -1e19, hu = u < 1e19;
    const double dl = hl ? s[r] - l : 1.0, du = hu ? u - s[r] : 1.0;
    const double zql = zl[r] / dl, zqu = zu[r] / du, ml = mu / dl, mq = mu / du;   // (unconditional: the four division chains interleave)
    const double sg = (hl ? zql : 0.0) + (hu ? zqu : 0.0);
    const double gmu = -(hl ? ml : 0.0) + (hu ? mq : 0.0);
    const double wr = sg * (g[r] - s[r]) + gmu;
    sig[r] = sg;
    w[r] = wr;
    stream[P.sig_pos[r]] = sg;
    stream[P.w_pos[r]] = wr;
  }
}

// max violation of the working rows (viol) and of the slack form (theta)
__device__ inline void infeasibility(const DevPlan &P, const double *__restrict__ g, const double *__restrict__ s,
                                     double *scratch, double &viol, double &theta) {
  double v = 0, t = 0;
#pragma unroll 4
  for (int i = threadIdx.x; i < P.n_eqw; i += blockDim.x) {
    const double gr = g[P.eq_idx[i]];
    if (!(gr == gr)) { v = t = INFINITY; continue; }   // fmax would swallow a NaN
    v = fmax(v, fabs(gr));
    t = fmax(t, fabs(gr));
  }
#pragma unroll 4
  for (int i = threadIdx.x; i < P.n_iq; i += blockDim.x) {
    const int r = P.iq_idx[i];
    const double gr = g[r], sr = s[r];
    if (!(gr == gr) || !(sr == sr)) { v = t = INFINITY; continue; }
    v = fmax(v, fmax(P.iq_lo[i] - gr, gr - P.iq_hi[i]));
    t = fmax(t, fabs(gr - sr));
  }
  viol = wg_reduce<1>(v, scratch);
  theta = wg_reduce<1>(t, scratch);
}

__device__ inline double l1_infeasibility(const DevPlan &P, const double *__restrict__ g, const double *__restrict__ s,
                                          const double *__restrict__ ds, double al, double *scratch) {
  double t = 0;
#pragma unroll 4
  for (int i = threadIdx.x; i < P.n_eqw; i += blockDim.x) t += fabs(g[P.eq_idx[i]]);
#pragma unroll 4
  for (int i = threadIdx.x; i < P.n_iq; i += blockDim.x) {
    const int r = P.iq_idx[i];
    t += fabs(g[r] - (s[r] + al * ds[r]));
  }
  return wg_reduce<0>(t, scratch);
}

__device__ inline void record_trace(const DevPlan &P, const DevWork &W, int b, int it, double viol,
                                    double theta, double al, double mu) {
  if (W.trace && it < P.max_iter + 1) {
    double *t = W.trace + ((size_t)b * (P.max_iter + 1) + it) * 4;
    t[0] = viol; t[1] = theta; t[2] = al; t[3] = mu;
  }
}

// A problem that is finished (converged, stalled, failed, or out of iterations) hands its result over on the spot: no export
// kernel behind the last iteration, whose launch the host could only queue after reading that it was the last.  x = the
// nodes as the workgroup's threads left them in memory (the caller has synchronised the workgroup).
__device__ inline void export_problem(const DevPlan &P, const DevWork &W, int b, const double *x, int status, int iters, double viol) {
  const size_t n = P.n_vars;
  for (int v = threadIdx.x; v < P.n_vars; v += blockDim.x) W.nodes_out[(size_t)b * n + v] = x[v];
  if (threadIdx.x == 0) {
    if (W.status_out) W.status_out[b] = status;
    if (W.iters_out) W.iters_out[b] = iters;
    if (W.viol_out) W.viol_out[b] = viol;
    if (status == 0) atomicAdd(W.totals, 1ull);
    atomicAdd(W.totals + 1, (unsigned long long)iters);
  }
}

// ---- starting point of a solve ---------------------------------------------------------------------
// With a table of nominal plans: bilinear interpolation over the goal displacement (clamped to the
// grid), then every position-like variable is shifted by the difference between the problem's start
// state and the interpolated plan's own (the flat-ground problem is translation invariant; the start
// stance / height / attitude of the problem need not be the nominal one).
struct TableCell {
  int i0, i1, j0, j1;
  double wx, wy;
};
__device__ inline TableCell table_cell(const DevPlan &P, const double *st, const double *gl) {
  TableCell c = {0, 0, 0, 0, 0.0, 0.0};
  if (!P.table) return c;
  const double dx = gl[0] - st[0], dy = gl[1] - st[1];
  int i = 0, j = 0;
  while (i + 2 < P.tab_ndx && dx >= P.tab_dx[i + 1]) ++i;
  while (j + 2 < P.tab_ndy && dy >= P.tab_dy[j + 1]) ++j;
  c.i0 = i; c.i1 = min(i + 1, P.tab_ndx - 1);
  c.j0 = j; c.j1 = min(j + 1, P.tab_ndy - 1);
  c.wx = c.i1 > c.i0 ? fmin(fmax((dx - P.tab_dx[c.i0]) / (P.tab_dx[c.i1] - P.tab_dx[c.i0]), 0.0), 1.0) : 0.0;
  c.wy = c.j1 > c.j0 ? fmin(fmax((dy - P.tab_dy[c.j0]) / (P.tab_dy[c.j1] - P.tab_dy[c.j0]), 0.0), 1.0) : 0.0;
  return c;
}
__device__ inline double table_value(const DevPlan &P, const TableCell &c, int v) {
  const size_t n = P.n_vars;
  const double a = P.table[((size_t)c.j0 * P.tab_ndx + c.i0) * n + v], b = P.table[((size_t)c.j0 * P.tab_ndx + c.i1) * n + v];
  const double d = P.table[((size_t)c.j1 * P.tab_ndx + c.i0) * n + v], e = P.table[((size_t)c.j1 * P.tab_ndx + c.i1) * n + v];
  return (1.0 - c.wy) * ((1.0 - c.wx) * a + c.wx * b) + c.wy * ((1.0 - c.wx) * d + c.wx * e);
}
__device__ inline double straight_line_value(const DevPlan &P, const InitDesc &I, const double *st, const double *gl, int map);
__device__ inline double initial_value(const DevPlan &P, const DevWork &W, int b, int v, const double *st, const double *gl,
                                       int map, const TableCell &tc) {
  const InitDesc I = P.init[v];
  if (I.fix_src >= 0) return I.fix_src < 24 ? st[I.fix_src] : (I.fix_src < 26 ? gl[I.fix_src - 24] : 0.0);
  if (W.warm) return W.warm[(size_t)b * P.n_vars + v];
  if (P.table) {
    double val = table_value(P, tc, v);
    if (!I.is_vel && I.set < 6) {
      const int ref = (I.set == 0 ? P.off_lin : (I.set == 1 ? P.off_ang : P.off_eem[I.set - 2])) + I.dim;
      const int src = I.set == 0 ? I.dim : (I.set == 1 ? 3 + I.dim : 6 + 3 * (I.set - 2) + I.dim);
      val += st[src] - table_value(P, tc, ref);
    }
    return val;
  }
  return straight_line_value(P, I, st, gl, map);
}
// towr's straight-line guess (nlp_formulation.cc Make*Variables)
__device__ inline double straight_line_value(const DevPlan &P, const InitDesc &I, const double *st, const double *gl, int map) {
  const double fin[3] = {gl[0], gl[1], terrain_at(P, map, gl[0], gl[1]).h - P.nominal[0][2]};
  double a, e;
  if (I.set == 0) { a = st[I.dim]; e = fin[I.dim]; }
  else if (I.set == 1) { a = st[3 + I.dim]; e = 0.0; }
  else if (I.set < 6) {
    const int ee = I.set - 2;
    a = st[6 + 3 * ee + I.dim];
    const double fx = fin[0] + P.nominal[ee][0], fy = fin[1] + P.nominal[ee][1];
    e = I.dim == 0 ? fx : (I.dim == 1 ? fy : terrain_at(P, map, fx, fy).h);
  } else { a = e = I.dim == 2 ? P.mass * P.gravity / NEE : 0.0; }
  return I.is_vel ? (e - a) / P.T : a + I.frac * (e - a);
}

// =================================================================================================
__global__ __launch_bounds__(ET) void k_start(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  __shared__ double scratch[ET];
  extern __shared__ double evl[];
  const int n = P.n_vars, m = P.n_cons, tid = threadIdx.x;
  double *x = W.x + (size_t)b * n, *g = W.g + (size_t)b * m;
  double *s = W.s + (size_t)b * m, *zl = W.zl + (size_t)b * m, *zu = W.zu + (size_t)b * m;
  const double *st = W.start + (size_t)b * QTOS_START_DOUBLES, *gl = W.goal + (size_t)b * 3;
  const int map = W.map_id ? W.map_id[b] : 0;
  const TableCell tc = table_cell(P, st, gl);
  for (int v = tid; v < n; v += blockDim.x) {
    const double val = initial_value(P, W, b, v, st, gl, map, tc);
    x[v] = val;
    W.xbest[(size_t)b * n + v] = val;   // best iterate so far: the starting point
    evl[v] = val;                       // (the evaluation below reads the nodes from LDS)
  }
  __syncthreads();
  // reduced swings: the swing rows left the KKT system and hold for every iterate only if they hold for the first (towr's
  // straight-line guess has the whole-plan mean velocity at the mid nodes, given nodes hold the rule to their print precision)
  for (int i = tid; i < P.n_psw; i += blockDim.x) {
    const double val = fma(P.psw_w[2 * i], x[P.psw_src[2 * i]], P.psw_w[2 * i + 1] * x[P.psw_src[2 * i + 1]]);
    const int v = P.psw_var[i];
    x[v] = val;
    W.xbest[(size_t)b * n + v] = val;
    evl[v] = val;
  }
  if (P.n_psw) __syncthreads();
  const bool project = P.n_coef && (W.warm || P.table);
  if (project) {
    // reduced base: given nodes need not be a spline of the coefficients' space (the reference's plans violate the
    // acceleration continuity by their CSV precision, 7e-4): project them onto it -- the continuity rows that left the KKT
    // system hold for every iterate only if they hold for the first
    for (int c = tid; c < P.n_coef; c += blockDim.x) {
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a) acc = fma(P.pc_w[4 * c + a], x[P.pc_var[4 * c + a]], acc);
      evl[c] = acc;
    }
    __syncthreads();
    for (int i = tid; i < P.n_pz; i += blockDim.x) {
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a) acc = fma(P.pz_w[4 * i + a], evl[P.pz_col[4 * i + a]], acc);
      x[P.pz_var[i]] = acc;
      W.xbest[(size_t)b * n + P.pz_var[i]] = acc;
    }
    __syncthreads();
  }
  // values and linearisation of the starting point in ONE pass (the Jacobian does not depend on the slack
  // initialisation below; a problem that turns out converged or invalid has merely written a stream nobody reads)
  // (a projected start is staged from memory again: the projection used the scratch)
#ifdef QTOS_STAMPS
  eval_all<true>(P, map, project ? x : nullptr, g, W.stream + (size_t)b * P.stream_len, evl, W.trace ? W.trace + ((size_t)b * (P.max_iter + 1) + 72) * 4 : nullptr, 0);
#else
  eval_all<true>(P, map, project ? x : nullptr, g, W.stream + (size_t)b * P.stream_len, evl, nullptr, 0);
#endif
  __syncthreads();
  // slack initialisation: push strictly inside the bounds (Ipopt bound_push / bound_frac)
  for (int r = tid; r < m; r += blockDim.x) {
    if (P.row_kind[r] != 2) { s[r] = 0; zl[r] = 0; zu[r] = 0; continue; }
    const double l = P.con_lo[r], u = P.con_hi[r];
    const bool hl = l > -1e19, hu = u < 1e19;
    const double kp = (W.warm || P.table) ? P.warm_slack_push : P.slack_push;   // large push on cold starts, Ipopt's 0.01 on warm starts (a table guess is one)
    double pl = hl ? kp * fmax(1.0, fabs(l)) : 0.0, pu = hu ? kp * fmax(1.0, fabs(u)) : 0.0;
    if (hl && hu) { pl = fmin(pl, kp * (u - l)); pu = fmin(pu, kp * (u - l)); }
    double si = g[r];
    if (hl) si = fmax(si, l + pl);
    if (hu) si = fmin(si, u - pu);
    s[r] = si;
  }
  __syncthreads();   // infeasibility() reads s through the compact row lists: another thread-to-row mapping
  double viol, theta;
  infeasibility(P, g, s, scratch, viol, theta);
  const double mu = fmax(P.mu_min, fmin(P.mu_init, 0.01 * theta * theta));
  for (int r = tid; r < m; r += blockDim.x) {
    if (P.row_kind[r] != 2) continue;
    const double l = P.con_lo[r], u = P.con_hi[r];
    zl[r] = l > -1e19 ? mu / (s[r] - l) : 0.0;
    zu[r] = u < 1e19 ? mu / (u - s[r]) : 0.0;
  }
  const bool conv = viol <= P.tol && theta <= P.tol;
  const bool bad = !(viol < INFINITY) || !(theta < INFINITY);   // NaN / inf in the inputs
  if (tid == 0) {
    W.mu[b] = mu;
    W.viol[b] = viol;
    W.iters[b] = 0;
    W.status[b] = conv ? 0 : (bad ? 2 : 1);
    W.done[b] = (conv || bad || P.max_iter <= 0) ? 1 : 0;   // (done = nothing left to launch for: the host stops on the count alone)
    W.best_viol[b] = viol;
    W.best_it[b] = 0;
    W.held[b] = 0;
    W.chord[b] = 0;
    W.chord_run[b] = 0;
    W.jam[b] = 0;
    if (!conv && !bad && P.max_iter > 0) atomicAdd(W.n_active, 1);
    record_trace(P, W, b, 0, viol, theta, 0.0, mu);
  }
  if (conv || bad || P.max_iter <= 0) {
    __syncthreads();   // (x of the other threads)
    export_problem(P, W, b, x, conv ? 0 : (bad ? 2 : 1), 0, viol);
    return;
  }
  __syncthreads();
  barrier_terms(P, g, s, zl, zu, mu, W.sig + (size_t)b * m, W.w + (size_t)b * m, W.stream + (size_t)b * P.stream_len);
}

// =================================================================================================
// Building blocks of the KKT kernels (kkt2.hpp: k_kkt2, k_chord): LDS-only barrier, the in-register LDL^T of a 16 x 16 pivot
// block, the assembly of a stage's records.
constexpr int KT = 512;    // (record limits of the prefetch path are stated in units of it: qtos_planner_create)
constexpr int PLD = PIV + 1;
__device__ __forceinline__ int tri(int r, int c) { return ((r * (r + 1)) >> 1) + c; }  // c <= r
__device__ __forceinline__ int trs(int a, int b) { return a >= b ? tri(a, b) : tri(b, a); }

// Workgroup barrier that waits for LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would
// stall every phase on the in-flight prefetch loads and factor-panel stores (cdna_hip_programming.md
// section 5 "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
// row_newbcast:K -- every lane of a 16-lane DPP row reads lane K of its own row (gfx90a+)
template <int K>
__device__ __forceinline__ double bc16(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// ---- LDL^T of a 16 x 16 block in the split layout (round 3) --------------------------------------------------------
// The block is held ONCE by the wave, not once per 16-lane row: lane (li, lk) = (lane & 15, lane >> 4) keeps
// a[g] = B[li][4 g + lk], g = 0..3 -- row li, the four columns = lk (mod 4); this is also the accumulator layout of the
// f64 MFMA for a symmetric matrix.  Step K (column K = 4 gK + qK, held by row group qK): 1 / d_K and the multipliers
// l_i = B[i][K] / d_K exist in row group qK only and cross to the other three row groups with v_permlane32_swap +
// v_permlane16_swap; the rank-1 update of the trailing columns is then ONE row-broadcast FMA per register that still holds
// live columns (row_mask keeps the finished columns of register gK), and W = L^-1 (same layout, starts as I) gets the
// same row operation on its columns <= K: 5 FMAs per step instead of 15.
template <int Q>
__device__ __forceinline__ double bcast_rowgroup(double x) {   // the values of row group Q on all four row groups
  int lo = __double2loint(x), hi = __double2hiint(x);
  auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);   // [0] = groups (0, 1, 0, 1), [1] = groups (2, 3, 2, 3)
  auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  lo = l[Q >> 1];
  hi = h[Q >> 1];
  auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // [0] = the even groups of the source, [1] = the odd ones
  auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(h2[Q & 1], l2[Q & 1]);
}
// a += bcast_K(a) * b on the row groups selected by RM (callers keep two instructions between a VALU write of `a` and
// this cross-lane read of it: here `a` was last written a whole step earlier)
template <int K, int RM>
__device__ __forceinline__ void fma_bc_self_rows(double &a, double b) {
  if constexpr (RM != 0)
    asm volatile("v_fmac_f64 %0, %0, %1 row_newbcast:%2 row_mask:%3 bank_mask:0xf" : "+v"(a) : "v"(b), "n"(K), "n"(RM));
}
// bc16 with the two wait states a DPP read of a fresh VALU result needs (the compiler does not see through the inline
// assembly that wrote the source)
template <int K>
__device__ __forceinline__ double bc16_safe(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v), olo, ohi;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
               : "=&v"(olo), "=&v"(ohi) : "v"(lo), "v"(hi), "n"(K));
  return __hiloint2double(ohi, olo);
}
template <int K>
__device__ __forceinline__ void ldlt16s_steps(double (&a)[4], double (&w)[4], double &myinv, int li, int lk) {
  if constexpr (K < PIV) {
    constexpr int gK = K >> 2, qK = K & 3;
    const double inv = fast_rcp(bc16_safe<K>(a[gK]));   // 1 / d_K on row group qK (other groups: don't care)
    if (li == K && lk == qK) myinv = inv;
    if constexpr (K < PIV - 1) {
      double nl = li > K ? -(a[gK] * inv) : 0.0;    // -L[i][K] for rows i > K, 0 for finished rows (row group qK)
      nl = bcast_rowgroup<qK>(nl);
      // trailing columns: register gK holds columns 4 gK + (0..3): only the row groups behind qK; later registers whole
      fma_bc_self_rows<K, (0xF << (qK + 1)) & 0xF>(a[gK], nl);
      if constexpr (gK + 1 < 4) fma_bc_self_rows<K, 0xF>(a[gK + 1 < 4 ? gK + 1 : 3], nl);
      if constexpr (gK + 2 < 4) fma_bc_self_rows<K, 0xF>(a[gK + 2 < 4 ? gK + 2 : 3], nl);
      if constexpr (gK + 3 < 4) fma_bc_self_rows<K, 0xF>(a[gK + 3 < 4 ? gK + 3 : 3], nl);
      // inverse factor: row i -= L[i][K] * (row K of W), columns <= K (W[K][K] = 1 delivers W[i][K] = -L[i][K])
      if constexpr (gK >= 1) fma_bc_self_rows<K, 0xF>(w[0], nl);
      if constexpr (gK >= 2) fma_bc_self_rows<K, 0xF>(w[1], nl);
      if constexpr (gK >= 3) fma_bc_self_rows<K, 0xF>(w[2], nl);
      fma_bc_self_rows<K, (1 << (qK + 1)) - 1>(w[gK], nl);
    }
    ldlt16s_steps<K + 1>(a, w, myinv, li, lk);
  }
}
// On return w[g] = (L^-1)[li][4 g + lk] (unit diagonal included, zeros above it) and, on the lanes with lk == (li & 3),
// myinv = 1 / d_li.
__device__ __forceinline__ void ldlt16s(double (&a)[4], double (&w)[4], double &myinv, int li, int lk) {
#pragma unroll
  for (int g = 0; g < 4; ++g) w[g] = li == 4 * g + lk ? 1.0 : 0.0;
  myinv = 0.0;
  asm volatile("s_nop 4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
  ldlt16s_steps<0>(a, w, myinv, li, lk);
}

#ifdef QTOS_STAMPS
// (accumulators in LDS, not in registers: the diagnostic build must not spill where the production build does not)
#define STAMPW(w, arr, i) do { if (tid == 64 * (w)) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); arr[i] += t_ - tl_##arr[0]; tl_##arr[0] = t_; } } while (0)
#else
#define STAMPW(w, arr, i) do {} while (0)
#endif

// Assembly of one stage's records (already in LDS) into A by `nth` threads, ONE pass without
// barriers: equality entries and multiplier right-hand sides are distinct targets, the inequality
// blocks go through the gather table (one thread per target entry sums its contributions
// sum_r sig_r G[r][a] G[r][c], or -sum_r G[r][a] w_r for the rhs, in a fixed order); no target of
// one kind is a target of another (pivot diagonals are added when the pivot columns are gathered).
constexpr int SHDR = 8;   // static record header ints
// sum over the (at most five) rows of one inequality block for one target: code = offset of the
// block's G in the dynamic record (12 bits) | a << 12 | c << 18 (63: right-hand side) | (n-1) << 24
// | (m-1) << 29.  All reads are issued together; rows beyond the block re-read its last row and are
// weighted by zero.
__device__ __forceinline__ double gather_term(const double *dbuf, int code) {
  const int a = (code >> 12) & 63, c = (code >> 18) & 63, qn = ((code >> 24) & 31) + 1, qm = (int)((unsigned)code >> 29) + 1;
  const double *Gb = dbuf + (code & 4095);
  const double *sg = Gb + qm * qn, *wq = sg + qm;
  const bool rhs = c == 63;
  const int cc = rhs ? a : c;
  // rows 0..2 always (most blocks have three rows: the range-of-motion boxes), rows 3 and 4 only when a lane of the
  // wave has a block that deep (friction pyramids)
  double ga[3], gc[3], sw[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // (rows beyond a shallow block's last read whatever follows it in the record -- still LDS -- and are discarded below)
    ga[r] = Gb[r * qn + a];
    gc[r] = Gb[r * qn + cc];
    sw[r] = rhs ? wq[r] : sg[r];
  }
  double acc = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double term = rhs ? -(ga[r] * sw[r]) : sw[r] * ga[r] * gc[r];
    acc += r < qm ? term : 0.0;
  }
  // a static contribution (c == 62: the proximal term between two B-spline coefficients of the reduced base): the value itself
  acc = c == 62 ? ga[0] : acc;
  if (__builtin_amdgcn_ballot_w64(qm > 3) != 0ull) {
    double gb[2], gd[2], sv[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int rr = min(r + 3, qm - 1);
      gb[r] = Gb[rr * qn + a];
      gd[r] = Gb[rr * qn + cc];
      sv[r] = rhs ? wq[rr] : sg[rr];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double term = rhs ? -(gb[r] * sv[r]) : sv[r] * gb[r] * gd[r];
      acc += r + 3 < qm ? term : 0.0;
    }
  }
  return acc;
}
// ---- Kronecker blocks (Symbolic::kron, k_kkt2<F, CONT, true>) -----------------------------------------------------
// The 33 sums of every Kronecker block of a record: lane t < 33 of a wave forms entry t of the blocks widx, widx + nw, ...
// (entries 0..26: T'[mu nu][d][e] = sum_i sig_i G[i][rep(mu, d)] G[i][rep(nu, e)]; 27..32: V'[mu][d] = sum_i G[i][rep(mu, d)] w_i)
__device__ __forceinline__ void kron_sums(const int *sbuf, const double *dbuf, double *ksm, int widx, int nw, int lane) {
  const int koff = sbuf[2] >> 9, nk = (sbuf[2] >> 5) & 15;
  if (nk == 0 || lane >= 33) return;
  const int *ks = sbuf + koff;
  const int t = lane;
  int mu, nu, d, e;
  if (t < 9) { mu = nu = 0; d = t / 3; e = t % 3; }
  else if (t < 18) { mu = 0; nu = 1; d = (t - 9) / 3; e = (t - 9) % 3; }
  else if (t < 27) { mu = nu = 1; d = (t - 18) / 3; e = (t - 18) % 3; }
  else { mu = nu = (t - 27) / 3; d = e = (t - 27) % 3; }
  const int sh1 = 5 * (3 * mu + d), sh2 = 5 * (3 * nu + e);
  for (int kb = widx; kb < nk; kb += nw) {
    const int *bd = ks + kb * 2;
    const int goff = bd[0] & 4095, n = ((bd[0] >> 12) & 31) + 1, reps = bd[1];
    const int c1 = (reps >> sh1) & 31, c2 = (reps >> sh2) & 31;
    const double *Gb = dbuf + goff, *sg = Gb + 3 * n, *wq = sg + 3;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double g1 = Gb[i * n + c1], g2 = Gb[i * n + c2], sv = sg[i], wv = wq[i];
      acc += t < 27 ? sv * g1 * g2 : g1 * wv;
    }
    ksm[kb * 33 + t] = acc;
  }
}
// one Kronecker contribution: rho_a rho_c T' (an entry of G' S G) or rho_a (-1) V' (of the right-hand side -G' w); rho = the
// record's weights (behind the blocks' two ints each)
__device__ __forceinline__ double kron_term(const double *rho, const double *ksm, int code) {
  return rho[(code >> 9) & 255] * rho[(unsigned)code >> 24] * ksm[code & 511];
}
__device__ __forceinline__ void assemble_stage_kron(double *A, int F, const int *sbuf, const double *dbuf, const double *ksm, int t0, int nth) {
  const int n_ent = sbuf[0], n_rhs = sbuf[1];
  const int *eidx = sbuf + SHDR + PIV;
  const double *eval = dbuf + PIV;
  for (int i = t0; i < n_ent; i += nth) A[eidx[i]] += eval[i];
  const int *rsl = eidx + n_ent;
  const double *rval = eval + n_ent;
  for (int i = t0; i < n_rhs; i += nth) A[rsl[i]] += rval[i];
  const int n_tgt = sbuf[5];
  if (n_tgt == 0) return;
  const int *tg = sbuf + sbuf[4];
  const int *cl = tg + n_tgt + 1;
  const double *ks = (const double *)(sbuf + (sbuf[2] >> 9) + 2 * ((sbuf[2] >> 5) & 15));   // the record's weights
  for (int t = t0; t < n_tgt; t += nth) {
    const int tv = tg[t], c0 = tv & 4095, c1 = tg[t + 1] & 4095;
    const double a_old = A[tv >> 12];
    double acc = 0;
    int j = c0;
    // a target's Kronecker contributions come first in its list: a loop of their own (a wave runs both kinds of a mixed
    // round one after the other)
    for (;;) {
      const int code = cl[min(j, c1 - 1)];
      const bool kk = j < c1 && ((code >> 18) & 63) == 61;
      if (__builtin_amdgcn_ballot_w64(kk) == 0ull) break;
      if (kk) { acc += kron_term(ks, ksm, code); ++j; }
    }
    for (; j < c1; ++j) acc += gather_term(dbuf, cl[j]);
    A[tv >> 12] = a_old + acc;
  }
}
__device__ __forceinline__ void assemble_stage(double *A, int F, const int *sbuf, const double *dbuf, int t0, int nth) {
  const int n_ent = sbuf[0], n_rhs = sbuf[1], n_iq = sbuf[2];
  const int *eidx = sbuf + SHDR + PIV;
  const double *eval = dbuf + PIV;
  for (int i = t0; i < n_ent; i += nth) A[eidx[i]] += eval[i];
  const int *rsl = eidx + n_ent;
  const double *rval = eval + n_ent;
  for (int i = t0; i < n_rhs; i += nth) A[rsl[i]] += rval[i];
  (void)n_iq;
  const int n_tgt = sbuf[5];
  if (n_tgt == 0) return;
  const int *tg = sbuf + sbuf[4];                     // n_tgt + 1 ints: (tri << 12) | first contribution
  const int *cl = tg + n_tgt + 1;                     // one self-contained int per contribution
  for (int t = t0; t < n_tgt; t += nth) {
    const int tv = tg[t], c0 = tv & 4095, c1 = tg[t + 1] & 4095;
    const double a_old = A[tv >> 12];                 // issued with the record reads, needed last
    double acc = 0;
    for (int j = c0; j < c1; ++j) acc += gather_term(dbuf, cl[j]);
    A[tv >> 12] = a_old + acc;
  }
}

// sum over the four lanes of a quad, on every lane of it
__device__ __forceinline__ double quadsum(double t) {
  double o = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(t), 0xB1, 0xf, 0xf, false),
                              __builtin_amdgcn_update_dpp(0, __double2loint(t), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  t += o;
  o = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(t), 0x4E, 0xf, 0xf, false),
                       __builtin_amdgcn_update_dpp(0, __double2loint(t), 0x4E, 0xf, 0xf, false));          // quad_perm [2,3,0,1]
  return t + o;
}

// =================================================================================================
// slot: the number of this launch within its call; kinds: the solve kernels launched in front of it (bit 0 the factorising
// kernel, bit 1 k_chord).  A launch queued from a PATTERN (qtos_plan_submit: the kernels the handle's last calls needed, without
// a look at the counts) may not have run the kernel a problem was waiting for: that problem sits the launch out -- nothing of
// its state moves, it reports the slot in n_active[2] -- and takes the step behind a later launch.  Its iteration number is
// therefore its own (W.iters), not the launch's: the plans do not depend on how the launches were queued.
__global__ __launch_bounds__(ET) void k_step(DevPlan P, DevWork W, int B, int slot, int kinds) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int it = W.iters[b];
  // (a finished problem leaves behind the first barrier, not here: the test of its flag is a memory round trip, and everything
  //  the kernel reads first would queue behind it)
  const int done_flag = W.done[b];
  __shared__ double scratch[ET];
  extern __shared__ double evl[];
  const int n = P.n_vars, m = P.n_cons, tid = threadIdx.x;
  // distinct buffers: __restrict__ lets the row loops below keep several rows' loads in flight
  double *__restrict__ x = W.x + (size_t)b * n, *__restrict__ dx = W.dx + (size_t)b * P.n_sol;
  double *__restrict__ g = W.g + (size_t)b * m, *__restrict__ gt = W.gt + (size_t)b * m;
  double *__restrict__ s = W.s + (size_t)b * m, *__restrict__ zl = W.zl + (size_t)b * m, *__restrict__ zu = W.zu + (size_t)b * m;
  double *__restrict__ ds = W.ds + (size_t)b * m, *__restrict__ dzl = W.dzl + (size_t)b * m, *__restrict__ dzu = W.dzu + (size_t)b * m;
  const double *__restrict__ G = W.stream + (size_t)b * P.stream_len;
  const int map = W.map_id ? W.map_id[b] : 0;
  double mu = W.mu[b];
  const double best_viol = W.best_viol[b];   // read before anybody writes them (tid 0, end of the kernel)
  const int best_it = W.best_it[b];
  const int chord_state = W.chord[b];        // 1: this iteration's dx came from a chord step; 2: chord steps are off for this solve
  const bool was_chord = chord_state == 1;
  const bool missed = !((was_chord ? 2 : 1) & kinds);   // the solve this problem waits for was not part of the launch
  const int chord_run = W.chord_run[b];
  const int jam_prev = W.jam[b];
  const double prev_viol = W.viol[b];        // violation in front of this step
  const int held0 = W.held[b];
  // The first trial point of the line search is evaluated WITH its Jacobian when the solve will very likely go on from it
  // (a Newton step is accepted whole at its fraction-to-the-boundary length nearly always, and a solve converges behind a
  // chord step): the linearisation at the accepted point is then this evaluation instead of a second pass over the same
  // point.  A wrong guess costs the difference of the two passes once (the problem converged: nobody reads the stream; the
  // trial point was cut: the linearisation below runs as before).
  // (not where Newton's method is about to converge: from a violation v the next one is about 0.05 v^2 on these problems --
  //  0.11 -> 5.7e-4 on the flat walk --, and a converged solve needs no Jacobian: --tol 1e-3 ends behind that step)
  const bool spec = P.spec_jac && !was_chord && it + 1 < P.max_iter && !(0.05 * prev_viol * prev_viol <= P.tol);
#ifdef QTOS_STAMPS
  unsigned long long ks[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kt0 = 0;
#define KSTAMP(i) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ks[i] += t_ - kt0; kt0 = t_; } } while (0)
  if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(kt0) :: "memory");
#else
#define KSTAMP(i) do {} while (0)
#endif
  // ds = Ji dx + (g - s), dx staged in LDS.  Every pass over a table below is a chain of memory round trips (index ->
  // value -> ...) of a microsecond each: the table reads that depend on nothing this kernel computes are issued first.
  const int nt = blockDim.x;
  // (a) reduced base: the step of the base node values from the step of the B-spline coefficients (dx_nodes = Z dc),
  //     RR rows per thread and batch
  constexpr int RR = 3;
  int rcol[RR][4], rvar[RR];
  double rw[RR][4];
  auto rec_load = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int i = min(i0 + r * nt, P.n_rec - 1);
      rvar[r] = P.rec_var[i];
#pragma unroll
      for (int a = 0; a < 4; ++a) { rcol[r][a] = P.rec_col[4 * i + a]; rw[r][a] = P.rec_w[4 * i + a]; }
    }
  };
  // (b) rows of the inequality blocks, four lanes per row: lane q takes the entries q, q + 4, ... of the whole groups of
  //     four (lane 0 the up to three entries behind them as well); the lanes of a quad read 32 consecutive bytes of the
  //     row.  TP lane tasks per thread and round, the row descriptors of the next round in flight during this one.
  constexpr int TP = 2, RU = 8;
  const int ntask = (4 * P.n_iq_rows + 63) & ~63, q = tid & 3;
  auto row_of = [&](int i4) __attribute__((always_inline)) { return P.iq_rows[min(i4 >> 2, P.n_iq_rows - 1)]; };
  IqRow Rn[TP];
  if (P.n_rec) rec_load(tid);
  if (!P.sw_on) {
#pragma unroll
    for (int t = 0; t < TP; ++t) Rn[t] = row_of(tid + t * nt);
  }
  // (c) the rows of the working set this thread owns: inequality rows tid + k nt (k < KR) and equality rows tid + k nt
  //     (k < KE) with their bounds, slacks, multipliers and values live in registers from here to the end of the kernel -- every
  //     pass below (ratio test, merit function, update, convergence test, barrier weights) read them through the row lists
  //     again: two memory round trips per pass and ten array passes of traffic per launch.  Rows beyond KR nt / KE nt (longer
  //     horizons) go through memory as before.
  constexpr int KR = 2, KE = 2;
  int rr[KR], er[KE];
  double rl[KR], ru[KR], rs[KR], rzl[KR], rzu[KR], rg[KR], eg[KE];
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int ic = min(tid + k * nt, max(P.n_iq - 1, 0));
    rr[k] = P.n_iq > 0 ? P.iq_idx[ic] : 0; rl[k] = P.iq_lo[ic]; ru[k] = P.iq_hi[ic];   // (an empty list is one unset element)
  }
#pragma unroll
  for (int k = 0; k < KE; ++k) er[k] = P.n_eqw > 0 ? P.eq_idx[min(tid + k * nt, max(P.n_eqw - 1, 0))] : 0;
  const bool rows_in_regs = P.n_iq <= KR * nt && P.n_eqw <= KE * nt;
  for (int v = tid; v < P.n_sol; v += nt) evl[v] = dx[v];
  // (behind the staging, whose stores waited for every load above: the row indices are there)
#pragma unroll
  for (int k = 0; k < KR; ++k) { rs[k] = s[rr[k]]; rzl[k] = zl[rr[k]]; rzu[k] = zu[rr[k]]; rg[k] = g[rr[k]]; }
#pragma unroll
  for (int k = 0; k < KE; ++k) eg[k] = g[er[k]];
  lds_barrier();   // (LDS only: the loads of the thread's rows stay in flight across it)
  if (done_flag) return;
  if (missed) {
    if (tid == 0) atomicMax(W.n_active + 2, (1 << 20) - slot);   // (the EARLIEST slot somebody sat out; 0 = nobody)
    return;
  }
  if (P.n_rec) {
    for (int i0 = tid;;) {
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) acc = rcol[r][a] >= 0 ? fma(rw[r][a], evl[max(rcol[r][a], 0)], acc) : acc;
        if (i0 + r * nt < P.n_rec) {
          dx[rvar[r]] = acc;
          evl[rvar[r]] = acc;   // (node values: no thread reads them in this loop, the sources are coefficients)
        }
      }
      i0 += RR * nt;
      if (i0 - tid >= P.n_rec) break;   // (uniform: the whole workgroup leaves together)
      rec_load(i0);
    }
    __syncthreads();
  }
  for (int base = 0; base < (P.sw_on ? 0 : ntask); base += TP * nt) {   // (sw_on: the backward sweep has left ds)
    IqRow R[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) R[t] = Rn[t];
    if (base + TP * nt < ntask) {
#pragma unroll
      for (int t = 0; t < TP; ++t) Rn[t] = row_of(base + TP * nt + t * nt + tid);
    }
    double gv[TP][RU], gr[TP][3], g_row[TP], s_row[TP];
    int cv[TP][RU], cr[TP][3];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      g_row[t] = g[R[t].row];
      s_row[t] = s[R[t].row];
      const double *Gr = G + R[t].goff;
      const int *cols = P.block_cols + R[t].col_off;
      const int n4 = R[t].n & ~3;
#pragma unroll
      for (int u = 0; u < RU; ++u) { const int e = max(min(q + 4 * u, n4 - 4 + q), 0); gv[t][u] = Gr[e]; cv[t][u] = cols[e]; }
#pragma unroll
      for (int u = 0; u < 3; ++u) { const int e = min(n4 + u, R[t].n - 1); gr[t][u] = Gr[e]; cr[t][u] = cols[e]; }
    }
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int i4 = base + t * nt + tid;
      if (i4 >= ntask) continue;   // (whole waves)
      const int n4 = R[t].n & ~3;
      double ev[RU], er[3];
#pragma unroll
      for (int u = 0; u < RU; ++u) ev[u] = evl[cv[t][u]];
#pragma unroll
      for (int u = 0; u < 3; ++u) er[u] = evl[cr[t][u]];
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < RU; ++u) acc = q + 4 * u < n4 ? fma(gv[t][u], ev[u], acc) : acc;
      // (rows of more than 35 entries: the rest of the whole groups, at memory latency)
      const double *Gr = G + R[t].goff;
      const int *cols = P.block_cols + R[t].col_off;
      for (int a = q + 4 * RU; a < n4; a += 4) acc = fma(Gr[a], evl[cols[a]], acc);
      if (q == 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u) acc = n4 + u < R[t].n ? fma(gr[t][u], er[u], acc) : acc;
      }
      acc = quadsum(acc);   // (a0 + a1) + (a2 + a3) of the four running sums
      if (q == 0 && (i4 >> 2) < P.n_iq_rows) ds[R[t].row] = acc + (g_row[t] - s_row[t]);
    }
  }
  __syncthreads();
  KSTAMP(0);
  const double tau = fmax(0.99, 1.0 - mu);
  double amax = 1.0, az = 1.0;
  double rds[KR], rdzl[KR], rdzu[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) rds[k] = ds[rr[k]];
  // ratio tests and the step of the multipliers (one row; the formulas of the passes below are these lambdas, on registers
  // for the thread's own rows and on memory for the rest)
  // (an IEEE division is a chain of a dozen dependent f64 instructions, and the compiler leaves one that sits behind a condition
  //  in a branch of its own: eight divisions of a row ran one after the other, 16 k cycles for the thread's two rows.  Every
  //  quotient is formed unconditionally -- on 1.0 where the row has no such bound -- and selected: the chains interleave)
  auto ratio_row = [&](bool valid, double l, double u, double sv, double d, double zlv, double zuv, double &a, double &c) __attribute__((always_inline)) {
    const bool hl = valid && l > -1e19, hu = valid && u < 1e19;
    const double dl = hl ? sv - l : 1.0, du = hu ? u - sv : 1.0;
    const double ml = mu / dl, zql = zlv / dl, mq = mu / du, zqu = zuv / du;
    const double a_ = ml - zlv - zql * d, c_ = mq - zuv + zqu * d;
    a = hl ? a_ : 0.0;
    c = hu ? c_ : 0.0;
    const double r1 = tau * dl / -d, r2 = tau * du / d, r3 = tau * zlv / -a_, r4 = tau * zuv / -c_;
    amax = (hl && d < 0) ? fmin(amax, r1) : amax;
    amax = (hu && d > 0) ? fmin(amax, r2) : amax;
    az = (hl && a_ < 0) ? fmin(az, r3) : az;
    az = (hu && c_ < 0) ? fmin(az, r4) : az;
  };
#pragma unroll
  for (int k = 0; k < KR; ++k) ratio_row(tid + k * nt < P.n_iq, rl[k], ru[k], rs[k], rds[k], rzl[k], rzu[k], rdzl[k], rdzu[k]);
  for (int i = tid + KR * nt; i < P.n_iq; i += nt) {
    const int r = P.iq_idx[i];
    double a, c;
    ratio_row(true, P.iq_lo[i], P.iq_hi[i], s[r], ds[r], zl[r], zu[r], a, c);
    dzl[r] = a;
    dzu[r] = c;
  }
  amax = wg_reduce<2>(amax, scratch);
  az = wg_reduce<2>(az, scratch);
  // l1 infeasibility of (c_E, c_I - (s + alpha ds)) for constraint values gI / gE (registers) and gm (memory), summed in the
  // order of l1_infeasibility: the thread's equality rows, then its inequality rows
  auto l1_rows = [&](const double (&gI)[KR], const double (&gE)[KE], const double *__restrict__ gm, double alq) __attribute__((always_inline)) {
    double t = 0;
#pragma unroll
    for (int k = 0; k < KE; ++k)
      if (tid + k * nt < P.n_eqw) t += fabs(gE[k]);
    for (int i = tid + KE * nt; i < P.n_eqw; i += nt) t += fabs(gm[P.eq_idx[i]]);
#pragma unroll
    for (int k = 0; k < KR; ++k)
      if (tid + k * nt < P.n_iq) t += fabs(gI[k] - (rs[k] + alq * rds[k]));
    for (int i = tid + KR * nt; i < P.n_iq; i += nt) {
      const int r = P.iq_idx[i];
      t += fabs(gm[r] - (s[r] + alq * ds[r]));
    }
    return wg_reduce<0>(t, scratch);
  };
  const double th0 = l1_rows(rg, eg, g, 0.0);
  KSTAMP(1);
  // backtracking on the l1 infeasibility of (c_E, c_I - s)
  double al = amax, th = 0;
  double rgt[KR], egt[KE];
  bool lin_done = false;   // g and the stream hold the linearisation at the accepted point already
  for (int ls = 0; ls < 6; ++ls) {
    // the trial point goes to LDS directly (and stays there for the linearisation below if it is accepted): written to
    // memory and staged back it would cost two memory round trips
    for (int v = tid; v < n; v += blockDim.x) evl[v] = x[v] + al * dx[v];
    __syncthreads();
    double *gv = (spec && ls == 0) ? g : gt;   // (th0 and the thread's rows of g are in registers)
    if (spec && ls == 0)
      eval_all<true>(P, map, nullptr, g, W.stream + (size_t)b * P.stream_len, evl, (W.trace && it == 1) ? W.trace + ((size_t)b * (P.max_iter + 1) + 72) * 4 : nullptr, held0);
    else
      eval_all<false>(P, map, nullptr, gt, nullptr, evl, (W.trace && it == 1 && ls == 0) ? W.trace + ((size_t)b * (P.max_iter + 1) + 76) * 4 : nullptr);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KR; ++k) rgt[k] = gv[rr[k]];
#pragma unroll
    for (int k = 0; k < KE; ++k) egt[k] = gv[er[k]];
    th = l1_rows(rgt, egt, gv, al);
    if (th <= (1.0 - 1e-4 * al) * th0 || th < 1e-9) { lin_done = spec && ls == 0; break; }
    if (ls < 5) al *= 0.5;
  }
  KSTAMP(2);
  // a chord step is taken whole or not at all: cut by the fraction-to-the-boundary rule or by the line search it
  // is discarded (the iterate stays, the next iteration factors): a damped chord step can park a slack right on
  // its bound, and the KKT matrix of that point is too badly scaled for the block elimination
  const bool reject = was_chord && al != 1.0;
  if (reject) { al = 0.0; az = 0.0; th = 0.0; }
  else {
    for (int v = tid; v < n; v += blockDim.x) x[v] = evl[v];   // (the trial point of the last evaluation)
#pragma unroll
    for (int k = 0; k < KR; ++k) rg[k] = rgt[k];
#pragma unroll
    for (int k = 0; k < KE; ++k) eg[k] = egt[k];
    // (the constraint values in memory: rewritten by the linearisation below whenever the solve goes on; copied here only
    //  for the rows that are read from memory before that)
    if (!rows_in_regs && !lin_done) {
#pragma unroll 4
      for (int r = tid; r < m; r += blockDim.x) g[r] = gt[r];
    }
  }
  auto update_row = [&](double l, double u, double &sv, double d, double &zlv, double &zuv, double dl_, double du_) __attribute__((always_inline)) {
    const bool hl = l > -1e19, hu = u < 1e19;
    const double sn = sv + al * d;
    sv = sn;
    const double a = zlv + az * dl_, c = zuv + az * du_;
    const double kap = 1e10;
    const double gl = hl ? sn - l : 1.0, gu = hu ? u - sn : 1.0;   // (the four quotients side by side, as in ratio_row)
    const double lo_l = mu / (kap * gl), hi_l = kap * mu / gl, lo_u = mu / (kap * gu), hi_u = kap * mu / gu;
    zlv = hl ? fmin(fmax(a, lo_l), hi_l) : a;
    zuv = hu ? fmin(fmax(c, lo_u), hi_u) : c;
  };
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const bool valid = tid + k * nt < P.n_iq;
    double sv = rs[k], zlv = rzl[k], zuv = rzu[k];
    update_row(rl[k], ru[k], sv, rds[k], zlv, zuv, rdzl[k], rdzu[k]);
    if (valid) {
      rs[k] = sv; rzl[k] = zlv; rzu[k] = zuv;
      s[rr[k]] = sv; zl[rr[k]] = zlv; zu[rr[k]] = zuv;
    }
  }
  for (int i = tid + KR * nt; i < P.n_iq; i += nt) {
    const int r = P.iq_idx[i];
    double sv = s[r], zlv = zl[r], zuv = zu[r];
    update_row(P.iq_lo[i], P.iq_hi[i], sv, ds[r], zlv, zuv, dzl[r], dzu[r]);
    s[r] = sv; zl[r] = zlv; zu[r] = zuv;
  }
  if (al > 0.3) mu = P.mu_superlinear ? fmax(fmax(P.mu_min, P.tol), fmin(0.2 * mu, mu * sqrt(mu))) : fmax(P.mu_min, 0.2 * mu);
  __syncthreads();
  // max violation of the working rows (viol) and of the slack form (theta): infeasibility() on the rows in registers
  double viol, theta;
  {
    double v = 0, t = 0;
    auto eq_row = [&](double gr) __attribute__((always_inline)) {
      if (!(gr == gr)) { v = t = INFINITY; return; }   // fmax would swallow a NaN
      v = fmax(v, fabs(gr));
      t = fmax(t, fabs(gr));
    };
    auto iq_row = [&](double gr, double sr, double l, double u) __attribute__((always_inline)) {
      if (!(gr == gr) || !(sr == sr)) { v = t = INFINITY; return; }
      v = fmax(v, fmax(l - gr, gr - u));
      t = fmax(t, fabs(gr - sr));
    };
#pragma unroll
    for (int k = 0; k < KE; ++k)
      if (tid + k * nt < P.n_eqw) eq_row(eg[k]);
    for (int i = tid + KE * nt; i < P.n_eqw; i += nt) eq_row(g[P.eq_idx[i]]);
#pragma unroll
    for (int k = 0; k < KR; ++k)
      if (tid + k * nt < P.n_iq) iq_row(rg[k], rs[k], rl[k], ru[k]);
    for (int i = tid + KR * nt; i < P.n_iq; i += nt) {
      const int r = P.iq_idx[i];
      iq_row(g[r], s[r], P.iq_lo[i], P.iq_hi[i]);
    }
    viol = wg_reduce<1>(v, scratch);
    theta = wg_reduce<1>(t, scratch);
  }
  KSTAMP(3);
  const bool conv = viol <= P.tol && theta <= P.tol;
  const bool bad = !(viol < INFINITY) || !(th < INFINITY);
  // stall detection: the iterate with the lowest violation is kept; a problem that has not improved
  // it for stall_iters iterations (cycling on a discontinuous terrain edge) stops with status 1 and
  // returns that iterate
  const bool improved = !conv && !bad && viol < best_viol;
  // jammed against its bounds (two steps in a row shorter than stall_alpha; a discarded chord step does not count): stops
  // like a stalled problem
  const int jam = (P.stall_alpha > 0 && !reject && al < P.stall_alpha) ? jam_prev + 1 : 0;
  const bool stalled = !conv && !bad && ((!improved && P.stall_iters > 0 && it + 1 - best_it >= P.stall_iters) || jam >= 2);
  if (improved)
    for (int v = tid; v < n; v += blockDim.x) W.xbest[(size_t)b * n + v] = x[v];
  // a numerical failure (NaN / inf after a step from a finite iterate) hands back that best iterate too
  const bool restore = stalled || (bad && best_viol < INFINITY);
  if (restore)
    for (int v = tid; v < n; v += blockDim.x) x[v] = W.xbest[(size_t)b * n + v];
  if (tid == 0) {
    W.mu[b] = mu;
    W.viol[b] = restore ? (improved ? viol : best_viol) : viol;   // (a jammed problem may have improved in its last step)
    W.iters[b] = it + 1;
    W.jam[b] = jam;
    if (improved) { W.best_viol[b] = viol; W.best_it[b] = it + 1; }
    record_trace(P, W, b, it + 1, viol, theta, al, mu);
    if (conv || bad || stalled) W.status[b] = conv ? 0 : (bad ? 2 : 1);
    if (conv || bad || stalled || it + 1 >= P.max_iter) {   // (out of iterations: the status k_start gave, 1)
      W.done[b] = 1;
      W.chord[b] = 0;
      atomicAdd(W.n_active, -1);
    }
    atomicMax(W.n_active + 3, slot + 1);   // launches of the call that had work
  }
  // finished, or out of iterations: the result leaves now (x as restored above; a problem out of iterations keeps the status
  // k_start gave it: 1)
  if (conv || bad || stalled || it + 1 >= P.max_iter) {
    __syncthreads();
    export_problem(P, W, b, x, conv ? 0 : (bad ? 2 : 1), it + 1, restore ? (improved ? viol : best_viol) : viol);
  }
  if (conv || bad || stalled || it + 1 >= P.max_iter) return;
  // chord step next?  (an iterate this close, reached by a full step of a freshly factored system)
  // (one discarded chord step and the solve factors every iteration from then on: near a terrain edge the attempt
  //  fails again and again, and every failure is an iteration the whole batch waits for)
  const bool chord_off = chord_state == 2 || reject;
  // (a full chord step that brought the violation down to chord_shrink of what it was may be followed by another one)
  const bool next_chord = P.chord_tol > 0 && !chord_off && al == 1.0 && viol <= P.chord_tol &&
                          (!was_chord || (chord_run < P.chord_max && viol <= P.chord_shrink * prev_viol));
  // two-phase solve: latch the hold once this iterate is close enough
  const int held = (W.held[b] || (P.hold_from > 0 && it + 1 >= P.hold_from && viol <= P.hold_tol)) ? 1 : 0;
  __syncthreads();
  if (tid == 0) W.held[b] = held;
  if (!lin_done) {
    eval_all<true>(P, map, reject ? x : nullptr, g, W.stream + (size_t)b * P.stream_len, evl, (W.trace && it == 1) ? W.trace + ((size_t)b * (P.max_iter + 1) + 72) * 4 : nullptr,
                   held);
  } else if (held != held0) {
    // the footholds are held from this iterate on: the proximal weights of the stance footholds (eval_terr), the one thing
    // of the linearisation that depends on the latch
    double *Gp = W.stream + (size_t)b * P.stream_len;
    for (int ti = tid; ti < P.n_terr; ti += nt) {
      const TerrDev TI = P.terr[ti];
      if (TI.d0 >= 0) Gp[TI.d0] = P.hold_weight + TI.ex;
      if (TI.d1 >= 0) Gp[TI.d1] = P.hold_weight + TI.ey;
    }
  }
  __syncthreads();
  KSTAMP(4);
  {
    // barrier weights of every inequality row, right-hand sides of the equality rows (barrier_terms, on the rows in registers;
    // their constraint values as the linearisation above has just written them: the same point as the last line-search
    // evaluation, but the two evaluation passes need not round alike)
#pragma unroll
    for (int k = 0; k < KR; ++k) rg[k] = g[rr[k]];
#pragma unroll
    for (int k = 0; k < KE; ++k) eg[k] = g[er[k]];
    double *__restrict__ sigp = W.sig + (size_t)b * m, *__restrict__ wp = W.w + (size_t)b * m, *__restrict__ strm = W.stream + (size_t)b * P.stream_len;
    auto bar_row = [&](bool valid, int r, int sp, int wpos, double l, double u, double sv, double zlv, double zuv, double gv) __attribute__((always_inline)) {
      const bool hl = l > -1e19, hu = u < 1e19;
      const double dl = hl ? sv - l : 1.0, du = hu ? u - sv : 1.0;
      const double zql = zlv / dl, zqu = zuv / du, ml = mu / dl, mq = mu / du;   // (side by side, as in ratio_row)
      const double sg = (hl ? zql : 0.0) + (hu ? zqu : 0.0);
      const double gmu = -(hl ? ml : 0.0) + (hu ? mq : 0.0);
      const double wrv = sg * (gv - sv) + gmu;
      if (valid) {
        sigp[r] = sg;
        wp[r] = wrv;
        strm[sp] = sg;
        strm[wpos] = wrv;
      }
    };
#pragma unroll
    for (int k = 0; k < KE; ++k)
      if (tid + k * nt < P.n_eqw) strm[P.rhs_pos[er[k]]] = -eg[k];
    for (int i = tid + KE * nt; i < P.n_eqw; i += nt) { const int r = P.eq_idx[i]; strm[P.rhs_pos[r]] = -g[r]; }
    int spos[KR], wpos[KR];   // (the positions first: their loads are in flight during the divisions)
#pragma unroll
    for (int k = 0; k < KR; ++k) { spos[k] = P.sig_pos[rr[k]]; wpos[k] = P.w_pos[rr[k]]; }
#pragma unroll
    for (int k = 0; k < KR; ++k) bar_row(tid + k * nt < P.n_iq, rr[k], spos[k], wpos[k], rl[k], ru[k], rs[k], rzl[k], rzu[k], rg[k]);
    for (int i = tid + KR * nt; i < P.n_iq; i += nt) {
      const int r = P.iq_idx[i];
      bar_row(true, r, P.sig_pos[r], P.w_pos[r], P.iq_lo[i], P.iq_hi[i], s[r], zl[r], zu[r], g[r]);
    }
  }
  KSTAMP(5);
  if (tid == 0) {
    W.chord[b] = next_chord ? 1 : (chord_off ? 2 : 0);
    W.chord_run[b] = next_chord ? (was_chord ? chord_run + 1 : 1) : 0;
    if (next_chord) atomicAdd(W.n_active + 1, 1);
  }
  if (next_chord) {
    // right-hand side of the KKT system at the new iterate, in elimination order, for k_chord: rhs[p] = -g[row] for a
    // multiplier, -sum_t G[gpos[t]] w[row[t]] over the unknown's list for a variable (a base coefficient has sixty entries
    // and every one is two dependent memory round trips: a list walked by its own thread costs a hundred of them).  The
    // factors of rhs_chunk entries at a time go to LDS with every load of the pass in flight, then every unknown sums its
    // part of the pass in list order.
    __syncthreads();   // the stream and w of this launch are complete
    const double *__restrict__ Gs = W.stream + (size_t)b * P.stream_len, *__restrict__ wr = W.w + (size_t)b * m;
    double *__restrict__ rhs = W.rhs + (size_t)b * P.n_unknowns;
    const int NUK = P.n_unknowns, CH = P.rhs_chunk;
    double *accL = evl;
    int *ptrL = (int *)(accL + ((NUK + 1) & ~1));
    double *gvL = (double *)(ptrL + ((NUK + 4) & ~3)), *wvL = gvL + CH;
    for (int p = tid; p <= NUK; p += nt) { ptrL[p] = P.rhs_ptr[p]; if (p < NUK) accL[p] = 0.0; }
    constexpr int SU = 12;   // entries per thread in flight
    for (int c0 = 0; c0 < P.n_rhs_ent; c0 += CH) {
      const int c1 = min(c0 + CH, P.n_rhs_ent);
      for (int t = c0 + tid; t < c1; t += SU * nt) {
        int gp[SU], rw[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) { const int tt = min(t + u * nt, c1 - 1); gp[u] = P.rhs_gpos[tt]; rw[u] = P.rhs_row[tt]; }
        double gv[SU], wv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {   // (a multiplier's single entry: 1 * g[row])
          gv[u] = gp[u] >= 0 ? Gs[max(gp[u], 0)] : 1.0;
          wv[u] = (gp[u] >= 0 ? wr : g)[rw[u]];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u)
          if (t + u * nt < c1) { gvL[t + u * nt - c0] = gv[u]; wvL[t + u * nt - c0] = wv[u]; }
      }
      __syncthreads();
      for (int p = tid; p < NUK; p += nt) {
        const int t0 = max(ptrL[p], c0), t1 = min(ptrL[p + 1], c1);
        if (t0 >= t1) continue;
        double acc = accL[p];
        for (int t = t0; t < t1; t += 8) {   // eight LDS round trips together, summed in list order
          double gq[8], wq[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { const int tt = min(t + u, t1 - 1) - c0; gq[u] = gvL[tt]; wq[u] = wvL[tt]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = t + u < t1 ? fma(-gq[u], wq[u], acc) : acc;
        }
        accL[p] = acc;
      }
      __syncthreads();
    }
    for (int p = tid; p < NUK; p += nt) rhs[p] = accL[p];
  }
#ifdef QTOS_STAMPS
  if (tid == 0 && W.trace && it == 1) for (int i = 0; i < 8; ++i) W.trace[((size_t)b * (P.max_iter + 1) + 70) * 4 + i] = (double)ks[i];
#endif
}

// =================================================================================================
// CSV sampling: row layout of QTOS/utils.py:107-148.  Spline lookup tables are built on the host.
struct SampleSpline {
  int n_polys;
  const double *tend;  // cumulative end time of each polynomial
  const double *dur;
  const int *idx;      // (n_polys+1) x 6
};
struct SamplePlan {
  SampleSpline lin, ang, eem[NEE], eef[NEE];
  int n_vars;
  double T;
};

__device__ inline void sample_spline(const SampleSpline &S, const double *x, double t, int deriv, double out[3]) {
  // first polynomial whose end time >= t - eps (towr spline.cc GetSegmentID)
  int lo = 0, hi = S.n_polys - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (S.tend[mid] >= t - 1e-10) hi = mid; else lo = mid + 1;
  }
  const int k = lo;
  const double T = S.dur[k], tau = t - (S.tend[k] - T);
  double w[4];
  const double T2 = T * T, T3 = T2 * T, t2 = tau * tau, t3 = t2 * tau;
  if (deriv == 0) {
    w[0] = 1 - 3 * t2 / T2 + 2 * t3 / T3; w[1] = tau - 2 * t2 / T + t3 / T2;
    w[2] = 3 * t2 / T2 - 2 * t3 / T3; w[3] = -t2 / T + t3 / T2;
  } else {
    w[0] = -6 * tau / T2 + 6 * t2 / T3; w[1] = 1 - 4 * tau / T + 3 * t2 / T2;
    w[2] = 6 * tau / T2 - 6 * t2 / T3; w[3] = -2 * tau / T + 3 * t2 / T2;
  }
  for (int d = 0; d < 3; ++d) {
    double acc = 0;
    for (int a = 0; a < 4; ++a) {
      const int v = S.idx[(k + (a >> 1)) * 6 + (a & 1) * 3 + d];
      if (v >= 0) acc += w[a] * x[v];
    }
    out[d] = acc;
  }
}

__global__ __launch_bounds__(256) void k_sample(SamplePlan S, const double *nodes, const double *t0, double hz,
                                                int n_rows, double *rows, int B) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || k >= n_rows) return;
  const double *x = nodes + (size_t)b * S.n_vars;
  double t = k / hz;
  if (t > S.T) t = S.T;
  double row[QTOS_CSV_COLS];
  row[0] = t0[b] + k / hz;
  sample_spline(S.lin, x, t, 0, row + 1);
  sample_spline(S.ang, x, t, 0, row + 4);
  for (int e = 0; e < NEE; ++e) sample_spline(S.eem[e], x, t, 0, row + 7 + 3 * e);
  sample_spline(S.lin, x, t, 1, row + 19);
  sample_spline(S.ang, x, t, 1, row + 22);
  for (int e = 0; e < NEE; ++e) sample_spline(S.eef[e], x, t, 0, row + 25 + 3 * e);
  double *out = rows + ((size_t)b * n_rows + k) * QTOS_CSV_COLS;
  for (int i = 0; i < QTOS_CSV_COLS; ++i) out[i] = row[i];
}

}  // namespace qtos
