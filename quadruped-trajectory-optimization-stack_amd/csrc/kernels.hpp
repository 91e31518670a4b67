// kernels.hpp -- gfx950 device code of the batched planner.  One workgroup per planning problem.
//
//   k_start : initial guess (or warm start), constraint values, slack / barrier initialisation,
//             first linearisation (per-instance dense Jacobian blocks G, barrier weights)
//   k_kkt   : fused front assembly + block LDL^T (Schur-complement chain, 16 pivots per stage,
//             front resident in LDS) + forward/backward substitution  -> Newton step dx
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

struct DevPlan {
  int n_vars, n_cons, n_stages, front;
  int n_dyn, n_rom, n_terr, n_force, n_lin, n_blocks;
  const DynInst *dyn;
  const RomInst *rom;
  const TerrInst *terr;
  const ForceInst *force;
  const LinRow *lin;
  const ColDesc *dyn_cols, *rom_cols;
  int n_dyn_cols, n_rom_cols;
  const Block *blocks;
  const int *block_cols;
  const double *g_static;
  const int *piv_slot, *piv_unknown;
  const double *piv_diag;
  const StageDesc *stages;
  const EqEntry *eq_entries;
  const EqRhs *eq_rhs;
  const IqBlock *iq_blocks;
  const short *iq_slots;
  int max_stage_g;
  const int *srec, *srec_off, *pack_src, *drec_off;  // packed per-stage records (symbolic.hpp)
  const int *eq_pos, *rhs_pos, *sig_pos, *w_pos;     // direct-write maps into the stream
  int max_srec, max_drec, stream_len;
  const double *con_lo, *con_hi;
  const int *row_kind;
  const InitDesc *init;
  double mass, gravity, Ib[9], mu_fric, f_max, T;
  double nominal[NEE][3];
  double tol, mu_init, mu_min, delta_x, eps_dual, slack_push;
  int max_iter;
  const double *height;
  int n_maps, hnx, hny, terrain_mode;
  double hcell, hx0, hy0;
  long long g_doubles, panel_stride;  // per problem
};

struct DevWork {
  const double *start, *goal, *warm;
  const int *map_id;
  double *x, *xt, *g, *gt, *s, *zl, *zu, *ds, *dzl, *dzu, *sig, *w, *G, *panel, *dx, *stream;
  double *mu, *viol, *trace;
  int *status, *iters, *done, *n_active;
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
__device__ inline double mk(double, double v) { return v; }
__device__ inline D1 mk(D1, double v) { return {v, 0.0}; }

// R = Rz(yaw) Ry(pitch) Rx(roll), th = (roll, pitch, yaw)
template <class T>
__device__ inline void rotation(const T th[3], T R[9]) {
  T sx, cx, sy, cy, sz, cz;
  sincos_t(th[0], sx, cx);
  sincos_t(th[1], sy, cy);
  sincos_t(th[2], sz, cz);
  R[0] = cy * cz; R[1] = cz * sx * sy - cx * sz; R[2] = sx * sz + cx * cz * sy;
  R[3] = cy * sz; R[4] = cx * cz + sx * sy * sz; R[5] = cx * sy * sz - cz * sx;
  R[6] = -sy;     R[7] = cy * sx;                R[8] = cx * cy;
}

// I_w wd + w x (I_w w) with I_w = R Ib R^T, w = M(th) thd, wd = Mdot thd + M thdd
template <class T>
__device__ inline void dyn_angular(const double *Ib, const T th[3], const T thd[3], const T thdd[3],
                                   T out[3]) {
  T sy, cy, sz, cz;
  sincos_t(th[1], sy, cy);
  sincos_t(th[2], sz, cz);
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
  rotation(th, R);
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

struct Terr {
  double h, hx, hy, hxy;
};
// bilinear height, clamped at the border (zero slope outside the map)
__device__ inline Terr terrain_at(const DevPlan &P, int map, double x, double y) {
  Terr t = {0, 0, 0, 0};
  if (!P.height || P.n_maps <= 0) return t;
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
//                                      and the column's combined Hermite weights (ColDesc)
constexpr int DYN_LOC = 54;  // A_th, A_thd, A_thdd (9 each), sum f (3), f_e (12), r - p_e (12)
constexpr int ROM_LOC = 18;  // R (9), d/dtheta_j [R^T (p - r)] as columns (9)

template <bool JAC>
__device__ inline void eval_dyn(const DevPlan &P, const DynInst &I, const double *x, double *g, double *loc) {
  double r[3], a[3], th[3], thd[3], thdd[3];
  vec_eval(I.r, x, r); vec_eval(I.a, x, a);
  vec_eval(I.th, x, th); vec_eval(I.thd, x, thd); vec_eval(I.thdd, x, thdd);
  double ga[3], gl[3];
  dyn_angular<double>(P.Ib, th, thd, thdd, ga);
  gl[0] = P.mass * a[0]; gl[1] = P.mass * a[1]; gl[2] = P.mass * a[2] + P.mass * P.gravity;
  const bool jac = JAC && I.in_kkt;
  if (jac) {
    // angular rows wrt Euler angles / rates / accelerations: 9 forward-mode passes
#pragma unroll
    for (int what = 0; what < 3; ++what)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        D1 t0[3], t1[3], t2[3], o[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          t0[i] = {th[i], (what == 0 && i == j) ? 1.0 : 0.0};
          t1[i] = {thd[i], (what == 1 && i == j) ? 1.0 : 0.0};
          t2[i] = {thdd[i], (what == 2 && i == j) ? 1.0 : 0.0};
        }
        dyn_angular<D1>(P.Ib, t0, t1, t2, o);
#pragma unroll
        for (int i = 0; i < 3; ++i) loc[9 * what + 3 * i + j] = o[i].d;
      }
  }
  double sf[3] = {0, 0, 0};
#pragma unroll
  for (int e = 0; e < NEE; ++e) {
    double pe[3], f[3];
    vec_eval(I.p[e], x, pe);
    vec_eval(I.f[e], x, f);
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

// element (i, d) of the cross-product matrix [v]x
__device__ __forceinline__ double skew_el(const double *v, int i, int d) {
  if (i == d) return 0.0;
  const int k = 3 - i - d;                       // the remaining axis
  const double sgn = ((d - i + 3) % 3 == 1) ? -1.0 : 1.0;   // [v]x[0][1] = -v2, [0][2] = +v1, ...
  return sgn * v[k];
}

__device__ inline void dyn_column(const DevPlan &P, const ColDesc &C, const double *loc_all, double *G) {
  const double *loc = loc_all + (size_t)C.inst * DYN_LOC;
  const int *pos = P.eq_pos + C.gbase;   // the block's rows land at their own stream positions
  const int d = C.dim, nc = C.ncol;
  double v[6];
  if (C.kind == 0) {        // base position / acceleration: d g_ang / d r = -[sum f]x ; d g_lin / d a = m I
    for (int i = 0; i < 3; ++i) v[i] = -skew_el(loc + 27, i, d) * C.w0;
    for (int i = 0; i < 3; ++i) v[3 + i] = i == d ? P.mass * C.w1 : 0.0;
  } else if (C.kind == 1) { // Euler angles / rates / accelerations
    for (int i = 0; i < 3; ++i)
      v[i] = loc[3 * i + d] * C.w0 + loc[9 + 3 * i + d] * C.w1 + loc[18 + 3 * i + d] * C.w2;
    for (int i = 0; i < 3; ++i) v[3 + i] = 0.0;
  } else if (C.kind < 6) {  // foot position: d g_ang / d p = [f]x
    const double *f = loc + 30 + 3 * (C.kind - 2);
    for (int i = 0; i < 3; ++i) v[i] = skew_el(f, i, d) * C.w0;
    for (int i = 0; i < 3; ++i) v[3 + i] = 0.0;
  } else {                  // foot force: d g_ang / d f = [r - p]x ; d g_lin / d f = -I
    const double *dd = loc + 42 + 3 * (C.kind - 6);
    for (int i = 0; i < 3; ++i) v[i] = skew_el(dd, i, d) * C.w0;
    for (int i = 0; i < 3; ++i) v[3 + i] = i == d ? -C.w0 : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) G[pos[i * nc]] = v[i];
}

template <bool JAC>
__device__ inline void eval_rom(const DevPlan &P, const RomInst &I, const double *x, double *g, double *loc) {
  double r[3], th[3], pe[3];
  vec_eval(I.r, x, r); vec_eval(I.th, x, th); vec_eval(I.p, x, pe);
  const double d[3] = {pe[0] - r[0], pe[1] - r[1], pe[2] - r[2]};
  double R[9];
  rotation<double>(th, R);
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
      rotation<D1>(t0, Rd);
#pragma unroll
      for (int i = 0; i < 3; ++i) loc[9 + 3 * i + j] = Rd[i].d * d[0] + Rd[3 + i].d * d[1] + Rd[6 + i].d * d[2];
    }
  }
}

__device__ inline void rom_column(const ColDesc &C, const double *loc_all, double *G) {
  const double *loc = loc_all + (size_t)C.inst * ROM_LOC;
  double *col = G + C.gbase;
  const int d = C.dim, nc = C.ncol;
  if (C.kind == 1) {
    for (int i = 0; i < 3; ++i) col[i * nc] = loc[9 + 3 * i + d] * C.w0;
  } else {
    const double sgn = C.kind == 2 ? 1.0 : -1.0;   // d/dp = R^T, d/dr = -R^T ; (R^T)[i][d] = R[3d + i]
    for (int i = 0; i < 3; ++i) col[i * nc] = sgn * loc[3 * d + i] * C.w0;
  }
}

template <bool JAC>
__device__ inline void eval_terr(const DevPlan &P, const TerrInst &I, int map, const double *x, double *g, double *Gp) {
  const Terr t = terrain_at(P, map, x[I.vx], x[I.vy]);
  g[I.row] = x[I.vz] - t.h;
  if (JAC && I.in_kkt) {
    if (P.row_kind[I.row] == 1) {   // stance row: equality block, entries at their stream positions
      const int *pos = P.eq_pos + I.goff;
      if (I.cx >= 0) Gp[pos[I.cx]] = -t.hx;
      if (I.cy >= 0) Gp[pos[I.cy]] = -t.hy;
      if (I.cz >= 0) Gp[pos[I.cz]] = 1.0;
    } else {
      double *G = Gp + I.goff;
      if (I.cx >= 0) G[I.cx] = -t.hx;
      if (I.cy >= 0) G[I.cy] = -t.hy;
      if (I.cz >= 0) G[I.cz] = 1.0;
    }
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
template <bool JAC>
__device__ inline void eval_all(const DevPlan &P, int map, const double *x, double *g, double *G, double *loc) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < P.n_dyn; i += nt) eval_dyn<JAC>(P, P.dyn[i], x, g, JAC ? loc + (size_t)i * DYN_LOC : nullptr);
  if (JAC) {
    __syncthreads();
    for (int c = tid; c < P.n_dyn_cols; c += nt) dyn_column(P, P.dyn_cols[c], loc, G);
    __syncthreads();
  }
  for (int i = tid; i < P.n_rom; i += nt) eval_rom<JAC>(P, P.rom[i], x, g, JAC ? loc + (size_t)i * ROM_LOC : nullptr);
  if (JAC) {
    __syncthreads();
    for (int c = tid; c < P.n_rom_cols; c += nt) rom_column(P.rom_cols[c], loc, G);
  }
  for (int i = tid; i < P.n_force; i += nt) eval_force<JAC>(P, P.force[i], map, x, g, G);
  for (int i = tid; i < P.n_terr; i += nt) eval_terr<JAC>(P, P.terr[i], map, x, g, G);
  for (int i = tid; i < P.n_lin; i += nt) {
    const LinRow &L = P.lin[i];
    double acc = 0;
    for (int k = 0; k < L.n; ++k) acc += L.coef[k] * x[L.var[k]];
    g[L.row] = acc;
  }
}

// ---- workgroup reductions (fixed tree => bitwise reproducible) --------------------------------
template <int OP>  // 0 sum, 1 max, 2 min
__device__ inline double wg_reduce(double v, double *scratch) {
  const int tid = threadIdx.x, nt = blockDim.x;
  __syncthreads();
  scratch[tid] = v;
  __syncthreads();
  for (int s = nt >> 1; s > 0; s >>= 1) {
    if (tid < s) {
      double a = scratch[tid], b = scratch[tid + s];
      scratch[tid] = OP == 0 ? a + b : (OP == 1 ? fmax(a, b) : fmin(a, b));
    }
    __syncthreads();
  }
  double r = scratch[0];
  __syncthreads();
  return r;
}

// barrier weights of every inequality row: sig = zl/(s-l) + zu/(u-s), w = sig (g - s) - mu/(s-l) + mu/(u-s)
__device__ inline void barrier_terms(const DevPlan &P, const double *g, const double *s, const double *zl,
                                     const double *zu, double mu, double *sig, double *w, double *stream) {
  for (int r = threadIdx.x; r < P.n_cons; r += blockDim.x) {
    if (P.row_kind[r] == 1) { stream[P.rhs_pos[r]] = -g[r]; continue; }
    if (P.row_kind[r] != 2) continue;
    const double l = P.con_lo[r], u = P.con_hi[r];
    const bool hl = l > -1e19, hu = u < 1e19;
    const double dl = hl ? s[r] - l : 1.0, du = hu ? u - s[r] : 1.0;
    const double sg = (hl ? zl[r] / dl : 0.0) + (hu ? zu[r] / du : 0.0);
    const double gmu = -(hl ? mu / dl : 0.0) + (hu ? mu / du : 0.0);
    const double wr = sg * (g[r] - s[r]) + gmu;
    sig[r] = sg;
    w[r] = wr;
    stream[P.sig_pos[r]] = sg;
    stream[P.w_pos[r]] = wr;
  }
}

// max violation of the working rows (viol) and of the slack form (theta)
__device__ inline void infeasibility(const DevPlan &P, const double *g, const double *s, double *scratch,
                                     double &viol, double &theta) {
  double v = 0, t = 0;
  for (int r = threadIdx.x; r < P.n_cons; r += blockDim.x) {
    const int k = P.row_kind[r];
    if (k == 0) continue;
    const double gr = g[r];
    if (!(gr == gr) || !(s[r] == s[r])) { v = t = INFINITY; continue; }   // fmax would swallow a NaN
    if (k == 1) { v = fmax(v, fabs(gr)); t = fmax(t, fabs(gr)); }
    else {
      v = fmax(v, fmax(P.con_lo[r] - gr, gr - P.con_hi[r]));
      t = fmax(t, fabs(gr - s[r]));
    }
  }
  viol = wg_reduce<1>(v, scratch);
  theta = wg_reduce<1>(t, scratch);
}

__device__ inline double l1_infeasibility(const DevPlan &P, const double *g, const double *s, const double *ds,
                                          double al, double *scratch) {
  double t = 0;
  for (int r = threadIdx.x; r < P.n_cons; r += blockDim.x) {
    const int k = P.row_kind[r];
    if (k == 1) t += fabs(g[r]);
    else if (k == 2) t += fabs(g[r] - (s[r] + al * ds[r]));
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

// Gathers this problem's linearisation into the stage-ordered stream k_kkt reads (one contiguous
// slice per stage).  Runs right after the values were produced, while they are hot in L2.
__device__ inline void pack_stream(const DevPlan &P, const double *G, const double *g, const double *sig,
                                   const double *w, double *stream) {
  for (int i = threadIdx.x; i < P.stream_len; i += blockDim.x) {
    const int s = P.pack_src[i], kind = s >> 28, idx = s & 0x0fffffff;
    double v;
    switch (kind) {
      case 0: v = G[idx]; break;
      case 1: v = P.g_static[idx]; break;
      case 2: v = -g[idx]; break;
      case 3: v = sig[idx]; break;
      case 4: v = w[idx]; break;
      default: v = P.piv_diag[idx]; break;
    }
    stream[i] = v;
  }
}

// =================================================================================================
__global__ __launch_bounds__(256) void k_start(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  __shared__ double scratch[256];
  extern __shared__ double evl[];
  const int n = P.n_vars, m = P.n_cons, tid = threadIdx.x;
  double *x = W.x + (size_t)b * n, *g = W.g + (size_t)b * m;
  double *s = W.s + (size_t)b * m, *zl = W.zl + (size_t)b * m, *zu = W.zu + (size_t)b * m;
  const double *st = W.start + (size_t)b * QTOS_START_DOUBLES, *gl = W.goal + (size_t)b * 3;
  const int map = W.map_id ? W.map_id[b] : 0;
  // end points of the linear-interpolation guess (towr nlp_formulation.cc Make*Variables)
  const double fin[3] = {gl[0], gl[1], terrain_at(P, map, gl[0], gl[1]).h - P.nominal[0][2]};
  for (int v = tid; v < n; v += blockDim.x) {
    const InitDesc I = P.init[v];
    double val;
    if (I.fix_src >= 0) {
      val = I.fix_src < 24 ? st[I.fix_src] : (I.fix_src < 26 ? gl[I.fix_src - 24] : 0.0);
    } else if (W.warm) {
      val = W.warm[(size_t)b * n + v];
    } else {
      double a, e;
      if (I.set == 0) { a = st[I.dim]; e = fin[I.dim]; }
      else if (I.set == 1) { a = st[3 + I.dim]; e = 0.0; }
      else if (I.set < 6) {
        const int ee = I.set - 2;
        a = st[6 + 3 * ee + I.dim];
        const double fx = fin[0] + P.nominal[ee][0], fy = fin[1] + P.nominal[ee][1];
        e = I.dim == 0 ? fx : (I.dim == 1 ? fy : terrain_at(P, map, fx, fy).h);
      } else { a = e = I.dim == 2 ? P.mass * P.gravity / NEE : 0.0; }
      val = I.is_vel ? (e - a) / P.T : a + I.frac * (e - a);
    }
    x[v] = val;
  }
  __syncthreads();
  eval_all<false>(P, map, x, g, nullptr, nullptr);
  __syncthreads();
  // slack initialisation: push strictly inside the bounds (Ipopt bound_push / bound_frac)
  for (int r = tid; r < m; r += blockDim.x) {
    if (P.row_kind[r] != 2) { s[r] = 0; zl[r] = 0; zu[r] = 0; continue; }
    const double l = P.con_lo[r], u = P.con_hi[r];
    const bool hl = l > -1e19, hu = u < 1e19;
    const double kp = W.warm ? 0.01 : P.slack_push;   // large push on cold starts, Ipopt's 0.01 on warm starts
    double pl = hl ? kp * fmax(1.0, fabs(l)) : 0.0, pu = hu ? kp * fmax(1.0, fabs(u)) : 0.0;
    if (hl && hu) { pl = fmin(pl, kp * (u - l)); pu = fmin(pu, kp * (u - l)); }
    double si = g[r];
    if (hl) si = fmax(si, l + pl);
    if (hu) si = fmin(si, u - pu);
    s[r] = si;
  }
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
    W.done[b] = (conv || bad) ? 1 : 0;
    if (!conv && !bad) atomicAdd(W.n_active, 1);
    record_trace(P, W, b, 0, viol, theta, 0.0, mu);
  }
  if (conv || bad) return;
  __syncthreads();
  eval_all<true>(P, map, x, g, W.stream + (size_t)b * P.stream_len, evl);
  __syncthreads();
  barrier_terms(P, g, s, zl, zu, mu, W.sig + (size_t)b * m, W.w + (size_t)b * m, W.stream + (size_t)b * P.stream_len);
}

// =================================================================================================
// k_kkt: one workgroup (KT threads) per problem.
//   LDS: A   lower triangle of the symmetric front, rows 0..F; row F is the right-hand side
//        Pn[(F+1) x 17] pivot columns (border panel C; pivot rows zeroed), Wn = Pn * Binv
constexpr int KT = 512;
constexpr int PLD = PIV + 1;
__device__ __forceinline__ int tri(int r, int c) { return ((r * (r + 1)) >> 1) + c; }  // c <= r
__device__ __forceinline__ int trs(int a, int b) { return a >= b ? tri(a, b) : tri(b, a); }

// Workgroup barrier that waits for LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would
// stall every phase on the in-flight prefetch loads and factor-panel stores (cdna_hip_programming.md
// section 5 "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
typedef double d4_t __attribute__((ext_vector_type(4)));

// In-register LDL^T of a symmetric 16 x 16 block by one wave, no pivoting (the KKT matrix is
// quasi-definite and the elimination order is fixed).  Lane i (mod 16) holds ROW i in 16
// registers.  Step k: every lane scales its own B[i][k] by 1/d_k (no cross-lane traffic), and row
// k -- wave-uniform -- is read from lane k with v_readlane; no LDS round trips on the 15-step
// dependency chain.  Reciprocals: v_rcp_f64 + two Newton steps instead of the IEEE division
// sequence (which costs hundreds of cycles per step on the critical path).
__device__ __forceinline__ double readlane_d(double v, int lane) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), lane);
  const unsigned hi = __builtin_amdgcn_readlane((int)(u >> 32), lane);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
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
// c += bcast_K(a) * b as ONE DP-ALU DPP instruction (the f64 FMA reads its first operand from lane K
// of the 16-lane row; v_fmac_f64 is the VOP2 form that can carry DPP); callers keep two instructions between a VALU write of `a` and this read.
template <int K>
__device__ __forceinline__ void fma_bc(double &c, double a, double b) {
  asm volatile("v_fmac_f64 %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(b), "n"(K));
}
// a += bcast_K(a) * b (the broadcast source is the accumulator itself)
template <int K>
__device__ __forceinline__ void fma_bc_self(double &a, double b) {
  asm volatile("v_fmac_f64 %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "n"(K));
}
template <int K, int J, int END>
__device__ __forceinline__ void bc_update(double (&r)[PIV], double m) {
  if constexpr (J < END) {
    fma_bc_self<K>(r[J], m);
    bc_update<K, J + 1, END>(r, m);
  }
}
template <int K>
__device__ __forceinline__ void ldlt16_steps(double (&a)[PIV], double (&v)[PIV], double &myinv, int i) {
  if constexpr (K < PIV) {
    const double inv = fast_rcp(bc16<K>(a[K]));   // 1 / d_K
    if (i == K) myinv = inv;
    if constexpr (K < PIV - 1) {
      const double li = i > K ? a[K] * inv : 0.0;  // L[i][K] for rows i > K, 0 for finished rows
      const double nli = -li;
      bc_update<K, K + 1, PIV>(a, nli);   // trailing block: a[j] -= L[i][K] B[K][j], j > K
      bc_update<K, 0, K>(v, nli);         // inverse: row i -= L[i][K] * (row K of L^-1), columns < K
      v[K] = nli;                         // (L^-1)[K][K] = 1
      asm volatile("s_nop 1" : "+v"(v[K]));   // DPP hazard: two wait states between this VALU write and the row read
      if (i > K) a[K] = li;
    }
    ldlt16_steps<K + 1>(a, v, myinv, i);
  }
}
// In-register LDL^T of a symmetric 16 x 16 block by one wave, no pivoting, plus the inverse of the
// unit lower-triangular factor (stage updates then become plain matrix products on the matrix
// cores).  Lane i of every 16-lane row holds ROW i of the block; the wave-uniform row K is read
// with DPP row broadcasts folded into the f64 FMAs: no LDS traffic and no scalar round trips on
// the 16-step dependency chain.  Lanes 0..15 write L (strictly lower part valid), lanes 16..31
// write L^-1 (full rows, zeros above the unit diagonal).
__device__ __forceinline__ void ldlt16(const double *Bsrc /* 16 x PLD */, double *Lm, double *Li, double *dinv, int lane) {
  const int i = lane & 15;
  double a[PIV], v[PIV];
#pragma unroll
  for (int j = 0; j < PIV; ++j) { a[j] = Bsrc[i * PLD + j]; v[j] = 0.0; }
  double myinv = 0.0;
  // pin the 32 row registers before the DPP chain starts (EXEC / VALU-write hazards of the hand-written
  // DPP instructions are not tracked by the compiler)
#define QTOS_PIN16(r) asm volatile("s_nop 4" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]))
  QTOS_PIN16(a);
  QTOS_PIN16(v);
#undef QTOS_PIN16
  ldlt16_steps<0>(a, v, myinv, i);
  if (lane < 2 * PIV) {
    double *dst = (lane < PIV ? Lm : Li) + i * PLD;
#pragma unroll
    for (int j = 0; j < PIV; ++j) dst[j] = lane < PIV ? a[j] : (j == i ? 1.0 : v[j]);
    if (lane < PIV) dinv[i] = myinv;
  }
}

#ifdef QTOS_STAMPS
#define STAMP2(i) do { if (tid == 64) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps2[i] += t_ - tlast2; tlast2 = t_; } } while (0)
#define STAMP(i) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMP2(i) do {} while (0)
#endif

constexpr int PFD = 6, PFS = 8;  // per-thread prefetch registers: doubles / ints of a stage's records

// assembly of one stage's records (already in LDS) into the front.  Pass 1: pivot diagonals,
// equality entries, multiplier right-hand sides (all distinct targets).  Pass 2: the inequality
// blocks through the gather table: one thread per target entry sums its contributions
// (sum_r sig_r G[r][a] G[r][c], or -sum_r G[r][a] w_r for the rhs) in a fixed order.
constexpr int SHDR = 8;   // static record header ints
__device__ inline void assemble_stage(double *A, int F, const int *sbuf, const double *dbuf, int tid) {
  const int n_ent = sbuf[0], n_rhs = sbuf[1], n_iq = sbuf[2];
  const int *ps = sbuf + SHDR;
  if (tid < PIV) A[tri(ps[tid], ps[tid])] += dbuf[tid];
  const int *eidx = sbuf + SHDR + PIV;
  const double *eval = dbuf + PIV;
  for (int i = tid; i < n_ent; i += KT) A[eidx[i]] += eval[i];
  const int *rsl = eidx + n_ent;
  const double *rval = eval + n_ent;
  for (int i = tid; i < n_rhs; i += KT) A[tri(F, rsl[i])] += rval[i];
  if (n_iq == 0) return;
  const int *iqh = rsl + n_rhs;
  const int n_tgt = sbuf[5];
  const int *tg = sbuf + sbuf[4];                     // n_tgt + 1 ints: (tri << 12) | first contribution
  const unsigned short *cl = (const unsigned short *)(tg + n_tgt + 1);
  lds_barrier();   // a pivot diagonal / rhs entry above can also be a gather target
  for (int t = tid; t < n_tgt; t += KT) {
    const int tv = tg[t], c0 = tv & 4095, c1 = tg[t + 1] & 4095;
    double acc = 0;
    for (int j = c0; j < c1; ++j) {
      const int code = cl[j], q = code >> 12, a = (code >> 6) & 63, c = code & 63;
      const int qm = iqh[4 * q], qn = iqh[4 * q + 1];
      const double *Gb = dbuf + iqh[4 * q + 2];
      const double *sg = Gb + qm * qn, *wq = sg + qm;
      if (c == 63) { for (int r = 0; r < qm; ++r) acc -= Gb[r * qn + a] * wq[r]; }
      else { for (int r = 0; r < qm; ++r) acc += sg[r] * Gb[r * qn + a] * Gb[r * qn + c]; }
    }
    A[tv >> 12] += acc;
  }
}
// Forward substitution y <- y L^-T with FOUR lanes per row: lane c of a quad owns columns c, c+4,
// c+8, c+12.  Step Q broadcasts y[Q] inside the quad with a DPP quad_perm move (no LDS) and every
// lane updates its columns j > Q.  The first quad carries the right-hand-side row as a second row.
template <int SRC>
__device__ __forceinline__ double quad_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, SRC * 0x55, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, SRC * 0x55, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int Q>
__device__ __forceinline__ void ysolve_steps(double (&y)[4], double (&yF)[4], const double *Lm, int cq) {
  if constexpr (Q < PIV - 1) {
    const double yq = quad_bcast<Q & 3>(y[Q >> 2]);
    const double yFq = quad_bcast<Q & 3>(yF[Q >> 2]);
#pragma unroll
    for (int jj = Q >> 2; jj < 4; ++jj) {
      const double l = Lm[(4 * jj + cq) * PLD + Q];
      const double m = (jj > (Q >> 2) || cq > (Q & 3)) ? l : 0.0;   // only columns j = 4 jj + cq > Q
      y[jj] -= yq * m;
      yF[jj] -= yFq * m;
    }
    ysolve_steps<Q + 1>(y, yF, Lm, cq);
  }
}

// k_kkt: one workgroup (KT threads, 8 waves) per problem; software-pipelined over the stage chain:
//   S1  Y_k = P_k L_k^-T (row per thread), retire the pivots of stage k
//   S2  assemble stage k+1 into the front (records prefetched one stage earlier)
//   S3  early gather: P_{k+1} = A[:, piv_{k+1}] - Y_k D_k^-1 Y_k[piv_{k+1}]^T  (the columns the NEXT
//       factorisation needs, with stage k's update applied on the fly)
//   S4  wave 0: LDL^T of the next pivot block   ||   waves 1-7: full Schur update of the front on
//       the f64 matrix cores + factor panel of stage k to HBM
__global__ __launch_bounds__(KT) void k_kkt(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x;
  if (b >= B || W.done[b]) return;
  extern __shared__ double lds[];
  const int F = P.front, tid = threadIdx.x;
  const int ntri = ((F + 1) * (F + 2)) >> 1;
  const int PSZ = (F + 1) * PLD;
  double *A = lds;                      // lower triangle incl. rhs row F
  double *Pbuf = A + ntri;              // 2 panels of (F+1) x PLD
  double *Lbuf = Pbuf + 2 * PSZ;        // 2 x (PIV x PLD)
  double *Libuf = Lbuf + 2 * PIV * PLD; // 2 x (PIV x PLD)  inverse of the unit lower factor
  double *dvb = Libuf + 2 * PIV * PLD;  // 2 x PIV   (1/d)
  double *xs = dvb + 2 * PIV;           // 128 (solution by front slot, backward pass)
  double *red = xs + 128;               // 34 x PLD  scratch
  double *dbuf = red + 34 * PLD;        // max_drec
  int *sbuf = (int *)(dbuf + P.max_drec);   // max_srec
  int *soff = sbuf + P.max_srec;        // n_stages + 1
  int *doff = soff + P.n_stages + 1;    // n_stages + 1
  int *psb = doff + P.n_stages + 1;     // 2 x PIV pivot slots (current / next stage)
  int *hib = psb + 2 * PIV;             // 2: hi of current / next stage
  int *tileRC = hib + 2;                // (R << 8) | C of lower-triangular tile t, t < 45
  const int n = P.n_vars, NS = P.n_stages;
  const double *stream = W.stream + (size_t)b * P.stream_len;
  double *panel = W.panel + (size_t)b * P.panel_stride;
  double *dx = W.dx + (size_t)b * n;
  const int pstride = (F + PIV + 4) * PIV;  // per stage: L^-1 (16 x 16), 1/d (16), y_F (16), pivot slots (16), hi (+15 pad), Y (hi x 16)

  for (int i = tid; i < ntri; i += KT) A[i] = 0.0;
  for (int v = tid; v < n; v += KT) dx[v] = 0.0;
  for (int i = tid; i <= NS; i += KT) { soff[i] = P.srec_off[i]; doff[i] = P.drec_off[i]; }
  if (tid < 45) {
    int R = 0;
    while (((R + 1) * (R + 2)) >> 1 <= tid) ++R;
    tileRC[tid] = (R << 8) | (tid - ((R * (R + 1)) >> 1));
  }
  __syncthreads();
  // ---- prologue: assemble stage 0, gather and factor its pivot block, stage records of stage 1 ----
  for (int i = tid; i < soff[1] - soff[0]; i += KT) sbuf[i] = P.srec[soff[0] + i];
  for (int i = tid; i < doff[1] - doff[0]; i += KT) dbuf[i] = stream[doff[0] + i];
  __syncthreads();
  if (tid < PIV) psb[tid] = sbuf[SHDR + tid];
  if (tid == 0) hib[0] = sbuf[3];
  assemble_stage(A, F, sbuf, dbuf, tid);
  __syncthreads();
  for (int i = tid; i < (F + 1) * PIV; i += KT) {
    const int r = i >> 4, j = i & 15;
    Pbuf[r * PLD + j] = A[trs(r, psb[j])];
  }
  if (NS > 1) {
    for (int i = tid; i < soff[2] - soff[1]; i += KT) sbuf[i] = P.srec[soff[1] + i];
    for (int i = tid; i < doff[2] - doff[1]; i += KT) dbuf[i] = stream[doff[1] + i];
  }
  __syncthreads();
  if (tid < 64) {
    double *Bs = red;  // 16 x PLD scratch
    for (int e = tid; e < PIV * PIV; e += 64) Bs[(e >> 4) * PLD + (e & 15)] = Pbuf[psb[e >> 4] * PLD + (e & 15)];
    ldlt16(Bs, Lbuf, Libuf, dvb, tid);
    for (int e = tid; e < PIV * PIV; e += 64) Pbuf[psb[e >> 4] * PLD + (e & 15)] = 0.0;  // pivot rows leave the panel
  }
  __syncthreads();

#ifdef QTOS_STAMPS
  unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  unsigned long long stamps2[4] = {0, 0, 0, 0}, tlast2 = 0;
  if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast) :: "memory");
#endif
  int cur = 0;
  for (int k = 0; k < NS; ++k) {
    const int nxt = cur ^ 1;
    double *Y = Pbuf + cur * PSZ, *Pn = Pbuf + nxt * PSZ;
    const double *Lm = Lbuf + cur * PIV * PLD, *dinv = dvb + cur * PIV;
    const int *ps = psb + cur * PIV;
    const int hi = hib[cur], hi16 = (hi + 15) & ~15;
    const bool has_next = k + 1 < NS;
    // prefetch the records of stage k+2 (installed at the end of this stage)
    double pfd[PFD];
    int pfs[PFS];
    int nd = 0, ns = 0;
    if (k + 2 < NS) {
      nd = doff[k + 3] - doff[k + 2];
      ns = soff[k + 3] - soff[k + 2];
      const double *dsrc = stream + doff[k + 2];
      const int *ssrc = P.srec + soff[k + 2];
#pragma unroll
      for (int j = 0; j < PFD; ++j) { const int i = tid + j * KT; pfd[j] = i < nd ? dsrc[i] : 0.0; }
#pragma unroll
      for (int j = 0; j < PFS; ++j) { const int i = tid + j * KT; pfs[j] = i < ns ? ssrc[i] : 0; }
    }
    // ---- S1: retire the pivots of stage k (their rows / columns are recycled), then Y = P L^-T with
    //      four lanes per row (rows 0..F-1 on quads 0..F-1, the rhs row F rides on quad 0) -------
    for (int i = tid; i < PIV * (hi + 1); i += KT) {
      const int j = i & 15, rr = i >> 4;
      A[trs(rr < hi ? rr : F, ps[j])] = 0.0;
    }
    {
      const int r0 = tid >> 2, cq = tid & 3;
      if (r0 < F) {
        double y[4], yF[4];
        double *row = Y + r0 * PLD, *rowF = Y + F * PLD;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { y[jj] = row[4 * jj + cq]; yF[jj] = r0 == 0 ? rowF[4 * jj + cq] : 0.0; }
        ysolve_steps<0>(y, yF, Lm, cq);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { row[4 * jj + cq] = y[jj]; if (r0 == 0) rowF[4 * jj + cq] = yF[jj]; }
      }
    }
    lds_barrier();
    STAMP(0);
    // ---- S2: assemble stage k+1 ------------------------------------------------------------------
    if (has_next) {
      if (tid < PIV) psb[nxt * PIV + tid] = sbuf[SHDR + tid];
      if (tid == 0) hib[nxt] = sbuf[3];
      assemble_stage(A, F, sbuf, dbuf, tid);
    }
    lds_barrier();
    STAMP(1);
    // ---- S3: early gather of the next pivot columns with this stage's update applied:
    //      P_next[r][j] = A[r][piv_j] - sum_q Y[r][q] (Y[piv_j][q] / d_q), 16-row tiles on the matrix
    //      cores (one tile per wave), the rhs row by 16 lanes -----------------------------------
    if (has_next) {
      const int *psn = psb + nxt * PIV;
      const int wv = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
      const int pcol = psn[li];
      const double *zrow = Y + pcol * PLD + lk;   // B operand: Z[k][j] = Y[piv_j][k] / d_k
      double zb[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) zb[s4] = zrow[4 * s4] * dinv[lk + 4 * s4];
      for (int R = wv; R < (F >> 4); R += KT / 64) {
        d4_t acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = A[trs(16 * R + lk + 4 * g, pcol)];
        const double *yrow = Y + (16 * R + li) * PLD + lk;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-yrow[4 * s4], zb[s4], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) Pn[(16 * R + lk + 4 * g) * PLD + li] = acc[g];
      }
      if (tid >= KT - PIV * PIV) {  // rhs row F: lane (j, q) forms one product, 16-lane shuffle tree sums over q
        const int t = tid - (KT - PIV * PIV), j = t >> 4, q = t & 15, pc = psn[j];
        double v = Y[F * PLD + q] * dinv[q] * Y[pc * PLD + q];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        if (q == 0) Pn[F * PLD + j] = A[tri(F, pc)] - v;
      }
    }
    lds_barrier();
    STAMP(2);
    // ---- S4: wave 0 factors the next pivot block; waves 1..7 apply stage k's update ------------
    if (tid < 64) {
      if (has_next) {
        const int *psn = psb + nxt * PIV;
        double *Bs = red;
        for (int e = tid; e < PIV * PIV; e += 64) Bs[(e >> 4) * PLD + (e & 15)] = Pn[psn[e >> 4] * PLD + (e & 15)];
        ldlt16(Bs, Lbuf + nxt * PIV * PLD, Libuf + nxt * PIV * PLD, dvb + nxt * PIV, tid);
        for (int e = tid; e < PIV * PIV; e += 64) Pn[psn[e >> 4] * PLD + (e & 15)] = 0.0;
      }
      STAMP(3);
    } else {
#ifdef QTOS_STAMPS
      if (tid == 64) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast2) :: "memory");
#endif
      const int wv = (tid >> 6) - 1, lane = tid & 63, li = lane & 15, lk = lane >> 4;
      const int nt16 = hi16 >> 4, ntile = (nt16 * (nt16 + 1)) >> 1;
      double dv4[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) dv4[s4] = -dinv[lk + 4 * s4];
      // two tiles in flight per wave: the second tile's LDS reads overlap the first tile's MFMA chain
      for (int t = wv; t < ntile; t += 2 * (KT / 64 - 1)) {
        const int t1 = t + (KT / 64 - 1);
        const bool two = t1 < ntile;
        const int rc0 = tileRC[t], rc1 = tileRC[two ? t1 : t];
        const int R0 = rc0 >> 8, C0 = rc0 & 255, R1 = rc1 >> 8, C1 = rc1 & 255;
        const int col0 = 16 * C0 + li, col1 = 16 * C1 + li;
        const double *w0 = Y + (16 * R0 + li) * PLD + lk, *p0 = Y + (16 * C0 + li) * PLD + lk;
        const double *w1 = Y + (16 * R1 + li) * PLD + lk, *p1 = Y + (16 * C1 + li) * PLD + lk;
        double wa0[4], pb0[4], wa1[4], pb1[4];
        int i0[4], i1[4];
        d4_t a0, a1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r0 = 16 * R0 + lk + 4 * g, r1 = 16 * R1 + lk + 4 * g;
          i0[g] = tri(r0, min(col0, r0));   // masked elements (col > row, diagonal tiles) read a valid
          i1[g] = tri(r1, min(col1, r1));   // dummy and are never written back
          a0[g] = A[i0[g]];
          a1[g] = A[i1[g]];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          wa0[s4] = w0[4 * s4] * dv4[s4]; pb0[s4] = p0[4 * s4];
          wa1[s4] = w1[4 * s4] * dv4[s4]; pb1[s4] = p1[4 * s4];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa0[s4], pb0[s4], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa1[s4], pb1[s4], a1, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (col0 <= 16 * R0 + lk + 4 * g) A[i0[g]] = a0[g];
          if (two && col1 <= 16 * R1 + lk + 4 * g) A[i1[g]] = a1[g];
        }
      }
      STAMP2(0);
      const int t2 = tid - 64;
      {  // right-hand-side row: 4 lanes per column, each sums 4 of the 16 products
        const int q4 = (t2 & 3) * 4;
        for (int c = t2 >> 2; c < hi16; c += (KT - 64) >> 2) {   // uniform within each 4-lane group
          double acc = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc += Y[F * PLD + q4 + q] * dinv[q4 + q] * Y[c * PLD + q4 + q];
          acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
          if ((t2 & 3) == 0) A[tri(F, c)] -= acc;
        }
      }
      STAMP2(1);
      // factor panel of stage k to HBM (everything the backward pass needs, so that it never has to
      // touch the static tables again)
      double *pk = panel + (size_t)k * pstride;
      for (int i = t2; i < PIV * PIV; i += KT - 64) pk[i] = Libuf[cur * PIV * PLD + (i >> 4) * PLD + (i & 15)];
      if (t2 < PIV) pk[PIV * PIV + t2] = dinv[t2];
      else if (t2 < 2 * PIV) pk[PIV * PIV + t2] = Y[F * PLD + (t2 - PIV)];
      else if (t2 < 3 * PIV) pk[PIV * PIV + t2] = (double)ps[t2 - 2 * PIV];
      else if (t2 == 3 * PIV) pk[PIV * PIV + t2] = (double)hi;
      for (int i = t2; i < hi * PIV; i += KT - 64) pk[PIV * PIV + 4 * PIV + i] = Y[(i >> 4) * PLD + (i & 15)];
      STAMP2(2);
    }
    lds_barrier();
    STAMP(4);
    // ---- install the records of stage k+2 ------------------------------------------------------
    if (k + 2 < NS) {
#pragma unroll
      for (int j = 0; j < PFD; ++j) { const int i = tid + j * KT; if (i < nd) dbuf[i] = pfd[j]; }
#pragma unroll
      for (int j = 0; j < PFS; ++j) { const int i = tid + j * KT; if (i < ns) sbuf[i] = pfs[j]; }
    }
    cur = nxt;
    STAMP(5);
  }
  // ---- backward substitution: x1 = L^-T D^-1 (y_F - Y^T x2).  Stage k belongs to wave (NS-1-k) mod 8,
  //      which holds that stage's whole factor panel in registers (loaded eight stages ahead, so the
  //      HBM latency is off the chain); per stage one wave does a 128 x 16 dot-product sweep against
  //      the solution kept in LDS, a 4-way lane fold and a 16 x 16 product with L^-T via DPP row
  //      broadcasts.  One LDS-only barrier per stage hands the solution to the next wave.
  __syncthreads();  // drains the factor-panel stores: they are read back below
  STAMP(6);
  for (int i = tid; i < 128; i += KT) xs[i] = 0.0;
  {
    constexpr int RB = 32;   // rows per lane: F / 4 <= 32
    const int wv = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
    double yv[RB], lc[PIV], dv = 0, yf = 0, psl = 0;   // psl: pivot slot, converted at use (no early wait on the load)
    int unk = -1;
    auto load_panel = [&](int k) {
      const double *pk = panel + (size_t)k * pstride;
#pragma unroll
      for (int jj = 0; jj < PIV; ++jj) lc[jj] = pk[jj * PIV + j];          // column j of L^-1
      dv = pk[PIV * PIV + j];
      yf = pk[PIV * PIV + PIV + j];
      psl = pk[PIV * PIV + 2 * PIV + j];
      unk = P.piv_unknown[k * PIV + j];
      const double *yk = pk + PIV * PIV + 4 * PIV;
#pragma unroll
      for (int i = 0; i < RB; ++i) yv[i] = 4 * i < F ? yk[(q + 4 * i) * PIV + j] : 0.0;   // wave-uniform predicate
    };
    if (NS - 1 - wv >= 0) load_panel(NS - 1 - wv);
    __syncthreads();
    for (int bs = 0; bs < NS; ++bs) {
      const int k = NS - 1 - bs;
      if ((bs & 7) == wv) {
        double acc0 = 0, acc1 = 0;
#pragma unroll
        for (int i = 0; i < RB; i += 2) {   // xs has 128 entries; rows >= F carry yv = 0, xs = 0
          acc0 = fma(yv[i], xs[q + 4 * i], acc0);
          acc1 = fma(yv[i + 1], xs[q + 4 * i + 4], acc1);
        }
        double acc = acc0 + acc1;
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        double u = (yf - acc) * dv;        // u_j on every lane with (lane & 15) == j
        asm volatile("s_nop 4" : "+v"(u));
        double x0 = 0, x1 = 0;             // x_i = sum_jj (L^-1)[jj][i] u_jj
        fma_bc<0>(x0, u, lc[0]);   fma_bc<1>(x1, u, lc[1]);   fma_bc<2>(x0, u, lc[2]);   fma_bc<3>(x1, u, lc[3]);
        fma_bc<4>(x0, u, lc[4]);   fma_bc<5>(x1, u, lc[5]);   fma_bc<6>(x0, u, lc[6]);   fma_bc<7>(x1, u, lc[7]);
        fma_bc<8>(x0, u, lc[8]);   fma_bc<9>(x1, u, lc[9]);   fma_bc<10>(x0, u, lc[10]); fma_bc<11>(x1, u, lc[11]);
        fma_bc<12>(x0, u, lc[12]); fma_bc<13>(x1, u, lc[13]); fma_bc<14>(x0, u, lc[14]); fma_bc<15>(x1, u, lc[15]);
        const double xi = x0 + x1;
        if (lane < PIV) {
          xs[(int)psl] = xi;
          if (unk >= 0 && unk < n) dx[unk] = xi;
        }
        if (k - 8 >= 0) load_panel(k - 8);
      }
      lds_barrier();
    }
  }
#ifdef QTOS_STAMPS
  STAMP(7);
  if (tid == 0 && W.trace) for (int i = 0; i < 8; ++i) W.trace[((size_t)b * (P.max_iter + 1) + 30) * 4 + i] = (double)stamps[i];
  if (tid == 64 && W.trace) for (int i = 0; i < 4; ++i) W.trace[((size_t)b * (P.max_iter + 1) + 32) * 4 + i] = (double)stamps2[i];
#endif
}

// =================================================================================================
__global__ __launch_bounds__(256) void k_step(DevPlan P, DevWork W, int B, int it) {
  const int b = blockIdx.x;
  if (b >= B || W.done[b]) return;
  __shared__ double scratch[256];
  extern __shared__ double evl[];
  const int n = P.n_vars, m = P.n_cons, tid = threadIdx.x;
  double *x = W.x + (size_t)b * n, *xt = W.xt + (size_t)b * n, *dx = W.dx + (size_t)b * n;
  double *g = W.g + (size_t)b * m, *gt = W.gt + (size_t)b * m;
  double *s = W.s + (size_t)b * m, *zl = W.zl + (size_t)b * m, *zu = W.zu + (size_t)b * m;
  double *ds = W.ds + (size_t)b * m, *dzl = W.dzl + (size_t)b * m, *dzu = W.dzu + (size_t)b * m;
  const double *G = W.stream + (size_t)b * P.stream_len;
  const int map = W.map_id ? W.map_id[b] : 0;
  double mu = W.mu[b];
  // ds = Ji dx + (g - s) through the inequality blocks
  for (int bi = tid; bi < P.n_blocks; bi += blockDim.x) {
    const Block blk = P.blocks[bi];
    if (blk.kind != 1) continue;
    const double *Gb = G + blk.goff;
    for (int r = 0; r < blk.m; ++r) {
      double acc = 0;
      for (int a = 0; a < blk.n; ++a) acc += Gb[r * blk.n + a] * dx[P.block_cols[blk.col_off + a]];
      const int row = blk.row0 + r;
      ds[row] = acc + (g[row] - s[row]);
    }
  }
  __syncthreads();
  const double tau = fmax(0.99, 1.0 - mu);
  double amax = 1.0, az = 1.0;
  for (int r = tid; r < m; r += blockDim.x) {
    if (P.row_kind[r] != 2) continue;
    const double l = P.con_lo[r], u = P.con_hi[r];
    const bool hl = l > -1e19, hu = u < 1e19;
    const double dl = hl ? s[r] - l : 1.0, du = hu ? u - s[r] : 1.0;
    const double d = ds[r];
    const double a = hl ? mu / dl - zl[r] - zl[r] / dl * d : 0.0;
    const double c = hu ? mu / du - zu[r] + zu[r] / du * d : 0.0;
    dzl[r] = a;
    dzu[r] = c;
    if (hl && d < 0) amax = fmin(amax, tau * dl / -d);
    if (hu && d > 0) amax = fmin(amax, tau * du / d);
    if (hl && a < 0) az = fmin(az, tau * zl[r] / -a);
    if (hu && c < 0) az = fmin(az, tau * zu[r] / -c);
  }
  amax = wg_reduce<2>(amax, scratch);
  az = wg_reduce<2>(az, scratch);
  const double th0 = l1_infeasibility(P, g, s, ds, 0.0, scratch);
  // backtracking on the l1 infeasibility of (c_E, c_I - s)
  double al = amax, th = 0;
  for (int ls = 0; ls < 6; ++ls) {
    for (int v = tid; v < n; v += blockDim.x) xt[v] = x[v] + al * dx[v];
    __syncthreads();
    eval_all<false>(P, map, xt, gt, nullptr, nullptr);
    __syncthreads();
    th = l1_infeasibility(P, gt, s, ds, al, scratch);
    if (th <= (1.0 - 1e-4 * al) * th0 || th < 1e-9) break;
    if (ls < 5) al *= 0.5;
  }
  for (int v = tid; v < n; v += blockDim.x) x[v] = xt[v];
  for (int r = tid; r < m; r += blockDim.x) {
    g[r] = gt[r];
    if (P.row_kind[r] != 2) continue;
    const double l = P.con_lo[r], u = P.con_hi[r];
    const bool hl = l > -1e19, hu = u < 1e19;
    const double sn = s[r] + al * ds[r];
    s[r] = sn;
    double a = zl[r] + az * dzl[r], c = zu[r] + az * dzu[r];
    const double kap = 1e10;
    if (hl) a = fmin(fmax(a, mu / (kap * (sn - l))), kap * mu / (sn - l));
    if (hu) c = fmin(fmax(c, mu / (kap * (u - sn))), kap * mu / (u - sn));
    zl[r] = a;
    zu[r] = c;
  }
  if (al > 0.3) mu = fmax(P.mu_min, 0.2 * mu);
  __syncthreads();
  double viol, theta;
  infeasibility(P, g, s, scratch, viol, theta);
  const bool conv = viol <= P.tol && theta <= P.tol;
  const bool bad = !(viol < INFINITY) || !(th < INFINITY);
  if (tid == 0) {
    W.mu[b] = mu;
    W.viol[b] = viol;
    W.iters[b] = it + 1;
    record_trace(P, W, b, it + 1, viol, theta, al, mu);
    if (conv || bad) {
      W.status[b] = conv ? 0 : 2;
      W.done[b] = 1;
      atomicAdd(W.n_active, -1);
    }
  }
  if (conv || bad) return;
  __syncthreads();
  eval_all<true>(P, map, x, g, W.stream + (size_t)b * P.stream_len, evl);
  __syncthreads();
  barrier_terms(P, g, s, zl, zu, mu, W.sig + (size_t)b * m, W.w + (size_t)b * m, W.stream + (size_t)b * P.stream_len);
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
