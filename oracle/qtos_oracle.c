/*
 * qtos_oracle.c -- see qtos_oracle.h.  TEST INFRASTRUCTURE ONLY (checker + cpu_baseline).
 *
 * Restates, in plain C, the NLP behind `docker exec <id> ./main <flags>`
 * (reference call sites: scripts/main.py:49-50,90-91; QTOS/generateHeightField.py:385-386;
 *  flag marshalling QTOS/utils.py:26,644-670).  The solver source itself is not in the reference
 * tree; structure is taken from logs/towr_log.out:99-129 (variable/constraint sets, sizes and
 * order) and the formulas from the published towr v1.4 (file names quoted per function below are
 * upstream towr paths, "UPSTREAM" = ethz-adrl/towr v1.4).
 */
#include "qtos_oracle.h"

#include <complex.h>
#include <malloc.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define INF_B 1e20
typedef double complex cplx;

/* ============================================================================================ */
/* Node-spline machinery                                                                        */
/* UPSTREAM towr/src/spline.cc, polynomial.cc (CubicHermitePolynomial), nodes_variables*.cc     */
/* ============================================================================================ */
typedef struct {
  int n_polys;
  double dur[QO_MAX_POLYS];
  int idx[QO_MAX_NODES][2][3]; /* [node][0=pos,1=vel][dim] -> variable index, -1 = constant 0 */
} qo_spline;

typedef struct {
  qo_layout L;
  qo_spline lin, ang, eem[QO_NEE], eef[QO_NEE];
  /* per foot: variable index of stance s position (dim 0) and force-node bookkeeping */
  int n_stance[QO_NEE], n_swing[QO_NEE];
  int n_fnodes[QO_NEE];               /* optimised force nodes                     */
  int fnode_id[QO_NEE][QO_MAX_NODES]; /* spline node id of optimised force node j  */
  int fnode_stance[QO_NEE][QO_MAX_NODES];
  int n_dyn, n_rom;
  double t_dyn[QO_MAX_POLYS * 2 + 4], t_rom[QO_MAX_POLYS * 2 + 4];
} qo_model;

static int time_grid(double T, double dt, double *out) {
  /* UPSTREAM time_discretization_constraint.cc: 0, dt, 2dt, ... (floor(T/dt) steps), then T */
  int n = 0;
  double t = 0.0;
  out[n++] = t;
  int steps = (int)floor(T / dt);
  for (int i = 0; i < steps; ++i) {
    t += dt;
    out[n++] = t;
  }
  out[n++] = T;
  return n;
}

static int build_model(const qo_params *p, qo_model *M) {
  memset(M, 0, sizeof(*M));
  qo_layout *L = &M->L;
  double T = 0;
  for (int k = 0; k < p->n_phases[0]; ++k) T += p->phase_dur[0][k];
  L->T = T;
  /* base polynomials: UPSTREAM parameters.cc GetBasePolyDurations */
  int nb = 0;
  {
    double t_left = T, eps = 1e-10;
    while (t_left > eps) {
      if (nb >= QO_MAX_POLYS) return -1;
      M->lin.dur[nb] = t_left > p->dt_base ? p->dt_base : t_left;
      nb++;
      t_left -= p->dt_base;
    }
  }
  M->lin.n_polys = M->ang.n_polys = nb;
  memcpy(M->ang.dur, M->lin.dur, sizeof(double) * nb);
  L->n_base_nodes = nb + 1;
  L->off_lin = 0;
  L->off_ang = 6 * (nb + 1);
  for (int k = 0; k <= nb; ++k)
    for (int q = 0; q < 2; ++q)
      for (int d = 0; d < 3; ++d) {
        M->lin.idx[k][q][d] = L->off_lin + 6 * k + 3 * q + d;
        M->ang.idx[k][q][d] = L->off_ang + 6 * k + 3 * q + d;
      }
  int off = 12 * (nb + 1);
  /* ee motion: stance = 1 constant poly (3 vars), swing = 2 polys with one free mid node
   * (px,vx,py,vy,pz).  UPSTREAM nodes_variables_phase_based.cc (NodesVariablesEEMotion). */
  for (int e = 0; e < QO_NEE; ++e) {
    int P = p->n_phases[e];
    if (P < 1 || P > QO_MAX_PHASES || (P % 2) == 0) return -2;
    qo_spline *S = &M->eem[e];
    L->off_eem[e] = off;
    int np = 0, node = 0, v = off;
    for (int ph = 0; ph < P; ++ph) {
      double d = p->phase_dur[e][ph];
      if (ph % 2 == 0) { /* stance */
        for (int dd = 0; dd < 3; ++dd) {
          S->idx[node][0][dd] = v + dd;
          S->idx[node + 1][0][dd] = v + dd;
          S->idx[node][1][dd] = S->idx[node + 1][1][dd] = -1;
        }
        S->dur[np++] = d;
        node += 1;
        v += 3;
      } else { /* swing: nodes (node)=end of stance, (node+1)=mid, (node+2)=start of next stance */
        S->dur[np++] = d / 2;
        S->dur[np++] = d / 2;
        int mid = node + 1;
        S->idx[mid][0][0] = v + 0;
        S->idx[mid][1][0] = v + 1;
        S->idx[mid][0][1] = v + 2;
        S->idx[mid][1][1] = v + 3;
        S->idx[mid][0][2] = v + 4;
        S->idx[mid][1][2] = -1;
        node += 2;
        v += 5;
      }
    }
    S->n_polys = np;
    M->n_stance[e] = (P + 1) / 2;
    M->n_swing[e] = (P - 1) / 2;
    L->n_eem[e] = v - off;
    off = v;
  }
  /* ee force: stance = force_polys_per_stance polys, swing = 1 zero poly; nodes touching a swing
   * poly are constant zero.  UPSTREAM NodesVariablesEEForce. */
  for (int e = 0; e < QO_NEE; ++e) {
    int P = p->n_phases[e];
    qo_spline *S = &M->eef[e];
    L->off_eef[e] = off;
    int np = 0;
    int poly_swing[QO_MAX_POLYS], poly_stance_id[QO_MAX_POLYS];
    for (int ph = 0; ph < P; ++ph) {
      double d = p->phase_dur[e][ph];
      if (ph % 2 == 0) {
        for (int j = 0; j < p->force_polys_per_stance; ++j) {
          poly_swing[np] = 0;
          poly_stance_id[np] = ph / 2;
          S->dur[np++] = d / p->force_polys_per_stance;
        }
      } else {
        poly_swing[np] = 1;
        poly_stance_id[np] = -1;
        S->dur[np++] = d;
      }
    }
    S->n_polys = np;
    int v = off, nf = 0;
    for (int node = 0; node <= np; ++node) {
      int constant = 0;
      if (node > 0 && poly_swing[node - 1]) constant = 1;
      if (node < np && poly_swing[node]) constant = 1;
      for (int dd = 0; dd < 3; ++dd) {
        S->idx[node][0][dd] = constant ? -1 : v + 2 * dd;
        S->idx[node][1][dd] = constant ? -1 : v + 2 * dd + 1;
      }
      if (!constant) {
        M->fnode_id[e][nf] = node;
        M->fnode_stance[e][nf] = node < np ? poly_stance_id[node] : poly_stance_id[node - 1];
        nf++;
        v += 6;
      }
    }
    M->n_fnodes[e] = nf;
    L->n_eef[e] = v - off;
    off = v;
  }
  L->n_vars = off;
  /* constraint layout, order of logs/towr_log.out:112-129 */
  M->n_dyn = time_grid(T, p->dt_dyn, M->t_dyn);
  M->n_rom = time_grid(T, p->dt_rom, M->t_rom);
  L->n_dyn_times = M->n_dyn;
  L->n_rom_times = M->n_rom;
  int c = 0;
  for (int e = 0; e < QO_NEE; ++e) {
    L->off_terrain[e] = c;
    c += M->eem[e].n_polys; /* nodes 1..n */
  }
  L->off_dyn = c;
  c += 6 * M->n_dyn;
  L->off_acc_lin = c;
  c += 3 * (nb - 1);
  L->off_acc_ang = c;
  c += 3 * (nb - 1);
  for (int e = 0; e < QO_NEE; ++e) {
    L->off_rom[e] = c;
    c += 3 * M->n_rom;
  }
  for (int e = 0; e < QO_NEE; ++e) {
    L->off_force[e] = c;
    c += 5 * M->n_fnodes[e];
  }
  for (int e = 0; e < QO_NEE; ++e) {
    L->off_swing[e] = c;
    c += 4 * M->n_swing[e];
  }
  L->n_cons = c;
  return 0;
}

/* UPSTREAM spline.cc GetSegmentID: at junctions the PREVIOUS polynomial is returned */
static void locate(const qo_spline *S, double t, int *k, double *tau) {
  const double eps = 1e-10;
  double acc = 0;
  int id = S->n_polys - 1;
  for (int i = 0; i < S->n_polys; ++i) {
    acc += S->dur[i];
    if (acc >= t - eps) {
      id = i;
      break;
    }
  }
  double tl = t;
  for (int i = 0; i < id; ++i) tl -= S->dur[i];
  *k = id;
  *tau = tl;
}

/* d-th derivative weights of a cubic Hermite polynomial wrt (p0, v0, p1, v1).
 * UPSTREAM polynomial.cc CubicHermitePolynomial::UpdateCoeff / GetDerivativeWrt* */
static void hermite_w(double T, double t, int deriv, double w[4]) {
  double T2 = T * T, T3 = T2 * T, t2 = t * t, t3 = t2 * t;
  if (deriv == 0) {
    w[0] = 1 - 3 * t2 / T2 + 2 * t3 / T3;
    w[1] = t - 2 * t2 / T + t3 / T2;
    w[2] = 3 * t2 / T2 - 2 * t3 / T3;
    w[3] = -t2 / T + t3 / T2;
  } else if (deriv == 1) {
    w[0] = -6 * t / T2 + 6 * t2 / T3;
    w[1] = 1 - 4 * t / T + 3 * t2 / T2;
    w[2] = 6 * t / T2 - 6 * t2 / T3;
    w[3] = -2 * t / T + 3 * t2 / T2;
  } else {
    w[0] = -6 / T2 + 12 * t / T3;
    w[1] = -4 / T + 6 * t / T2;
    w[2] = 6 / T2 - 12 * t / T3;
    w[3] = -2 / T + 6 * t / T2;
  }
}

static double nodeval(const qo_spline *S, const double *x, int node, int q, int d) {
  int i = S->idx[node][q][d];
  return i < 0 ? 0.0 : x[i];
}

static void eval_poly(const qo_spline *S, const double *x, int k, double tau, int deriv,
                      double out[3]) {
  double w[4];
  hermite_w(S->dur[k], tau, deriv, w);
  for (int d = 0; d < 3; ++d)
    out[d] = w[0] * nodeval(S, x, k, 0, d) + w[1] * nodeval(S, x, k, 1, d) +
             w[2] * nodeval(S, x, k + 1, 0, d) + w[3] * nodeval(S, x, k + 1, 1, d);
}

static void eval_spline(const qo_spline *S, const double *x, double t, int deriv, double out[3]) {
  int k;
  double tau;
  locate(S, t, &k, &tau);
  eval_poly(S, x, k, tau, deriv, out);
}

/* J[r0+r][var] += A[r][d] * d(spline^(deriv)_d (t)) / d var, for r < m */
static void scatter_poly(double *J, int n, int r0, int m, const double *A /* m x 3 */,
                         const qo_spline *S, int k, double tau, int deriv, double scale) {
  double w[4];
  hermite_w(S->dur[k], tau, deriv, w);
  for (int a = 0; a < 4; ++a) {
    int node = k + (a >> 1), q = a & 1;
    for (int d = 0; d < 3; ++d) {
      int v = S->idx[node][q][d];
      if (v < 0) continue;
      for (int r = 0; r < m; ++r) J[(size_t)(r0 + r) * n + v] += scale * A[r * 3 + d] * w[a];
    }
  }
}
static void scatter(double *J, int n, int r0, int m, const double *A, const qo_spline *S, double t,
                    int deriv) {
  int k;
  double tau;
  locate(S, t, &k, &tau);
  scatter_poly(J, n, r0, m, A, S, k, tau, deriv, 1.0);
}

/* ============================================================================================ */
/* Terrain: height map on a regular grid (QTOS/generateHeightField.py:590-632 writes the file)  */
/* UPSTREAM height_map.cc (GetNormalizedBasis and derivatives)                                  */
/* ============================================================================================ */
typedef struct {
  double h, hx, hy, hxy;
} terr;

static terr terrain_at(const qo_params *p, double x, double y) {
  terr t = {0, 0, 0, 0};
  if (!p->height) return t;
  double fx = (x - p->hx0) / p->hcell, fy = (y - p->hy0) / p->hcell;
  double mx = p->hnx - 1, my = p->hny - 1;
  if (p->terrain_mode == 1) { /* nearest cell: ledges stay flat, steps are jumps */
    int jx = (int)floor(fx + 0.5), jy = (int)floor(fy + 0.5);
    if (jx < 0) jx = 0;
    if (jy < 0) jy = 0;
    if (jx > p->hnx - 1) jx = p->hnx - 1;
    if (jy > p->hny - 1) jy = p->hny - 1;
    t.h = p->height[jx * p->hny + jy];
    return t;
  }
  int cx = 0, cy = 0; /* clamped => zero gradient in that direction */
  if (fx <= 0) { fx = 0; cx = 1; }
  if (fx >= mx) { fx = mx; cx = 1; }
  if (fy <= 0) { fy = 0; cy = 1; }
  if (fy >= my) { fy = my; cy = 1; }
  int ix = (int)floor(fx), iy = (int)floor(fy);
  if (ix > p->hnx - 2) ix = p->hnx - 2;
  if (iy > p->hny - 2) iy = p->hny - 2;
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  double u = fx - ix, v = fy - iy;
  int ix1 = p->hnx > 1 ? ix + 1 : ix, iy1 = p->hny > 1 ? iy + 1 : iy;
  double h00 = p->height[ix * p->hny + iy], h10 = p->height[ix1 * p->hny + iy],
         h01 = p->height[ix * p->hny + iy1], h11 = p->height[ix1 * p->hny + iy1];
  t.h = h00 * (1 - u) * (1 - v) + h10 * u * (1 - v) + h01 * (1 - u) * v + h11 * u * v;
  double c = p->hcell;
  t.hx = cx ? 0 : ((h10 - h00) * (1 - v) + (h11 - h01) * v) / c;
  t.hy = cy ? 0 : ((h01 - h00) * (1 - u) + (h11 - h10) * u) / c;
  t.hxy = (cx || cy) ? 0 : (h11 - h10 - h01 + h00) / (c * c);
  return t;
}

double qo_terrain_height(const qo_params *p, double x, double y) { return terrain_at(p, x, y).h; }

/* normalised basis b (0 normal, 1 tangent1, 2 tangent2) and its derivative wrt x and y */
static void terrain_basis(const terr *t, int which, double b[3], double dbdx[3], double dbdy[3]) {
  double v[3], vx[3], vy[3];
  if (which == 0) {
    v[0] = -t->hx; v[1] = -t->hy; v[2] = 1;
    vx[0] = 0; vx[1] = -t->hxy; vx[2] = 0;      /* h_xx = 0 for a bilinear patch */
    vy[0] = -t->hxy; vy[1] = 0; vy[2] = 0;
  } else if (which == 1) {
    v[0] = 1; v[1] = 0; v[2] = t->hx;
    vx[0] = 0; vx[1] = 0; vx[2] = 0;
    vy[0] = 0; vy[1] = 0; vy[2] = t->hxy;
  } else {
    v[0] = 0; v[1] = 1; v[2] = t->hy;
    vx[0] = 0; vx[1] = 0; vx[2] = t->hxy;
    vy[0] = 0; vy[1] = 0; vy[2] = 0;
  }
  double nn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  for (int i = 0; i < 3; ++i) b[i] = v[i] / nn;
  double bx = b[0] * vx[0] + b[1] * vx[1] + b[2] * vx[2];
  double by = b[0] * vy[0] + b[1] * vy[1] + b[2] * vy[2];
  for (int i = 0; i < 3; ++i) {
    dbdx[i] = (vx[i] - b[i] * bx) / nn;
    dbdy[i] = (vy[i] - b[i] * by) / nn;
  }
}

/* ============================================================================================ */
/* Rigid-body kinematics in complex arithmetic (complex-step derivatives for the Jacobian)      */
/* UPSTREAM euler_converter.cc: GetRotationMatrixBaseToWorld, GetM, GetMdot;                    */
/*          single_rigid_body_dynamics.cc: GetDynamicViolation                                  */
/* ============================================================================================ */
static void rot_c(const cplx th[3], cplx R[9]) {
  cplx x = th[0], y = th[1], z = th[2];
  cplx cx = ccos(x), sx = csin(x), cy = ccos(y), sy = csin(y), cz = ccos(z), sz = csin(z);
  R[0] = cy * cz; R[1] = cz * sx * sy - cx * sz; R[2] = sx * sz + cx * cz * sy;
  R[3] = cy * sz; R[4] = cx * cz + sx * sy * sz; R[5] = cx * sy * sz - cz * sx;
  R[6] = -sy;     R[7] = cy * sx;                R[8] = cx * cy;
}

/* angular part: I_w wd + w x (I_w w), I_w = R Ib R^T, w = M thd, wd = Mdot thd + M thdd */
static void dyn_ang_c(const double Ib[9], const cplx th[3], const cplx thd[3], const cplx thdd[3],
                      cplx out[3]) {
  cplx y = th[1], z = th[2], yd = thd[1], zd = thd[2];
  cplx cy = ccos(y), sy = csin(y), cz = ccos(z), sz = csin(z);
  cplx Mm[9] = {cy * cz, -sz, 0, cy * sz, cz, 0, -sy, 0, 1};
  cplx Md[9] = {-cz * sy * yd - cy * sz * zd, -cz * zd, 0, cy * cz * zd - sy * sz * yd, -sz * zd, 0,
                -cy * yd, 0, 0};
  cplx w[3], wd[3];
  for (int i = 0; i < 3; ++i) {
    w[i] = 0;
    wd[i] = 0;
    for (int j = 0; j < 3; ++j) {
      w[i] += Mm[3 * i + j] * thd[j];
      wd[i] += Md[3 * i + j] * thd[j] + Mm[3 * i + j] * thdd[j];
    }
  }
  cplx R[9], RI[9], Iw[9];
  rot_c(th, R);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      RI[3 * i + j] = 0;
      for (int k = 0; k < 3; ++k) RI[3 * i + j] += R[3 * i + k] * Ib[3 * k + j];
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Iw[3 * i + j] = 0;
      for (int k = 0; k < 3; ++k) Iw[3 * i + j] += RI[3 * i + k] * R[3 * j + k];
    }
  cplx Iwd[3], Iww[3];
  for (int i = 0; i < 3; ++i) {
    Iwd[i] = 0;
    Iww[i] = 0;
    for (int j = 0; j < 3; ++j) {
      Iwd[i] += Iw[3 * i + j] * wd[j];
      Iww[i] += Iw[3 * i + j] * w[j];
    }
  }
  out[0] = Iwd[0] + w[1] * Iww[2] - w[2] * Iww[1];
  out[1] = Iwd[1] + w[2] * Iww[0] - w[0] * Iww[2];
  out[2] = Iwd[2] + w[0] * Iww[1] - w[1] * Iww[0];
}

/* range-of-motion vector R^T (p - r) */
static void rom_c(const cplx th[3], const double d[3], cplx out[3]) {
  cplx R[9];
  rot_c(th, R);
  for (int i = 0; i < 3; ++i) out[i] = R[0 + i] * d[0] + R[3 + i] * d[1] + R[6 + i] * d[2];
}

static void skew(const double a[3], double S[9]) {
  S[0] = 0; S[1] = -a[2]; S[2] = a[1];
  S[3] = a[2]; S[4] = 0; S[5] = -a[0];
  S[6] = -a[1]; S[7] = a[0]; S[8] = 0;
}

/* ============================================================================================ */
/* public: layout, bounds, initial guess                                                        */
/* ============================================================================================ */
int qo_get_layout(const qo_params *p, qo_layout *L) {
  qo_model *M = (qo_model *)malloc(sizeof(qo_model));
  int rc = build_model(p, M);
  if (rc == 0) *L = M->L;
  free(M);
  return rc;
}

/* UPSTREAM nlp_formulation.cc MakeBaseVariables / MakeEndeffectorVariables (Add*Bound calls);
 * parameters.cc bounds_final_* (lin pos {X,Y}; lin vel, ang pos, ang vel {X,Y,Z}). */
int qo_var_bounds(const qo_params *p, const qo_problem *q, double *lo, double *hi) {
  qo_model *M = (qo_model *)malloc(sizeof(qo_model));
  int rc = build_model(p, M);
  if (rc) { free(M); return rc; }
  const qo_layout *L = &M->L;
  for (int i = 0; i < L->n_vars; ++i) { lo[i] = -INF_B; hi[i] = INF_B; }
  int nl = L->n_base_nodes - 1;
#define FIX(i, v) do { lo[i] = (v); hi[i] = (v); } while (0)
  for (int d = 0; d < 3; ++d) {
    FIX(L->off_lin + d, q->s[d]);
    FIX(L->off_lin + 3 + d, q->s_vel[d]);
    FIX(L->off_ang + d, q->s_ang[d]);
    FIX(L->off_ang + 3 + d, q->s_ang_vel[d]);
    if (d < 2) FIX(L->off_lin + 6 * nl + d, q->g[d]);
    FIX(L->off_lin + 6 * nl + 3 + d, 0.0);
    FIX(L->off_ang + 6 * nl + d, 0.0);
    FIX(L->off_ang + 6 * nl + 3 + d, 0.0);
    for (int e = 0; e < QO_NEE; ++e) FIX(L->off_eem[e] + d, q->ee[e][d]);
  }
#undef FIX
  free(M);
  return 0;
}

int qo_con_bounds(const qo_params *p, double *lo, double *hi) {
  qo_model *M = (qo_model *)malloc(sizeof(qo_model));
  int rc = build_model(p, M);
  if (rc) { free(M); return rc; }
  const qo_layout *L = &M->L;
  for (int i = 0; i < L->n_cons; ++i) lo[i] = hi[i] = 0.0;
  for (int e = 0; e < QO_NEE; ++e) {
    /* UPSTREAM terrain_constraint.cc GetBounds: stance nodes = 0, swing nodes in [0, 1e20] */
    const qo_spline *S = &M->eem[e];
    for (int node = 1; node <= S->n_polys; ++node)
      if (S->idx[node][1][0] >= 0) hi[L->off_terrain[e] + node - 1] = INF_B;
    /* UPSTREAM range_of_motion_constraint.cc UpdateBoundsAtInstance */
    for (int k = 0; k < M->n_rom; ++k)
      for (int d = 0; d < 3; ++d) {
        lo[L->off_rom[e] + 3 * k + d] = p->nominal_stance[e][d] - p->max_dev[d];
        hi[L->off_rom[e] + 3 * k + d] = p->nominal_stance[e][d] + p->max_dev[d];
      }
    /* UPSTREAM force_constraint.cc GetBounds */
    for (int j = 0; j < M->n_fnodes[e]; ++j) {
      int r = L->off_force[e] + 5 * j;
      lo[r] = 0; hi[r] = p->f_max;
      lo[r + 1] = -INF_B; hi[r + 1] = 0;
      lo[r + 2] = 0; hi[r + 2] = INF_B;
      lo[r + 3] = -INF_B; hi[r + 3] = 0;
      lo[r + 4] = 0; hi[r + 4] = INF_B;
    }
  }
  free(M);
  return 0;
}

/* UPSTREAM nodes_variables.cc SetByLinearInterpolation: position by node id / (n_nodes-1),
 * velocity = (final - initial)/T; for stance phases the value of the LAST node sharing the
 * variable wins (GetValues loop order). */
static void lin_interp(const qo_spline *S, double *x, const double a[3], const double b[3],
                       double T) {
  int n_nodes = S->n_polys + 1;
  for (int node = 0; node < n_nodes; ++node)
    for (int d = 0; d < 3; ++d) {
      int ip = S->idx[node][0][d], iv = S->idx[node][1][d];
      if (ip >= 0) x[ip] = a[d] + (double)node / (double)(n_nodes - 1) * (b[d] - a[d]);
      if (iv >= 0) x[iv] = (b[d] - a[d]) / T;
    }
}

int qo_initial_guess(const qo_params *p, const qo_problem *q, double *x) {
  qo_model *M = (qo_model *)malloc(sizeof(qo_model));
  int rc = build_model(p, M);
  if (rc) { free(M); return rc; }
  const qo_layout *L = &M->L;
  double T = L->T;
  memset(x, 0, sizeof(double) * L->n_vars);
  double fin[3] = {q->g[0], q->g[1],
                   qo_terrain_height(p, q->g[0], q->g[1]) - p->nominal_stance[0][2]};
  lin_interp(&M->lin, x, q->s, fin, T);
  double zero[3] = {0, 0, 0};
  lin_interp(&M->ang, x, q->s_ang, zero, T);
  for (int e = 0; e < QO_NEE; ++e) {
    double fe[3] = {fin[0] + p->nominal_stance[e][0], fin[1] + p->nominal_stance[e][1], 0};
    fe[2] = qo_terrain_height(p, fe[0], fe[1]);
    lin_interp(&M->eem[e], x, q->ee[e], fe, T);
    double f[3] = {0, 0, p->mass * p->gravity / QO_NEE};
    lin_interp(&M->eef[e], x, f, f, T);
  }
  /* fixed variables sit on their bounds (Ipopt fixed_variable_treatment = make_parameter) */
  double *lo = (double *)malloc(sizeof(double) * 2 * L->n_vars), *hi = lo + L->n_vars;
  qo_var_bounds(p, q, lo, hi);
  for (int i = 0; i < L->n_vars; ++i)
    if (lo[i] == hi[i]) x[i] = lo[i];
  free(lo);
  free(M);
  return 0;
}

/* ============================================================================================ */
/* constraints and Jacobian                                                                     */
/* ============================================================================================ */
static int stance_var(const qo_model *M, int e, int s) { return M->L.off_eem[e] + 8 * s; }

/* zrp / zci: CSR pattern (free columns) of the Jacobian, or NULL.  With a pattern only ITS entries of J are cleared before the
 * evaluation -- the solver reads J through that pattern and nothing else (the dense clear is 30 MB per evaluation on the
 * 100-knot problem: at one problem per thread a many-core host spends its memory bandwidth on zeros; bench.py cpu_baseline). */
static void eval_all_z(const qo_params *p, const qo_model *M, const double *x, double *g, double *J, const int *zrp, const int *zci) {
  const qo_layout *L = &M->L;
  const int n = L->n_vars;
  const double hcs = 1e-30;
  if (J && !zrp) memset(J, 0, sizeof(double) * (size_t)L->n_cons * n);
  if (J && zrp)
    for (int r = 0; r < L->n_cons; ++r) {
      double *Jr = J + (size_t)r * n;
      for (int a = zrp[r]; a < zrp[r + 1]; ++a) Jr[zci[a]] = 0.0;
    }

  /* ---- terrain: z - h(x,y) at ee-motion nodes 1..N (UPSTREAM terrain_constraint.cc) -------- */
  for (int e = 0; e < QO_NEE; ++e) {
    const qo_spline *S = &M->eem[e];
    for (int node = 1; node <= S->n_polys; ++node) {
      int r = L->off_terrain[e] + node - 1;
      double px = nodeval(S, x, node, 0, 0), py = nodeval(S, x, node, 0, 1),
             pz = nodeval(S, x, node, 0, 2);
      terr t = terrain_at(p, px, py);
      if (g) g[r] = pz - t.h;
      if (J) {
        J[(size_t)r * n + S->idx[node][0][2]] += 1.0;
        J[(size_t)r * n + S->idx[node][0][0]] += -t.hx;
        J[(size_t)r * n + S->idx[node][0][1]] += -t.hy;
      }
    }
  }

  /* ---- dynamics (UPSTREAM dynamic_constraint.cc, single_rigid_body_dynamics.cc) ------------ */
  for (int k = 0; k < M->n_dyn; ++k) {
    double t = M->t_dyn[k];
    int r0 = L->off_dyn + 6 * k;
    double r[3], a[3], th[3], thd[3], thdd[3];
    eval_spline(&M->lin, x, t, 0, r);
    eval_spline(&M->lin, x, t, 2, a);
    eval_spline(&M->ang, x, t, 0, th);
    eval_spline(&M->ang, x, t, 1, thd);
    eval_spline(&M->ang, x, t, 2, thdd);
    cplx cth[3], cthd[3], cthdd[3], out[3];
    for (int i = 0; i < 3; ++i) { cth[i] = th[i]; cthd[i] = thd[i]; cthdd[i] = thdd[i]; }
    dyn_ang_c(p->Ib, cth, cthd, cthdd, out);
    double gang[3] = {creal(out[0]), creal(out[1]), creal(out[2])};
    double glin[3] = {p->mass * a[0], p->mass * a[1], p->mass * a[2] + p->mass * p->gravity};
    double A[18];
    if (J) {
      /* wrt Euler angles / rates / accelerations: complex step on the angular rows */
      for (int what = 0; what < 3; ++what) {
        memset(A, 0, sizeof(A));
        for (int j = 0; j < 3; ++j) {
          cplx a0[3], a1[3], a2[3];
          for (int i = 0; i < 3; ++i) { a0[i] = th[i]; a1[i] = thd[i]; a2[i] = thdd[i]; }
          if (what == 0) a0[j] += hcs * I;
          if (what == 1) a1[j] += hcs * I;
          if (what == 2) a2[j] += hcs * I;
          dyn_ang_c(p->Ib, a0, a1, a2, out);
          for (int i = 0; i < 3; ++i) A[i * 3 + j] = cimag(out[i]) / hcs;
        }
        scatter(J, n, r0, 3, A, &M->ang, t, what);
      }
      /* linear rows wrt base acceleration */
      memset(A, 0, sizeof(A));
      A[0] = A[4] = A[8] = p->mass;
      scatter(J, n, r0 + 3, 3, A, &M->lin, t, 2);
    }
    double Sf[9] = {0};
    for (int e = 0; e < QO_NEE; ++e) {
      double pe[3], f[3];
      eval_spline(&M->eem[e], x, t, 0, pe);
      eval_spline(&M->eef[e], x, t, 0, f);
      double d[3] = {r[0] - pe[0], r[1] - pe[1], r[2] - pe[2]};
      /* tau_sum += f x (r - p);  g_ang -= tau_sum;  g_lin -= f */
      gang[0] -= f[1] * d[2] - f[2] * d[1];
      gang[1] -= f[2] * d[0] - f[0] * d[2];
      gang[2] -= f[0] * d[1] - f[1] * d[0];
      for (int i = 0; i < 3; ++i) glin[i] -= f[i];
      if (J) {
        double Fx[9], Dx[9];
        skew(f, Fx);
        skew(d, Dx);
        for (int i = 0; i < 9; ++i) Sf[i] += Fx[i];
        scatter(J, n, r0, 3, Fx, &M->eem[e], t, 0);      /* d g_ang / d p_e = +[f]x */
        scatter(J, n, r0, 3, Dx, &M->eef[e], t, 0);      /* d g_ang / d f_e = +[d]x */
        double mI[9] = {-1, 0, 0, 0, -1, 0, 0, 0, -1};
        scatter(J, n, r0 + 3, 3, mI, &M->eef[e], t, 0);  /* d g_lin / d f_e = -I   */
      }
    }
    if (J) {
      for (int i = 0; i < 9; ++i) Sf[i] = -Sf[i];
      scatter(J, n, r0, 3, Sf, &M->lin, t, 0);           /* d g_ang / d r = -sum [f]x */
    }
    if (g)
      for (int i = 0; i < 3; ++i) { g[r0 + i] = gang[i]; g[r0 + 3 + i] = glin[i]; }
  }

  /* ---- acceleration continuity (UPSTREAM spline_acc_constraint.cc) -------------------------- */
  for (int which = 0; which < 2; ++which) {
    const qo_spline *S = which ? &M->ang : &M->lin;
    int r0 = which ? L->off_acc_ang : L->off_acc_lin;
    for (int j = 0; j + 1 < S->n_polys; ++j) {
      double ap[3], an[3];
      eval_poly(S, x, j, S->dur[j], 2, ap);
      eval_poly(S, x, j + 1, 0.0, 2, an);
      if (g)
        for (int d = 0; d < 3; ++d) g[r0 + 3 * j + d] = ap[d] - an[d];
      if (J) {
        double Id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        scatter_poly(J, n, r0 + 3 * j, 3, Id, S, j, S->dur[j], 2, 1.0);
        scatter_poly(J, n, r0 + 3 * j, 3, Id, S, j + 1, 0.0, 2, -1.0);
      }
    }
  }

  /* ---- range of motion (UPSTREAM range_of_motion_constraint.cc) ----------------------------- */
  for (int e = 0; e < QO_NEE; ++e)
    for (int k = 0; k < M->n_rom; ++k) {
      double t = M->t_rom[k];
      int r0 = L->off_rom[e] + 3 * k;
      double r[3], th[3], pe[3];
      eval_spline(&M->lin, x, t, 0, r);
      eval_spline(&M->ang, x, t, 0, th);
      eval_spline(&M->eem[e], x, t, 0, pe);
      double d[3] = {pe[0] - r[0], pe[1] - r[1], pe[2] - r[2]};
      cplx cth[3] = {th[0], th[1], th[2]}, out[3];
      rom_c(cth, d, out);
      if (g)
        for (int i = 0; i < 3; ++i) g[r0 + i] = creal(out[i]);
      if (J) {
        cplx R[9];
        rot_c(cth, R);
        double Rt[9], mRt[9], A[9];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) {
            Rt[3 * i + j] = creal(R[3 * j + i]);
            mRt[3 * i + j] = -Rt[3 * i + j];
          }
        scatter(J, n, r0, 3, Rt, &M->eem[e], t, 0);
        scatter(J, n, r0, 3, mRt, &M->lin, t, 0);
        for (int j = 0; j < 3; ++j) {
          cplx a0[3] = {th[0], th[1], th[2]};
          a0[j] += hcs * I;
          rom_c(a0, d, out);
          for (int i = 0; i < 3; ++i) A[3 * i + j] = cimag(out[i]) / hcs;
        }
        scatter(J, n, r0, 3, A, &M->ang, t, 0);
      }
    }

  /* ---- force: unilateral + friction pyramid at optimised force nodes (force_constraint.cc) -- */
  for (int e = 0; e < QO_NEE; ++e)
    for (int j = 0; j < M->n_fnodes[e]; ++j) {
      int node = M->fnode_id[e][j], sv = stance_var(M, e, M->fnode_stance[e][j]);
      int r0 = L->off_force[e] + 5 * j;
      double f[3] = {nodeval(&M->eef[e], x, node, 0, 0), nodeval(&M->eef[e], x, node, 0, 1),
                     nodeval(&M->eef[e], x, node, 0, 2)};
      terr t = terrain_at(p, x[sv], x[sv + 1]);
      double b[3][3], bx[3][3], by[3][3];
      for (int w = 0; w < 3; ++w) terrain_basis(&t, w, b[w], bx[w], by[w]);
      /* rows: n, t1 - mu n, t1 + mu n, t2 - mu n, t2 + mu n */
      const double ct[5][3] = {{1, 0, 0}, {-p->mu, 1, 0}, {p->mu, 1, 0}, {-p->mu, 0, 1}, {p->mu, 0, 1}};
      for (int row = 0; row < 5; ++row) {
        double v[3], vx[3], vy[3];
        for (int i = 0; i < 3; ++i) {
          v[i] = ct[row][0] * b[0][i] + ct[row][1] * b[1][i] + ct[row][2] * b[2][i];
          vx[i] = ct[row][0] * bx[0][i] + ct[row][1] * bx[1][i] + ct[row][2] * bx[2][i];
          vy[i] = ct[row][0] * by[0][i] + ct[row][1] * by[1][i] + ct[row][2] * by[2][i];
        }
        if (g) g[r0 + row] = f[0] * v[0] + f[1] * v[1] + f[2] * v[2];
        if (J) {
          for (int i = 0; i < 3; ++i)
            J[(size_t)(r0 + row) * n + M->eef[e].idx[node][0][i]] += v[i];
          J[(size_t)(r0 + row) * n + sv] += f[0] * vx[0] + f[1] * vx[1] + f[2] * vx[2];
          J[(size_t)(r0 + row) * n + sv + 1] += f[0] * vy[0] + f[1] * vy[1] + f[2] * vy[2];
        }
      }
    }

  /* ---- swing (UPSTREAM swing_constraint.cc): mid node xy = centre, vel = dist / t_swing_avg - */
  for (int e = 0; e < QO_NEE; ++e) {
    const qo_spline *S = &M->eem[e];
    int row = L->off_swing[e];
    for (int node = 1; node < S->n_polys; ++node) {
      if (S->idx[node][1][0] < 0) continue; /* not a swing mid node */
      for (int d = 0; d < 2; ++d) {
        int ip = S->idx[node - 1][0][d], in = S->idx[node + 1][0][d];
        int ic = S->idx[node][0][d], iv = S->idx[node][1][d];
        double dist = x[in] - x[ip];
        if (g) {
          g[row] = x[ic] - (x[ip] + 0.5 * dist);
          g[row + 1] = x[iv] - dist / p->t_swing_avg;
        }
        if (J) {
          J[(size_t)row * n + ic] += 1;
          J[(size_t)row * n + ip] += -0.5;
          J[(size_t)row * n + in] += -0.5;
          J[(size_t)(row + 1) * n + iv] += 1;
          J[(size_t)(row + 1) * n + ip] += 1 / p->t_swing_avg;
          J[(size_t)(row + 1) * n + in] += -1 / p->t_swing_avg;
        }
        row += 2;
      }
    }
  }
}
static void eval_all(const qo_params *p, const qo_model *M, const double *x, double *g, double *J) {
  eval_all_z(p, M, x, g, J, NULL, NULL);
}


int qo_constraints(const qo_params *p, const double *x, double *g) {
  qo_model *M = (qo_model *)malloc(sizeof(qo_model));
  int rc = build_model(p, M);
  if (rc == 0) eval_all(p, M, x, g, NULL);
  free(M);
  return rc;
}

int qo_jacobian(const qo_params *p, const double *x, double *J) {
  qo_model *M = (qo_model *)malloc(sizeof(qo_model));
  int rc = build_model(p, M);
  if (rc == 0) eval_all(p, M, x, NULL, J);
  free(M);
  return rc;
}

/* CSV row contract: QTOS/utils.py:107-148 (vec_to_cmd_pose), scripts/run.py:129-137 */
int qo_sample(const qo_params *p, const double *x, double t0, double hz, int n_rows, double *rows) {
  qo_model *M = (qo_model *)malloc(sizeof(qo_model));
  int rc = build_model(p, M);
  if (rc) { free(M); return rc; }
  for (int k = 0; k < n_rows; ++k) {
    double t = k / hz;
    if (t > M->L.T) t = M->L.T;
    double *row = rows + (size_t)37 * k;
    row[0] = t0 + k / hz;
    eval_spline(&M->lin, x, t, 0, row + 1);
    eval_spline(&M->ang, x, t, 0, row + 4);
    for (int e = 0; e < QO_NEE; ++e) eval_spline(&M->eem[e], x, t, 0, row + 7 + 3 * e);
    eval_spline(&M->lin, x, t, 1, row + 19);
    eval_spline(&M->ang, x, t, 1, row + 22);
    for (int e = 0; e < QO_NEE; ++e) eval_spline(&M->eef[e], x, t, 0, row + 25 + 3 * e);
  }
  free(M);
  return 0;
}

static double max_violation(const qo_model *M, const double *g, const double *clo,
                            const double *chi) {
  double v = 0;
  for (int i = 0; i < M->L.n_cons; ++i) {
    double a = clo[i] - g[i], b = g[i] - chi[i];
    if (a > v) v = a;
    if (b > v) v = b;
  }
  return v;
}

double qo_max_violation(const qo_params *p, const double *x) {
  qo_model *M = (qo_model *)malloc(sizeof(qo_model));
  if (build_model(p, M)) { free(M); return -1; }
  int m = M->L.n_cons;
  double *g = (double *)malloc(sizeof(double) * 3 * m), *lo = g + m, *hi = lo + m;
  eval_all(p, M, x, g, NULL);
  qo_con_bounds(p, lo, hi);
  double v = max_violation(M, g, lo, hi);
  free(g);
  free(M);
  return v;
}


/* ============================================================================================ */
/* Solver: primal-dual interior point on the feasibility NLP with a proximal Hessian W = delta I */
/* (the reference runs Ipopt with an L-BFGS Hessian on a problem without cost terms,            */
/*  logs/towr_log.out:42,131; each step is therefore a damped least-norm Newton step on the     */
/*  constraints).  Condensed KKT  [W + Ji' S Ji, Je'; Je, -eps I]  factorised by a skyline LDL'.  */
/* ============================================================================================ */
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

typedef struct {
  int n;
  int *first;    /* first stored column of row i */
  size_t *start; /* offset of row i (entry for column first[i]) */
  double *a;     /* row-wise skyline, diagonal is the last entry of each row */
} skyline;

static void sky_alloc(skyline *S, int n, const int *first) {
  S->n = n;
  S->first = (int *)malloc(sizeof(int) * n);
  S->start = (size_t *)malloc(sizeof(size_t) * (n + 1));
  size_t tot = 0;
  for (int i = 0; i < n; ++i) {
    S->first[i] = first[i];
    S->start[i] = tot;
    tot += (size_t)(i - first[i] + 1);
  }
  S->start[n] = tot;
  S->a = (double *)calloc(tot, sizeof(double));
}
static void sky_free(skyline *S) { free(S->first); free(S->start); free(S->a); }
static inline double *sky_at(skyline *S, int i, int j) { /* j <= i, j >= first[i] */
  return S->a + S->start[i] + (j - S->first[i]);
}

/* in-place LDL^T (no pivoting; the matrix is quasi-definite).  Returns 0 or -1 on a zero pivot */
static int sky_factor(skyline *S) {
  int n = S->n;
  for (int i = 0; i < n; ++i) {
    double *ri = S->a + S->start[i];
    int fi = S->first[i];
    /* u[i][j] for j = fi..i-1 */
    for (int j = fi; j < i; ++j) {
      int fj = S->first[j];
      const double *rj = S->a + S->start[j];
      int k0 = fi > fj ? fi : fj;
      double acc = ri[j - fi];
      for (int k = k0; k < j; ++k) acc -= ri[k - fi] * rj[k - fj];
      ri[j - fi] = acc; /* holds u = L*D for now */
    }
    double d = ri[i - fi];
    for (int j = fi; j < i; ++j) {
      double dj = *(S->a + S->start[j] + (j - S->first[j]));
      double l = ri[j - fi] / dj;
      d -= ri[j - fi] * l;
      ri[j - fi] = l;
    }
    if (d == 0.0 || d != d) return -1;
    ri[i - fi] = d;
  }
  return 0;
}
static void sky_solve(const skyline *S, double *b) {
  int n = S->n;
  for (int i = 0; i < n; ++i) {
    const double *ri = S->a + S->start[i];
    int fi = S->first[i];
    double acc = b[i];
    for (int j = fi; j < i; ++j) acc -= ri[j - fi] * b[j];
    b[i] = acc;
  }
  for (int i = 0; i < n; ++i) b[i] /= S->a[S->start[i] + (i - S->first[i])];
  for (int i = n - 1; i >= 0; --i) {
    const double *ri = S->a + S->start[i];
    int fi = S->first[i];
    double bi = b[i];
    for (int j = fi; j < i; ++j) b[j] -= ri[j - fi] * bi;
  }
}

int qo_ldlt_solve_dense(int n, const double *A, double *b) {
  int *first = (int *)malloc(sizeof(int) * n);
  for (int i = 0; i < n; ++i) {
    int f = i;
    for (int j = 0; j < i; ++j)
      if (A[(size_t)i * n + j] != 0.0) { f = j; break; }
    first[i] = f;
  }
  skyline S;
  sky_alloc(&S, n, first);
  for (int i = 0; i < n; ++i)
    for (int j = first[i]; j <= i; ++j) *sky_at(&S, i, j) = A[(size_t)i * n + j];
  int rc = sky_factor(&S);
  if (rc == 0) sky_solve(&S, b);
  sky_free(&S);
  free(first);
  return rc;
}

void qo_default_options(qo_options *o) {
  o->max_iter = 24;
  o->tol = 1e-4;
  o->mu_init = 0.1;
  o->mu_min = 1e-9;
  o->delta_x = 1e-2;
  o->eps_dual = 1e-8;
  o->slack_push = 0.2;
  o->warm_slack_push = 0.01;
  o->chord_tol = 4e-3;
  o->stall_alpha = 1e-2;
  o->chord_max = 2;
  o->chord_shrink = 1.0 / 3.0;
  o->stall_iters = 5;
  o->hold_from = 2;
  o->hold_weight = 1e6;
  o->hold_tol = 0.25;
  o->warm_start = 0;
  o->verbose = 0;
  o->swing_start_on_rule = 0;
  o->eps_dual_swing = -1.0;
  o->eps_dual_acc = -1.0;
  o->mu_superlinear = 1;
}

/* time stamp of every variable / constraint row: used only to order the KKT unknowns */
static void unknown_times(const qo_params *p, const qo_model *M, double *tv, double *tc) {
  const qo_layout *L = &M->L;
  const qo_spline *sp[2 + 2 * QO_NEE];
  int ns = 0;
  sp[ns++] = &M->lin;
  sp[ns++] = &M->ang;
  for (int e = 0; e < QO_NEE; ++e) sp[ns++] = &M->eem[e];
  for (int e = 0; e < QO_NEE; ++e) sp[ns++] = &M->eef[e];
  for (int s = 0; s < ns; ++s) {
    double t = 0;
    for (int node = 0; node <= sp[s]->n_polys; ++node) {
      for (int q = 0; q < 2; ++q)
        for (int d = 0; d < 3; ++d) {
          int v = sp[s]->idx[node][q][d];
          if (v >= 0) tv[v] = t; /* later node sharing the variable wins (end of stance) */
        }
      if (node < sp[s]->n_polys) t += sp[s]->dur[node];
    }
  }
  for (int e = 0; e < QO_NEE; ++e) {
    const qo_spline *S = &M->eem[e];
    double t = 0;
    int sw = 0;
    for (int node = 1; node <= S->n_polys; ++node) {
      t += S->dur[node - 1];
      tc[L->off_terrain[e] + node - 1] = t;
      if (node < S->n_polys && S->idx[node][1][0] >= 0) {
        for (int r = 0; r < 4; ++r) tc[L->off_swing[e] + 4 * sw + r] = t;
        sw++;
      }
    }
    for (int k = 0; k < M->n_rom; ++k)
      for (int d = 0; d < 3; ++d) tc[L->off_rom[e] + 3 * k + d] = M->t_rom[k];
    for (int j = 0; j < M->n_fnodes[e]; ++j) {
      double tt = 0;
      for (int i = 0; i < M->fnode_id[e][j]; ++i) tt += M->eef[e].dur[i];
      for (int r = 0; r < 5; ++r) tc[L->off_force[e] + 5 * j + r] = tt;
    }
  }
  for (int k = 0; k < M->n_dyn; ++k)
    for (int d = 0; d < 6; ++d) tc[L->off_dyn + 6 * k + d] = M->t_dyn[k];
  for (int j = 0; j + 1 < M->lin.n_polys; ++j)
    for (int d = 0; d < 3; ++d) {
      tc[L->off_acc_lin + 3 * j + d] = (j + 1) * p->dt_base;
      tc[L->off_acc_ang + 3 * j + d] = (j + 1) * p->dt_base;
    }
}

typedef struct { double key; int id; } keyed;
static int cmp_keyed(const void *a, const void *b) {
  const keyed *x = (const keyed *)a, *y = (const keyed *)b;
  if (x->key < y->key) return -1;
  if (x->key > y->key) return 1;
  return x->id - y->id;
}

/* The swing mid nodes (x, y, v_x, v_y) placed on the swing rule (UPSTREAM swing_constraint.cc, solved for the mid node):
 * mirrors what the product's reduce_swing does to a starting point, see qo_options.swing_start_on_rule. */
static void project_swings(const qo_params *p, const qo_model *M, double *x) {
  for (int e = 0; e < QO_NEE; ++e) {
    const qo_spline *S = &M->eem[e];
    for (int node = 1; node < S->n_polys; ++node) {
      if (S->idx[node][1][0] < 0) continue;
      for (int d = 0; d < 2; ++d) {
        const int ip = S->idx[node - 1][0][d], in = S->idx[node + 1][0][d], ic = S->idx[node][0][d], iv = S->idx[node][1][d];
        x[ic] = fma(0.5, x[ip], 0.5 * x[in]);
        x[iv] = fma(-1.0 / p->t_swing_avg, x[ip], (1.0 / p->t_swing_avg) * x[in]);
      }
    }
  }
}
int qo_project_swings(const qo_params *p, double *x) {
  qo_model *M = (qo_model *)malloc(sizeof(qo_model));
  if (!M || build_model(p, M)) { free(M); return -1; }
  project_swings(p, M, x);
  free(M);
  return 0;
}

/* the dense Jacobian of a solve, kept per thread between the solves of one batch (qo_solve_batch releases it) */
static _Thread_local double *tls_J = NULL;
static _Thread_local size_t tls_J_cap = 0;
static void tls_J_release(void) { free(tls_J); tls_J = NULL; tls_J_cap = 0; }

int qo_solve(const qo_params *p, const qo_problem *q, const qo_options *o, double *x, qo_info *info) {
  qo_model *M = (qo_model *)malloc(sizeof(qo_model));
  if (build_model(p, M)) { free(M); return -1; }
  const qo_layout *L = &M->L;
  const int n = L->n_vars, m = L->n_cons;
  double t_eval = 0, t_fac = 0, t0;
  memset(info, 0, sizeof(*info));

  double *xl = (double *)malloc(sizeof(double) * 2 * n), *xh = xl + n;
  double *cl = (double *)malloc(sizeof(double) * 2 * m), *ch = cl + m;
  qo_var_bounds(p, q, xl, xh);
  qo_con_bounds(p, cl, ch);
  if (!o->warm_start) qo_initial_guess(p, q, x);
  for (int i = 0; i < n; ++i)
    if (xl[i] == xh[i]) x[i] = xl[i];
  if (o->swing_start_on_rule) project_swings(p, M, x);

  double *g = (double *)malloc(sizeof(double) * m), *gt = (double *)malloc(sizeof(double) * m);
  /* the dense Jacobian (30 MB on the 100-knot problem) is kept per thread between solves: allocated and freed per solve
   * it is an mmap / munmap pair with page faults and TLB shoot-downs across every thread of the process (qo_solve_batch) */
  if (tls_J_cap < (size_t)m * n) {
    free(tls_J);
    tls_J = (double *)malloc(sizeof(double) * (size_t)m * n);
    tls_J_cap = tls_J ? (size_t)m * n : 0;
  }
  double *J = tls_J;
  if (!J) { free(xl); free(cl); free(g); free(gt); return -1; }   /* (without it eval_all would silently skip the Jacobian) */
  double *xt = (double *)malloc(sizeof(double) * n);

  /* ---- working sets: free variables, de-duplicated equality rows, inequality rows ---------- */
  int *vpos = (int *)malloc(sizeof(int) * n); /* var -> KKT position or -1 */
  int nf = 0;
  for (int i = 0; i < n; ++i) nf += (xl[i] != xh[i]);
  /* row structure from one Jacobian evaluation at a perturbed point */
  {
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      xt[i] = x[i] + 1e-2 * ((double)(s >> 8) / 16777216.0 - 0.5);
    }
    eval_all(p, M, xt, NULL, J);
  }
  int *rowtype = (int *)calloc(m, sizeof(int)); /* 0 dropped, 1 equality, 2 inequality */
  int nE = 0, nI = 0;
  /* CSR of the structural pattern restricted to free variables */
  int *rp = (int *)malloc(sizeof(int) * (m + 1));
  int nnz = 0;
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < n; ++c)
      if (xl[c] != xh[c] && J[(size_t)r * n + c] != 0.0) nnz++;
  int *ci = (int *)malloc(sizeof(int) * (nnz + 1));
  nnz = 0;
  for (int r = 0; r < m; ++r) {
    rp[r] = nnz;
    for (int c = 0; c < n; ++c)
      if (xl[c] != xh[c] && J[(size_t)r * n + c] != 0.0) ci[nnz++] = c;
  }
  rp[m] = nnz;
  for (int r = 0; r < m; ++r) {
    int len = rp[r + 1] - rp[r];
    if (cl[r] != ch[r]) { rowtype[r] = 2; nI++; continue; }
    if (len == 0) continue; /* constant row: nothing can change it */
    /* the time grid repeats T (floor(T/dt)*dt == T): the second copy of the dynamics block is
     * the same six equations and is left out of the working set */
    if (M->n_dyn >= 2 && fabs(M->t_dyn[M->n_dyn - 1] - M->t_dyn[M->n_dyn - 2]) < 1e-9 &&
        r >= L->off_dyn + 6 * (M->n_dyn - 1) && r < L->off_dyn + 6 * M->n_dyn)
      continue;
    int dup = 0;
    for (int r2 = 0; r2 < r && !dup; ++r2) {
      if (rowtype[r2] != 1 || rp[r2 + 1] - rp[r2] != len) continue;
      int same = 1;
      for (int k = 0; k < len && same; ++k) {
        int c = ci[rp[r] + k];
        if (ci[rp[r2] + k] != c ||
            fabs(J[(size_t)r * n + c] - J[(size_t)r2 * n + c]) > 1e-9 * (1 + fabs(J[(size_t)r * n + c])))
          same = 0;
      }
      dup = same;
    }
    if (!dup) { rowtype[r] = 1; nE++; }
  }
  int *Er = (int *)malloc(sizeof(int) * (nE + 1)), *Ir = (int *)malloc(sizeof(int) * (nI + 1));
  nE = nI = 0;
  for (int r = 0; r < m; ++r) {
    if (rowtype[r] == 1) Er[nE++] = r;
    if (rowtype[r] == 2) Ir[nI++] = r;
  }
  /* ---- ordering of the nf + nE unknowns by time ------------------------------------------- */
  const int N = nf + nE;
  double *tv = (double *)calloc(n, sizeof(double)), *tc = (double *)calloc(m, sizeof(double));
  unknown_times(p, M, tv, tc);
  keyed *ks = (keyed *)malloc(sizeof(keyed) * N);
  {
    int k = 0;
    for (int i = 0; i < n; ++i)
      if (xl[i] != xh[i]) { ks[k].key = tv[i]; ks[k].id = i; k++; }
    for (int e = 0; e < nE; ++e) { ks[k].key = tc[Er[e]] + 1e-7; ks[k].id = n + e; k++; }
  }
  qsort(ks, N, sizeof(keyed), cmp_keyed);
  int *epos = (int *)malloc(sizeof(int) * (nE + 1));
  for (int i = 0; i < n; ++i) vpos[i] = -1;
  for (int k = 0; k < N; ++k) {
    if (ks[k].id < n) vpos[ks[k].id] = k;
    else epos[ks[k].id - n] = k;
  }
  int *first = (int *)malloc(sizeof(int) * N);
  for (int i = 0; i < N; ++i) first[i] = i;
  for (int r = 0; r < m; ++r) {
    if (rowtype[r] != 2) continue;
    int mn = N;
    for (int k = rp[r]; k < rp[r + 1]; ++k) if (vpos[ci[k]] < mn) mn = vpos[ci[k]];
    for (int k = rp[r]; k < rp[r + 1]; ++k) if (mn < first[vpos[ci[k]]]) first[vpos[ci[k]]] = mn;
  }
  for (int e = 0; e < nE; ++e) {
    int r = Er[e], pe = epos[e];
    for (int k = rp[r]; k < rp[r + 1]; ++k) {
      int pv = vpos[ci[k]];
      int a = pv < pe ? pv : pe, b = pv < pe ? pe : pv;
      if (a < first[b]) first[b] = a;
    }
  }
  skyline K;
  sky_alloc(&K, N, first);
  if (o->verbose) fprintf(stderr, "oracle: N=%d (free %d, eq %d), ineq %d, skyline %zu\n", N, nf, nE, nI, K.start[N]);

  /* ---- interior-point state ---------------------------------------------------------------- */
  double *s = (double *)malloc(sizeof(double) * 8 * (nI + 1));
  double *zl = s + nI, *zu = zl + nI, *ds = zu + nI, *dzl = ds + nI, *dzu = dzl + nI, *Sig = dzu + nI, *wv = Sig + nI;
  double *rhs = (double *)malloc(sizeof(double) * N);
  double *dx = (double *)calloc(n, sizeof(double));
  t0 = now_s();
  eval_all(p, M, x, g, NULL);
  t_eval += now_s() - t0;
  double theta0 = 0;
  for (int i = 0; i < nI; ++i) {
    int r = Ir[i];
    double l = cl[r], u = ch[r];
    int hl = l > -1e19, hu = u < 1e19;
    /* slack start: pushed inside the bounds by a fraction of the range (Ipopt bound_push /
     * bound_frac).  A cold start uses a LARGE push (o->slack_push, 0.2): the first Newton steps are
     * then not cut by the fraction-to-the-boundary rule (4 iterations instead of 4-6 on the
     * benchmark goals); a warm start keeps Ipopt's 0.01 so that a feasible point stays put. */
    const double kp = o->warm_start ? (o->warm_slack_push > 0 ? o->warm_slack_push : 0.01) : o->slack_push;
    double pl = hl ? kp * fmax(1.0, fabs(l)) : 0, pu = hu ? kp * fmax(1.0, fabs(u)) : 0;
    if (hl && hu) { pl = fmin(pl, kp * (u - l)); pu = fmin(pu, kp * (u - l)); }
    double si = g[r];
    if (hl) si = fmax(si, l + pl);
    if (hu) si = fmin(si, u - pu);
    s[i] = si;
    theta0 = fmax(theta0, fabs(g[r] - si));
  }
  for (int e = 0; e < nE; ++e) theta0 = fmax(theta0, fabs(g[Er[e]]));
  double mu = fmax(o->mu_min, fmin(o->mu_init, 0.01 * theta0 * theta0));
  for (int i = 0; i < nI; ++i) {
    int r = Ir[i];
    zl[i] = cl[r] > -1e19 ? mu / (s[i] - cl[r]) : 0.0;
    zu[i] = ch[r] < 1e19 ? mu / (ch[r] - s[i]) : 0.0;
  }
  info->inf_pr0 = max_violation(M, g, cl, ch);
  int status = 1, it, held = 0;
  double viol = 0;
  /* stall detection: remember the iterate with the lowest violation; give up when it has not been
   * improved for stall_iters iterations (cycling on a discontinuous terrain edge) and return it */
  double best_viol = INFINITY;
  int best_it = 0;
  double *xbest = (double *)malloc(sizeof(double) * n);
  int chord_ok = 0;      /* the last step was a full step (alpha = 1) of a freshly factored system */
  int chord_again = 0;   /* the last step was a full chord step and the factorisation has chord steps left */
  int chord_run = 0;     /* chord steps taken with the current factorisation */
  double viol_prev = INFINITY;   /* violation in front of the last step */
  int chord_banned = 0;  /* a chord step of this solve was discarded: every later iteration factors */
  int n_chord = 0;
  int jam = 0;           /* steps in a row shorter than stall_alpha */
  for (it = 0; it < o->max_iter; ++it) {
    double theta = 0;
    viol = 0;
    for (int e = 0; e < nE; ++e) theta = fmax(theta, fabs(g[Er[e]]));
    viol = theta;
    for (int i = 0; i < nI; ++i) {
      int r = Ir[i];
      theta = fmax(theta, fabs(g[r] - s[i]));
      viol = fmax(viol, fmax(cl[r] - g[r], g[r] - ch[r]));
    }
    if (o->verbose) fprintf(stderr, "oracle: it %2d viol %.3e theta %.3e mu %.1e\n", it, viol, theta, mu);
    if (viol <= o->tol && theta <= o->tol) { status = 0; break; }
    if (viol < best_viol) { best_viol = viol; best_it = it; memcpy(xbest, x, sizeof(double) * n); }
    else if (o->stall_iters > 0 && it - best_it >= o->stall_iters) {
      memcpy(x, xbest, sizeof(double) * n);
      eval_all(p, M, x, g, NULL);
      break;
    }
    if (jam >= 2) {   /* jammed against its bounds: stop like a stalled problem */
      memcpy(x, xbest, sizeof(double) * n);
      eval_all(p, M, x, g, NULL);
      break;
    }
    t0 = now_s();
    eval_all_z(p, M, x, NULL, J, rp, ci);   /* (everything the solver reads of J lies in the pattern rp / ci) */
    t_eval += now_s() - t0;
    t0 = now_s();
    const int chord = o->chord_tol > 0 && !chord_banned && viol <= o->chord_tol &&
                      (chord_ok || (chord_again && viol <= o->chord_shrink * viol_prev));
    chord_ok = 0;
    n_chord += chord;
    if (!chord) memset(K.a, 0, sizeof(double) * K.start[N]);
    memset(rhs, 0, sizeof(double) * N);
    if (!chord)
    for (int i = 0; i < n; ++i)
      if (vpos[i] >= 0) *sky_at(&K, vpos[i], vpos[i]) = o->delta_x;
    /* two-phase solve: once the first iterations have placed the feet, the stance footholds stay */
    if (o->hold_from > 0 && it >= o->hold_from && viol <= o->hold_tol) held = 1;
    if (held && !chord)
      for (int e = 0; e < QO_NEE; ++e)
        for (int sn = 0; sn < M->n_stance[e]; ++sn)
          for (int d = 0; d < 2; ++d) {
            const int v = stance_var(M, e, sn) + d;
            if (vpos[v] >= 0) *sky_at(&K, vpos[v], vpos[v]) = o->hold_weight;
          }
    for (int i = 0; i < nI; ++i) {
      int r = Ir[i];
      double l = cl[r], u = ch[r];
      int hl = l > -1e19, hu = u < 1e19;
      double dl = hl ? s[i] - l : 1, du = hu ? u - s[i] : 1;
      double sg = (hl ? zl[i] / dl : 0) + (hu ? zu[i] / du : 0);
      double gmu = -(hl ? mu / dl : 0) + (hu ? mu / du : 0);
      double rI = g[r] - s[i];
      Sig[i] = sg;
      double w = sg * rI + gmu;
      const double *Jr = J + (size_t)r * n;
      for (int a = rp[r]; a < rp[r + 1]; ++a) {
        int ca = ci[a], pa = vpos[ca];
        double ja = Jr[ca];
        rhs[pa] -= ja * w;
        if (!chord) for (int b = rp[r]; b < rp[r + 1]; ++b) {
          int pb = vpos[ci[b]];
          if (pb <= pa) *sky_at(&K, pa, pb) += sg * ja * Jr[ci[b]];
        }
      }
    }
    for (int e = 0; e < nE; ++e) {
      int r = Er[e], pe = epos[e];
      const double *Jr = J + (size_t)r * n;
      rhs[pe] = -g[r];
      if (chord) continue;
      {
        /* rows the product may have eliminated from its system carry their own eps (qo_options.eps_dual_swing / _acc) */
        double eps = o->eps_dual;
        if (o->eps_dual_swing >= 0 && r >= L->off_swing[0]) eps = o->eps_dual_swing;
        if (o->eps_dual_acc >= 0 && r >= L->off_acc_lin && r < L->off_rom[0]) eps = o->eps_dual_acc;
        *sky_at(&K, pe, pe) = -eps;
      }
      for (int a = rp[r]; a < rp[r + 1]; ++a) {
        int pv = vpos[ci[a]];
        if (pv < pe) *sky_at(&K, pe, pv) += Jr[ci[a]];
        else *sky_at(&K, pv, pe) += Jr[ci[a]];
      }
    }
    if (!chord && sky_factor(&K)) { status = 2; break; }
    sky_solve(&K, rhs);
    t_fac += now_s() - t0;
    for (int i = 0; i < n; ++i) dx[i] = vpos[i] >= 0 ? rhs[vpos[i]] : 0.0;
    /* slack / dual steps and fraction to the boundary */
    double tau = fmax(0.99, 1 - mu), amax = 1.0, az = 1.0, th0 = 0;
    for (int e = 0; e < nE; ++e) th0 += fabs(g[Er[e]]);
    for (int i = 0; i < nI; ++i) {
      int r = Ir[i];
      double l = cl[r], u = ch[r];
      int hl = l > -1e19, hu = u < 1e19;
      double dl = hl ? s[i] - l : 1, du = hu ? u - s[i] : 1;
      double rI = g[r] - s[i], jd = 0;
      const double *Jr = J + (size_t)r * n;
      for (int a = rp[r]; a < rp[r + 1]; ++a) jd += Jr[ci[a]] * dx[ci[a]];
      ds[i] = jd + rI;
      dzl[i] = hl ? mu / dl - zl[i] - zl[i] / dl * ds[i] : 0;
      dzu[i] = hu ? mu / du - zu[i] + zu[i] / du * ds[i] : 0;
      if (hl && ds[i] < 0) amax = fmin(amax, tau * dl / -ds[i]);
      if (hu && ds[i] > 0) amax = fmin(amax, tau * du / ds[i]);
      if (hl && dzl[i] < 0) az = fmin(az, tau * zl[i] / -dzl[i]);
      if (hu && dzu[i] < 0) az = fmin(az, tau * zu[i] / -dzu[i]);
      th0 += fabs(rI);
    }
    /* backtracking on the l1 infeasibility */
    double al = amax, th = 0;
    int ls;
    for (ls = 0; ls < 6; ++ls) {
      for (int i = 0; i < n; ++i) xt[i] = x[i] + al * dx[i];
      t0 = now_s();
      eval_all(p, M, xt, gt, NULL);
      t_eval += now_s() - t0;
      th = 0;
      for (int e = 0; e < nE; ++e) th += fabs(gt[Er[e]]);
      for (int i = 0; i < nI; ++i) th += fabs(gt[Ir[i]] - (s[i] + al * ds[i]));
      if (th <= (1 - 1e-4 * al) * th0 || th < 1e-9) break;
      if (ls < 5) al *= 0.5;
    }
    if (o->verbose) fprintf(stderr, "oracle:    amax %.3f alpha %.4f az %.3f ls %d th %.3e -> %.3e%s\n", amax, al, az, ls, th0, th, chord ? (al == 1.0 ? " (chord)" : " (chord, rejected)") : "");
    /* a chord step is taken whole or not at all: cut by the fraction-to-the-boundary rule or by the line search
     * it is discarded (the iterate stays, the next iteration factors) -- a damped chord step can park a slack
     * right on its bound, and the KKT matrix of that point is too badly scaled for the block elimination */
    jam = (o->stall_alpha > 0 && !(chord && al != 1.0) && al < o->stall_alpha) ? jam + 1 : 0;
    if (chord && al != 1.0) {
      chord_banned = 1;   /* one discarded chord step and the solve factors every iteration from then on */
      al = 0.0;
      az = 0.0;
      memcpy(xt, x, sizeof(double) * n);
      memcpy(gt, g, sizeof(double) * m);
    }
    memcpy(x, xt, sizeof(double) * n);
    memcpy(g, gt, sizeof(double) * m);
    for (int i = 0; i < nI; ++i) {
      int r = Ir[i];
      double l = cl[r], u = ch[r];
      int hl = l > -1e19, hu = u < 1e19;
      s[i] += al * ds[i];
      zl[i] += az * dzl[i];
      zu[i] += az * dzu[i];
      const double kap = 1e10;
      if (hl) zl[i] = fmin(fmax(zl[i], mu / (kap * (s[i] - l))), kap * mu / (s[i] - l));
      if (hu) zu[i] = fmin(fmax(zu[i], mu / (kap * (u - s[i]))), kap * mu / (u - s[i]));
    }
    if (al > 0.3) mu = o->mu_superlinear ? fmax(fmax(o->mu_min, o->tol), fmin(0.2 * mu, mu * sqrt(mu))) : fmax(o->mu_min, 0.2 * mu);   /* Ipopt's monotone update (qo_options) */
    chord_ok = !chord && al == 1.0;
    chord_run = chord ? chord_run + 1 : 0;
    chord_again = chord && al == 1.0 && chord_run < o->chord_max;
    viol_prev = viol;
  }
  info->status = status;
  info->iters = it;
  info->inf_pr = max_violation(M, g, cl, ch);
  info->mu = mu;
  info->eval_secs = t_eval;
  info->factor_secs = t_fac;
  (void)wv;
  sky_free(&K);
  free(first); free(epos); free(ks); free(tv); free(tc); free(Er); free(Ir); free(ci); free(rp);
  free(rowtype); free(vpos); free(xt); free(gt); free(g); free(cl); free(xl); free(s);
  free(rhs); free(dx); free(M); free(xbest);
  return status;
}

/* Batch of independent solves, OpenMP over the problems (bench.py cpu_baseline: the multi-core figure).
 * x_io: n_problems x n_vars; every solve is exactly qo_solve. */
int qo_solve_batch(const qo_params *p, int n_problems, const qo_problem *q, const qo_options *o,
                   double *x_io, qo_info *info, int n_threads) {
  qo_layout L;
  if (qo_get_layout(p, &L)) return -1;
  int bad = 0;
  /* a solve allocates and frees a few MB in blocks above glibc's mmap threshold: each would be an mmap / munmap pair -- page
   * faults on fresh pages and a TLB shoot-down across every thread of the process per munmap.  Keep them in the threads' arenas. */
  mallopt(M_MMAP_THRESHOLD, 32 * 1024 * 1024);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : bad)
#endif
  for (int i = 0; i < n_problems; ++i)
    if (qo_solve(p, q + i, o, x_io + (size_t)i * L.n_vars, info + i) < 0) bad++;
  return bad ? -1 : 0;
}

/* Every thread of the OpenMP team (and the caller) gives its dense Jacobian back -- 30 MB each on the 100-knot problem, kept
 * between batches so that a timed pass does not pay for first-touch page faults -- and the allocator gets its defaults again.
 * bench.py calls it behind the cpu_baseline leg; a host process that runs one batch and goes on should too. */
void qo_release_buffers(int n_threads) {
#ifdef _OPENMP
  /* the team that allocated the buffers frees them; the caller's OpenMP thread count is left as it was */
  const int before = omp_get_max_threads();
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
#endif
  tls_J_release();
  tls_J_release();
#ifdef _OPENMP
  omp_set_num_threads(before);
#endif
  /* (glibc: a freed block above M_MMAP_THRESHOLD went back to the system at once; below it, malloc_trim returns what the
   *  arenas hold.  No mallopt: setting a threshold would switch glibc's dynamic adjustment off for the rest of the process.) */
  malloc_trim(0);
}

