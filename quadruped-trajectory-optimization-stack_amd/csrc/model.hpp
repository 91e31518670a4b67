// model.hpp -- host-side description of the planning NLP and the POD descriptors the device
// kernels consume.
//
// The NLP is the one the reference hands to Ipopt (sets, sizes and order: logs/towr_log.out:99-129;
// formulation: towr v1.4, of which the reference's solver is a fork -- Dockerfile:45).  Everything
// time-dependent is resolved here, on the host, once per planner: every constraint instance is
// reduced to "local inputs are fixed sparse linear combinations of node variables"
// (cubic-Hermite weights at a fixed time), so the device code never walks a spline.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/qtos_planner.h"

namespace qtos {

constexpr int NEE = QTOS_NEE;
constexpr double BIG = 1e20;

// ---- descriptors shared with the device ------------------------------------------------------
// 3-vector input: value[d] = sum_a w[a] * x[var[3a+d]]  (var < 0 contributes 0);
// a = (node k pos, node k vel, node k+1 pos, node k+1 vel) of the polynomial containing the time.
struct VecIn {
  double w[4];
  int var[12];
};

struct DynInst {  // 6 rows: angular (3) then linear (3) momentum balance at one time
  VecIn r, a, th, thd, thdd, p[NEE], f[NEE];
  int row0, goff, ncol, in_kkt;
  short c_lin[12], c_ang[12], c_p[NEE][12], c_f[NEE][12];  // block column of input slot, or -1
};
struct RomInst {  // 3 rows: R(th)^T (p - r) for one foot at one time
  VecIn r, th, p;
  int row0, goff, ncol, ee;
  short c_lin[12], c_ang[12], c_p[12];
};
struct TerrInst {  // 1 row: z - h(x,y) at a foot node
  int vx, vy, vz, row, goff, ncol, in_kkt;
  short cx, cy, cz, pad;
};
struct ForceInst {  // 5 rows: normal force + friction pyramid at a force node
  int vf[3], vsx, vsy, row0, goff, ncol;
  short cf[3], csx, csy, pad;
};
struct LinRow {  // 1 row with constant coefficients (acceleration continuity, swing)
  int row, n;
  int var[8];
  double coef[8];
};
// Local Jacobians of one instance, kept in LDS between the instance pass and the entry pass:
constexpr int DYN_LOC = 54;  // A_th, A_thd, A_thdd (9 each), sum f (3), f_e (12), r - p_e (12)
constexpr int ROM_LOC = 18;  // R (9), d/dtheta_j [R^T (p - r)] as columns (9)
// One column of a dynamics / range-of-motion Jacobian block (host only): which local Jacobian feeds
// it and the combined Hermite weights of the node slots that map onto this variable.
struct ColDesc {
  int inst, gbase, ncol;       // instance index, offset of G[0][col], row stride
  short kind, dim;             // dyn: 0 lin, 1 ang, 2+e foot e position, 6+e foot e force; rom: 0 lin, 1 ang, 2 foot
  double w0, w1, w2;
};
// What the device sees of those columns: every Jacobian entry that depends on the iterate is a linear
// form over the per-instance local Jacobians in LDS (`loc`), G[pos] = sum_t a_t * loc[off_t], listed in
// stream order so that neighbouring threads write neighbouring stream positions.  Entries that do
// not depend on the iterate (structural zeros of the 6-row columns, m*w, -w) are written into the
// stream once, with the other constants, when the planner is created.
struct LinTerm1 { int pos, off; double a; };
struct LinTerm3 { int pos, off[3]; double a[3]; };
// the same lists as the evaluation kernels read them (every workgroup streams them from L2 in every linearisation: bytes
// matter): 16-bit offsets into the local Jacobians, coefficients as indices into a table of the distinct values (a few
// thousand: the Hermite weights of the knot times), which the kernels keep in LDS
struct PackedTerm1 { unsigned pos; unsigned short off, ai; };
struct PackedTerm3 { unsigned pos; unsigned short off[3], ai[3]; };
// K2 assembly block: m consecutive constraint rows sharing one dense column list
struct Block {
  int kind;  // 0 equality, 1 inequality
  int m, n, row0, goff, gstatic, col_off, pad;
};
// one row of an inequality block, for the row-parallel product ds = Ji dx (k_step)
struct IqRow {
  int goff;     // stream offset of the row's first Jacobian entry
  int n;        // entries
  int col_off;  // into block_cols
  int row;      // constraint row
};
// one row of an inequality block for the helper waves of the backward sweep (sweep_backward: ds = Ji dx there)
struct SwTask {
  int goff;      // stream offset of the row's first Jacobian entry
  int n;         // entries
  int row;       // constraint row; -1: an empty place of the round
  int c16_off;   // into sw_c16: the positions of the block's columns in elimination order as 16-bit values, in the order the four lanes of a row read them
  int cpos_off;  // into sw_cpos: the same positions as ints (rows of more than 35 entries)
  int pad[3];
};
// initial guess per variable (towr SetByLinearInterpolation): x = a + frac*(b-a) or (b-a)/T
struct InitDesc {
  int set;  // 0 base-lin, 1 base-ang, 2+e ee-motion, 6+e ee-force
  int dim;
  int is_vel;
  int fix_src;  // -1 free; 0..23 start[k]; 24,25 goal x,y; 26 zero
  double frac;
};

struct Spline {
  int n_polys = 0;
  std::vector<double> dur;
  std::vector<std::array<int, 6>> idx;  // [node][q*3+d] -> variable or -1 (constant zero)

  void locate(double t, int &k, double &tau) const {
    // at junctions the previous polynomial is used (towr spline.cc GetSegmentID, eps 1e-10)
    const double eps = 1e-10;
    double acc = 0;
    int id = n_polys - 1;
    for (int i = 0; i < n_polys; ++i) {
      acc += dur[i];
      if (acc >= t - eps) { id = i; break; }
    }
    double tl = t;
    for (int i = 0; i < id; ++i) tl -= dur[i];
    k = id;
    tau = tl;
  }
  double node_time(int node) const {
    double t = 0;
    for (int i = 0; i < node; ++i) t += dur[i];
    return t;
  }
};

inline void hermite_weights(double T, double t, int deriv, double w[4]) {
  const double T2 = T * T, T3 = T2 * T, t2 = t * t, t3 = t2 * t;
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

// Cubic B-spline basis on a knot vector U: the four functions that live on span i (U[i] <= t <= U[i+1]) and their first
// two derivatives (The NURBS Book, algorithm A2.3, degree 3).  ders[d][a] = d-th derivative of N_{i-3+a} at t.
inline void bspline_ders(const std::vector<double> &U, int i, double t, double ders[3][4]) {
  const int p = 3;
  double ndu[4][4], left[4], right[4], a[2][4];
  ndu[0][0] = 1.0;
  for (int j = 1; j <= p; ++j) {
    left[j] = t - U[i + 1 - j];
    right[j] = U[i + j] - t;
    double saved = 0.0;
    for (int r = 0; r < j; ++r) {
      ndu[j][r] = right[r + 1] + left[j - r];
      const double tmp = ndu[r][j - 1] / ndu[j][r];
      ndu[r][j] = saved + right[r + 1] * tmp;
      saved = left[j - r] * tmp;
    }
    ndu[j][j] = saved;
  }
  for (int j = 0; j <= p; ++j) ders[0][j] = ndu[j][p];
  for (int r = 0; r <= p; ++r) {
    int s1 = 0, s2 = 1;
    a[0][0] = 1.0;
    for (int k = 1; k <= 2; ++k) {
      double d = 0.0;
      const int rk = r - k, pk = p - k;
      if (r >= k) { a[s2][0] = a[s1][0] / ndu[pk + 1][rk]; d = a[s2][0] * ndu[rk][pk]; }
      const int j1 = rk >= -1 ? 1 : -rk, j2 = (r - 1 <= pk) ? k - 1 : p - r;
      for (int j = j1; j <= j2; ++j) {
        a[s2][j] = (a[s1][j] - a[s1][j - 1]) / ndu[pk + 1][rk + j];
        d += a[s2][j] * ndu[rk + j][pk];
      }
      if (r <= pk) { a[s2][k] = -a[s1][k - 1] / ndu[pk + 1][r]; d += a[s2][k] * ndu[r][pk]; }
      ders[k][r] = d;
      std::swap(s1, s2);
    }
  }
  double f = p;
  for (int k = 1; k <= 2; ++k) {
    for (int j = 0; j <= p; ++j) ders[k][j] *= f;
    f *= (p - k);
  }
}

// a 3-vector input in SOLVER space with a weight per slot AND dimension (host only: the column structure and the weights of
// a Jacobian block; the values are evaluated in node space through VecIn)
struct VecInW {
  double w[12];
  int var[12];
};
inline VecInW widen(const VecIn &v) {
  VecInW o;
  for (int i = 0; i < 12; ++i) { o.var[i] = v.var[i]; o.w[i] = v.w[i / 3]; }
  return o;
}

inline VecIn make_in_poly(const Spline &S, int k, double tau, int deriv) {
  VecIn v;
  hermite_weights(S.dur[k], tau, deriv, v.w);
  for (int a = 0; a < 4; ++a)
    for (int d = 0; d < 3; ++d) v.var[3 * a + d] = S.idx[k + (a >> 1)][(a & 1) * 3 + d];
  return v;
}
inline VecIn make_in(const Spline &S, double t, int deriv) {
  int k;
  double tau;
  S.locate(t, k, tau);
  return make_in_poly(S, k, tau, deriv);
}

struct HostModel {
  QtosParams P;
  double T = 0;
  int n_vars = 0, n_cons = 0, n_base_nodes = 0;
  Spline lin, ang, eem[NEE], eef[NEE];
  int off_lin = 0, off_ang = 0, off_eem[NEE], off_eef[NEE];
  int off_terrain[NEE], off_dyn = 0, off_acc_lin = 0, off_acc_ang = 0, off_rom[NEE],
      off_force[NEE], off_swing[NEE];
  std::vector<double> t_dyn, t_rom;
  std::vector<double> var_time, con_time, con_lo, con_hi;   // var_time / con_time: the time KEYS of the elimination order (Symbolic), not physics
  std::vector<double> node_time;                            // the time of the node a model variable belongs to (k_shift_warm samples the previous plan there)
  std::vector<int> row_kind;  // 0 dropped from the working set, 1 equality, 2 inequality
  std::vector<InitDesc> init;
  // instances
  std::vector<DynInst> dyn;
  std::vector<RomInst> rom;
  std::vector<TerrInst> terr;
  std::vector<ForceInst> force;
  std::vector<LinRow> linrow;
  // assembly blocks (+ shared column list, static G for constant-coefficient rows)
  std::vector<Block> blocks;
  std::vector<int> block_cols;
  std::vector<double> g_static;
  long long g_doubles = 0;
  std::string err;

  bool is_free(int v) const { return init[v].fix_src < 0; }

  // ---- reduced base (QtosParams.reduce_base): the acceleration-continuity rows are linear with constant coefficients;
  // inside the KKT solve the base node values are replaced by the coefficients of a clamped cubic B-spline on the same
  // knots (a basis of exactly the C2 splines the rows describe): dx_nodes = Z dc.  No multipliers for those rows and half
  // the base unknowns; the Newton step is the same (the proximal term delta |dx_nodes|^2 becomes delta dc' Z'Z dc: static
  // entries of K).  Solver variables = the model variables (ids < n_vars; base node values are then not unknowns) followed
  // by the coefficients (ids n_vars ..).  The iterate, the constraint evaluation and the results stay in node space.
  int reduce_base = 0, n_sol = 0, n_coef = 0;
  std::vector<double> knots;                 // clamped knot vector of the base splines
  std::vector<int> span_of;                  // knot span of every base polynomial
  std::vector<int> cvar[2][3];               // [lin / ang][dim][coefficient] -> solver variable or -1 (no freedom: fixed end state)
  std::vector<char> replaced;                // per model variable: a base node value that the coefficients replace
  std::vector<char> coef_unknown;            // per coefficient id (solver variable - n_vars): is an unknown of the KKT system
  std::vector<double> sol_diag;              // per solver variable: proximal diagonal (delta_x, or delta_x (Z'Z)_jj for a coefficient)
  struct SymEntry { int a, b; double val; }; // static entries of K between two solver variables (delta_x Z'Z off the diagonal)
  std::vector<SymEntry> sym_static;
  std::vector<int> rec_var, rec_col;         // recovery dx_nodes = Z dc: model variable, then 4 (solver column or -1, weight) each
  std::vector<double> rec_w;
  // projection of given node values (a warm start) onto the coefficients' space: coefficient c = sum_a pc_w[4c + a] *
  // x[pc_var[4c + a]] (the blossom of one polynomial piece inside its support: exact for a spline of the space), then every
  // free base node value = sum_a pz_w[4i + a] * c[pz_col[4i + a]]
  std::vector<int> pc_var, pz_var, pz_col;
  std::vector<double> pc_w, pz_w;
  bool is_unknown(int v) const {
    if (v < n_vars) return init[v].fix_src < 0 && !replaced[v];
    return coef_unknown[v - n_vars] != 0;
  }
  // base input in solver space: the four B-spline coefficients that live on the polynomial containing t
  VecIn make_in_sol(int which, const Spline &S, double t, int deriv) const {
    if (!reduce_base) return make_in(S, t, deriv);
    int k;
    double tau;
    S.locate(t, k, tau);
    double ders[3][4];
    const int sp = span_of[k];
    bspline_ders(knots, sp, t, ders);
    VecIn v;
    for (int a = 0; a < 4; ++a) {
      v.w[a] = ders[deriv][a];
      for (int d = 0; d < 3; ++d) v.var[3 * a + d] = cvar[which][d][sp - 3 + a];
    }
    return v;
  }
  struct BaseIn { VecIn r, a, th, thd, thdd; };   // solver-space inputs of an instance (column structure and weights of its Jacobian)
  std::vector<BaseIn> dyn_sol, rom_sol;
  // ---- reduced swings (QtosParams.reduce_swing): the swing rule -- mid node x, y = centre of the neighbouring footholds,
  // v_x, v_y = their distance / t_swing_avg -- has constant coefficients: inside the KKT solve the four mid-node variables are
  // their linear image of the two footholds (`replaced`, recovery through rec_var / rec_col like the base's node values; the
  // proximal term of the replaced variables becomes static entries between the footholds).  A foot position that a Jacobian
  // block depends on is then a combination of footholds with one weight per slot and dimension (x, y folded; z as before).
  int reduce_swing = 0;
  std::vector<std::array<VecInW, NEE>> dyn_psol;   // per dynamics knot: the feet in solver space
  std::vector<VecInW> rom_psol;                    // per range-of-motion instance: its foot
  // projection of a starting point onto the rule: x[psw_var] = psw_w[2 i] x[psw_src[2 i]] + psw_w[2 i + 1] x[psw_src[2 i + 1]]
  std::vector<int> psw_var, psw_src;
  std::vector<double> psw_w;
  // Time keys of the elimination order (Symbolic orders the unknowns by var_time / con_time).  Rule 0 (rounds 1 - 5): a force node
  // at its node time, a B-spline coefficient at the knot in the middle of its support, a multiplier at its row's time.  Rule 1
  // (round 6, "late force nodes"): a force node half a polynomial LATER -- eliminated at its node time it pulls the dynamics rows
  // of the next polynomial (0.35 s: 48 multipliers) into the front at once --, the coefficients one base polynomial earlier, and
  // the multipliers of the FIRST dynamics knot behind every variable of their block: the base state is fixed at t = 0, and
  // with the force nodes later those six rows would be the first pivots, -eps alone under the block's largest entries (growth
  // 1e11 in the factors, KKT residuals of 1 .. 1e7: profiles/r06_experiments/order_keys.log).  With it the 100-knot walk and the
  // 200-knot transcription fit 96 slots instead of 112 at the same number of stages (-10 % per KKT launch), one KKT solve is
  // accurate to 2e-8 (4e-9 with rule 0).  Rule 2 (round 6, second half): rule 1 WITHOUT the late force nodes -- the early
  // coefficients and the guard alone.  Under rule 0 the multipliers of the second dynamics knot and of the first junction's
  // acceleration rows are eliminated behind ONE coefficient per dimension: a 12 x 12 block that rests on six coefficients and
  // whatever stance forces the gait has at t = 0 -- accurate on the walk, 1e-3 .. 7e-3 on short trot horizons
  // (profiles/r06_experiments/order_fuzz.log); with the coefficients one polynomial earlier two per dimension come first and
  // every transcription of the fuzz is accurate to 6e-9.  On a reduced base the planner builds rules 2 and 1 and keeps the smaller
  // front (qtos_planner.hip pick_order_rule; QTOS_ORDER forces one); rule 0 stays the order of the full-base systems.
  int order_rule = 0;
  bool order_late_force() const { return order_rule == 1; }
  bool order_early_coef() const { return order_rule == 1 || order_rule == 2; }
  bool order_guard() const { return order_rule == 1 || order_rule == 2; }
  mutable bool foot_sol_overflow = false;   // make_foot_sol met more than four (variable, weight) pairs in one dimension
  VecInW make_foot_sol(int e, double t) const {
    const VecIn in = make_in(eem[e], t, 0);
    VecInW o = widen(in);
    if (!reduce_swing) return o;
    for (int d = 0; d < 2; ++d) {
      // the (variable, weight) list of this dimension with every replaced mid-node variable expanded
      int vars[8], nv = 0;
      double ws[8];
      auto addv = [&](int v, double w) {
        if (v < 0 || w == 0.0) return;
        for (int i = 0; i < nv; ++i) if (vars[i] == v) { ws[i] += w; return; }
        vars[nv] = v; ws[nv] = w; ++nv;
      };
      for (int a = 0; a < 4; ++a) {
        const int v = in.var[3 * a + d];
        if (v < 0) continue;
        if (!replaced[v]) { addv(v, in.w[a]); continue; }
        for (size_t q = 0; q < psw_var.size(); ++q)
          if (psw_var[q] == v) { addv(psw_src[2 * q], in.w[a] * psw_w[2 * q]); addv(psw_src[2 * q + 1], in.w[a] * psw_w[2 * q + 1]); }
      }
      // (a VecInW has four slots per dimension; today's gaits give at most two pairs -- a schedule with several mid nodes per
      //  swing or neighbouring swings could give more: build() reports it instead of dropping Jacobian columns)
      if (nv > 4) foot_sol_overflow = true;
      for (int a = 0; a < 4; ++a) { o.var[3 * a + d] = a < nv ? vars[a] : -1; o.w[3 * a + d] = a < nv ? ws[a] : 0.0; }
    }
    return o;
  }

  static std::vector<double> time_grid(double T, double dt) {
    // towr time_discretization_constraint.cc: 0, dt, ..., floor(T/dt)*dt (accumulated), T
    std::vector<double> out;
    double t = 0;
    out.push_back(t);
    int steps = (int)std::floor(T / dt);
    for (int i = 0; i < steps; ++i) {
      t += dt;
      out.push_back(t);
    }
    out.push_back(T);
    return out;
  }

  // column list helper: assigns block columns to the free variables of an input group
  struct ColBuilder {
    std::vector<int> cols;
    const HostModel *M;
    short add(int var) {
      if (var < 0 || !M->is_unknown(var)) return -1;
      for (size_t i = 0; i < cols.size(); ++i)
        if (cols[i] == var) return (short)i;
      cols.push_back(var);
      return (short)(cols.size() - 1);
    }
    void group(const VecIn &in, short out[12]) {
      for (int i = 0; i < 12; ++i) out[i] = add(in.var[i]);
    }
    void group(const VecInW &in, short out[12]) {
      for (int i = 0; i < 12; ++i) out[i] = add(in.var[i]);
    }
  };

  // returns the block id; dynamic G offsets are assigned later, in stage order (Symbolic::build)
  int add_block(int kind, int m, int row0, const std::vector<int> &cols, bool gstatic,
                const double *gvals) {
    Block b;
    std::memset(&b, 0, sizeof(b));
    b.kind = kind;
    b.m = m;
    b.n = (int)cols.size();
    b.row0 = row0;
    b.col_off = (int)block_cols.size();
    b.gstatic = gstatic ? 1 : 0;
    b.goff = -1;
    if (gstatic) {
      b.goff = (int)g_static.size();
      g_static.insert(g_static.end(), gvals, gvals + (size_t)m * cols.size());
    }
    block_cols.insert(block_cols.end(), cols.begin(), cols.end());
    blocks.push_back(b);
    return (int)blocks.size() - 1;
  }
  std::vector<ColDesc> dyn_cols, rom_cols;
  int dyn_chunk = 1;   // dynamics knots per pass through the LDS scratch of the evaluation kernels (<= 128)
  // instances hold a block id in .goff until the offsets are final
  void finalize_goff() {
    auto fix = [&](int &goff) { if (goff >= 0) goff = blocks[goff].goff; };
    for (auto &i : dyn) fix(i.goff);
    for (auto &i : rom) fix(i.goff);
    for (auto &i : terr) fix(i.goff);
    for (auto &i : force) fix(i.goff);
    // column descriptors: every G entry of a dynamics / range-of-motion block is written once
    auto addw = [](std::vector<ColDesc> &cols, size_t first, int inst, int gbase0, int ncol, int kind,
                   const short cmap[12], const VecInW *in0) {
      for (int s = 0; s < 4; ++s)
        for (int d = 0; d < 3; ++d) {
          const int c = cmap[3 * s + d];
          if (c < 0) continue;
          ColDesc *cd = nullptr;
          for (size_t i = first; i < cols.size(); ++i)
            if (cols[i].gbase == gbase0 + c) { cd = &cols[i]; break; }
          if (!cd) {
            ColDesc nw;
            std::memset(&nw, 0, sizeof(nw));
            nw.inst = inst; nw.gbase = gbase0 + c; nw.ncol = ncol; nw.kind = (short)kind; nw.dim = (short)d;
            cols.push_back(nw);
            cd = &cols.back();
          }
          cd->w0 += in0->w[3 * s + d];
        }
    };
    auto add = [](std::vector<ColDesc> &cols, size_t first, int inst, int gbase0, int ncol, int kind,
                  const short cmap[12], const VecIn *in0, const VecIn *in1, const VecIn *in2) {
      for (int s = 0; s < 4; ++s)
        for (int d = 0; d < 3; ++d) {
          const int c = cmap[3 * s + d];
          if (c < 0) continue;
          ColDesc *cd = nullptr;
          for (size_t i = first; i < cols.size(); ++i)
            if (cols[i].gbase == gbase0 + c) { cd = &cols[i]; break; }
          if (!cd) {
            ColDesc nw;
            std::memset(&nw, 0, sizeof(nw));
            nw.inst = inst; nw.gbase = gbase0 + c; nw.ncol = ncol; nw.kind = (short)kind; nw.dim = (short)d;
            cols.push_back(nw);
            cd = &cols.back();
          }
          cd->w0 += in0->w[s];
          if (in1) cd->w1 += in1->w[s];
          if (in2) cd->w2 += in2->w[s];
        }
    };
    dyn_cols.clear();
    rom_cols.clear();
    for (size_t k = 0; k < dyn.size(); ++k) {
      const DynInst &I = dyn[k];
      if (!I.in_kkt) continue;
      const size_t first = dyn_cols.size();
      const BaseIn &Bi = dyn_sol[k];
      add(dyn_cols, first, (int)k, I.goff, I.ncol, 0, I.c_lin, &Bi.r, &Bi.a, nullptr);
      add(dyn_cols, first, (int)k, I.goff, I.ncol, 1, I.c_ang, &Bi.th, &Bi.thd, &Bi.thdd);
      for (int e = 0; e < NEE; ++e) {
        addw(dyn_cols, first, (int)k, I.goff, I.ncol, 2 + e, I.c_p[e], &dyn_psol[k][e]);
        add(dyn_cols, first, (int)k, I.goff, I.ncol, 6 + e, I.c_f[e], &I.f[e], nullptr, nullptr);
      }
    }
    for (size_t k = 0; k < rom.size(); ++k) {
      const RomInst &I = rom[k];
      const size_t first = rom_cols.size();
      const BaseIn &Bi = rom_sol[k];
      add(rom_cols, first, (int)k, I.goff, I.ncol, 0, I.c_lin, &Bi.r, nullptr, nullptr);
      add(rom_cols, first, (int)k, I.goff, I.ncol, 1, I.c_ang, &Bi.th, nullptr, nullptr);
      addw(rom_cols, first, (int)k, I.goff, I.ncol, 2, I.c_p, &rom_psol[k]);
    }
  }

  int build(const QtosParams &params) {
    P = params;
    T = 0;
    for (int k = 0; k < P.n_phases[0]; ++k) T += P.phase_dur[0][k];
    if (!(T > 0) || !(P.dt_base > 0) || !(P.dt_dyn > 0) || !(P.dt_rom > 0) ||
        P.force_polys_per_stance < 1) {
      err = "bad durations";
      return -1;
    }
    for (int e = 0; e < NEE; ++e) {
      int n = P.n_phases[e];
      if (n < 1 || n > QTOS_MAX_PHASES || n % 2 == 0) { err = "n_phases must be odd, <= QTOS_MAX_PHASES"; return -1; }
      double s = 0;
      for (int k = 0; k < n; ++k) s += P.phase_dur[e][k];
      if (std::fabs(s - T) > 1e-6) { err = "phase durations of all feet must sum to T"; return -1; }
    }
    // ---- base splines (towr parameters.cc GetBasePolyDurations) ----
    {
      double t_left = T;
      const double eps = 1e-10;
      while (t_left > eps) {
        lin.dur.push_back(t_left > P.dt_base ? P.dt_base : t_left);
        t_left -= P.dt_base;
      }
      lin.n_polys = (int)lin.dur.size();
      ang = lin;
    }
    const int nb = lin.n_polys;
    n_base_nodes = nb + 1;
    off_lin = 0;
    off_ang = 6 * (nb + 1);
    lin.idx.resize(nb + 1);
    ang.idx.resize(nb + 1);
    for (int k = 0; k <= nb; ++k)
      for (int q = 0; q < 2; ++q)
        for (int d = 0; d < 3; ++d) {
          lin.idx[k][q * 3 + d] = off_lin + 6 * k + 3 * q + d;
          ang.idx[k][q * 3 + d] = off_ang + 6 * k + 3 * q + d;
        }
    int off = 12 * (nb + 1);
    // ---- foot motion: stance = one constant polynomial, swing = two polynomials, free mid node
    //      with (px, vx, py, vy, pz); towr NodesVariablesEEMotion ----
    std::vector<int> stance_var[NEE];
    for (int e = 0; e < NEE; ++e) {
      Spline &S = eem[e];
      off_eem[e] = off;
      int v = off;
      std::array<int, 6> none;
      none.fill(-1);
      S.idx.push_back(none);
      for (int ph = 0; ph < P.n_phases[e]; ++ph) {
        double d = P.phase_dur[e][ph];
        if (ph % 2 == 0) {
          std::array<int, 6> nd = none;
          for (int dd = 0; dd < 3; ++dd) nd[dd] = v + dd;
          S.idx.back() = nd;
          S.idx.push_back(nd);
          S.dur.push_back(d);
          stance_var[e].push_back(v);
          v += 3;
        } else {
          std::array<int, 6> mid = none;
          mid[0] = v + 0; mid[3] = v + 1; mid[1] = v + 2; mid[4] = v + 3; mid[2] = v + 4;
          S.idx.push_back(mid);
          S.idx.push_back(none);  // start of the next stance, filled by that stance
          S.dur.push_back(d / 2);
          S.dur.push_back(d / 2);
          v += 5;
        }
      }
      S.n_polys = (int)S.dur.size();
      off = v;
    }
    // ---- foot force: stance = force_polys polynomials, swing = zero; nodes touching a swing
    //      polynomial are constant zero; towr NodesVariablesEEForce ----
    std::vector<int> fnode[NEE], fnode_stance[NEE];
    for (int e = 0; e < NEE; ++e) {
      Spline &S = eef[e];
      off_eef[e] = off;
      std::vector<int> swing_poly, stance_of;
      for (int ph = 0; ph < P.n_phases[e]; ++ph) {
        double d = P.phase_dur[e][ph];
        if (ph % 2 == 0)
          for (int j = 0; j < P.force_polys_per_stance; ++j) {
            S.dur.push_back(d / P.force_polys_per_stance);
            swing_poly.push_back(0);
            stance_of.push_back(ph / 2);
          }
        else {
          S.dur.push_back(d);
          swing_poly.push_back(1);
          stance_of.push_back(-1);
        }
      }
      S.n_polys = (int)S.dur.size();
      int v = off;
      for (int node = 0; node <= S.n_polys; ++node) {
        bool constant = (node > 0 && swing_poly[node - 1]) || (node < S.n_polys && swing_poly[node]);
        std::array<int, 6> nd;
        nd.fill(-1);
        if (!constant) {
          for (int dd = 0; dd < 3; ++dd) { nd[dd] = v + 2 * dd; nd[3 + dd] = v + 2 * dd + 1; }
          fnode[e].push_back(node);
          fnode_stance[e].push_back(node < S.n_polys ? stance_of[node] : stance_of[node - 1]);
          v += 6;
        }
        S.idx.push_back(nd);
      }
      off = v;
    }
    n_vars = off;

    // ---- variable bookkeeping: node time (the later node sharing a variable wins), initial
    //      guess rule, fixed variables (towr nlp_formulation.cc Add*Bound; parameters.cc
    //      bounds_final_*: final lin pos {X,Y}, final lin vel / ang pos / ang vel {X,Y,Z}) ----
    var_time.assign(n_vars, 0.0);
    init.assign(n_vars, InitDesc{0, 0, 0, -1, 0.0});
    auto book = [&](const Spline &S, int set) {
      for (int node = 0; node <= S.n_polys; ++node) {
        double t = S.node_time(node);
        for (int q = 0; q < 2; ++q)
          for (int d = 0; d < 3; ++d) {
            int v = S.idx[node][q * 3 + d];
            if (v < 0) continue;
            var_time[v] = t;
            init[v].set = set;
            init[v].dim = d;
            init[v].is_vel = q;
            init[v].frac = (double)node / (double)S.n_polys;
          }
      }
    };
    book(lin, 0);
    book(ang, 1);
    for (int e = 0; e < NEE; ++e) book(eem[e], 2 + e);
    for (int e = 0; e < NEE; ++e) book(eef[e], 6 + e);
    // (the keys below move: reduce_swing puts a foothold's key at the end of the swing behind its stance, order rule 1 the force
    //  nodes' half a polynomial later -- the node times themselves are kept for the time-shifted warm start.  Up to round 5
    //  k_shift_warm read the keys: a foothold in front of a swing was sampled at the START of the next stance, i.e. it got the
    //  next foothold's position)
    node_time = var_time;
    if (order_late_force())   // late force nodes: half a polynomial behind the node's time
      for (int e = 0; e < NEE; ++e) {
        const Spline &S = eef[e];
        for (int node = 0; node <= S.n_polys; ++node) {
          const double dn = S.dur[std::min(node, S.n_polys - 1)];
          for (int i = 0; i < 6; ++i)
            if (S.idx[node][i] >= 0) var_time[S.idx[node][i]] = S.node_time(node) + 0.5 * dn;
        }
      }
    for (int d = 0; d < 3; ++d) {
      init[off_lin + d].fix_src = d;                                   // start CoM
      init[off_lin + 3 + d].fix_src = P.honor_start_velocity ? 18 + d : 26;
      init[off_ang + d].fix_src = 3 + d;                               // start Euler
      init[off_ang + 3 + d].fix_src = P.honor_start_velocity ? 21 + d : 26;
      if (d < 2) init[off_lin + 6 * nb + d].fix_src = 24 + d;          // goal x, y
      init[off_lin + 6 * nb + 3 + d].fix_src = 26;
      init[off_ang + 6 * nb + d].fix_src = 26;
      init[off_ang + 6 * nb + 3 + d].fix_src = 26;
      for (int e = 0; e < NEE; ++e) init[off_eem[e] + d].fix_src = 6 + 3 * e + d;  // start feet
    }

    // ---- solver variables ----
    // (at least 20 base polynomials: on very short horizons -- 6 polynomials, both double knots next to the ends -- the
    //  factorisation without pivoting of the reduced system lost digits, 1e-2 residual of a solve; nothing is gained there)
    reduce_base = P.reduce_base != 0 && nb >= 20;
    if (reduce_base) {
      // The reduction needs towr's straight-line starting point to be a spline of the coefficients' space: C2 at every
      // interior junction but the first and the last (those two are double knots).  Node positions start + node / n_polys *
      // (goal - start) and node velocities (goal - start) / T are that exactly when the base polynomials have equal
      // durations; with a shorter last polynomial (T not a multiple of dt_base) the guess has an acceleration jump at every
      // junction and the full system is kept.
      for (int j = 1; j + 2 < nb && reduce_base; ++j) {   // junction j + 1 between polynomials j and j + 1, no fixed end state involved
        const VecIn prev = make_in_poly(lin, j, lin.dur[j], 2), next = make_in_poly(lin, j + 1, 0.0, 2);
        double kappa = 0.0, scale = 0.0;
        for (int a = 0; a < 4; ++a) {
          const double fr_p = (double)(j + (a >> 1)) / nb, fr_n = (double)(j + 1 + (a >> 1)) / nb;
          kappa += prev.w[a] * ((a & 1) ? 1.0 / T : fr_p) - next.w[a] * ((a & 1) ? 1.0 / T : fr_n);
          scale = std::max(scale, std::fabs(prev.w[a]) * ((a & 1) ? 1.0 / T : 1.0));
        }
        if (std::fabs(kappa) > 1e-9 * scale) reduce_base = 0;
      }
    }
    n_sol = n_vars;
    n_coef = 0;
    replaced.assign(n_vars, 0);
    sol_diag.assign(n_vars, P.delta_x);
    std::vector<std::vector<std::pair<int, double>>> quad;   // per recovered node value: (solver column, weight)
    if (reduce_base) {
      // clamped knots: t_0 x 4, the interior node times, t_nb x 4; coefficient j lives on the polynomials j-3 .. j
      // The first and the last interior junction are DOUBLE knots (C1 there): towr's starting point -- constant node
      // velocities, but the end velocities fixed -- has an acceleration jump exactly there, and the iterate has to stay
      // representable.  The continuity rows of those two junctions stay in the KKT system (in the coefficients' space, see
      // the constant-coefficient rows below); all the others hold identically.
      knots.assign(3, 0.0);
      for (int k = 0; k <= nb; ++k) {
        knots.push_back(lin.node_time(k));
        if (k == 1 || k == nb - 1) knots.push_back(lin.node_time(k));
      }
      for (int i = 0; i < 3; ++i) knots.push_back(T);
      span_of.assign(nb, 0);
      for (int k = 0, i = 3; k < nb; ++k) {
        while (i + 1 < (int)knots.size() && knots[i + 1] <= lin.node_time(k) + 1e-12) ++i;   // last copy of the interval's left knot
        span_of[k] = i;
      }
      const int ncj = (int)knots.size() - 4;
      n_coef = 6 * ncj;
      n_sol = n_vars + n_coef;
      coef_unknown.assign(n_coef, 1);
      var_time.resize(n_sol, 0.0);
      sol_diag.assign(n_sol, 0.0);
      for (int v = 0; v < n_vars; ++v) sol_diag[v] = P.delta_x;
      for (int which = 0; which < 2; ++which) {
        const Spline &S = which ? ang : lin;
        for (int d = 0; d < 3; ++d) {
          cvar[which][d].assign(ncj, -1);
          for (int j = 0; j < ncj; ++j) {
            const int id = n_vars + (which * 3 + d) * ncj + j;
            cvar[which][d][j] = id;
            // elimination time: the knot in the middle of the coefficient's support -- each dynamics row then finds coefficients
            // of its polynomial eliminated in front of it (eliminated at the END of their support the multipliers of the rows
            // come first with pivots of -eps_dual: the factorisation without pivoting breaks down)
            // (order rule 1: one base polynomial earlier)
            var_time[id] = knots[std::min(j + 3, (int)knots.size() - 1)] - (order_early_coef() ? P.dt_base : 0.0);
          }
          // end states: p(t_0) = c_0, p'(t_0) = 3 (c_1 - c_0) / h_0; p(T) = c_last, p'(T) = 3 (c_last - c_last-1) / h_last.
          // A fixed position removes c_0 (c_last); a fixed velocity ties c_1 to c_0 (c_last-1 to c_last): no freedom if that
          // one is gone, the same unknown otherwise.
          auto ends = [&](int jp, int jv, int node) {
            const bool p_fixed = !is_free(S.idx[node][d]), v_fixed = !is_free(S.idx[node][3 + d]);
            if (p_fixed) { coef_unknown[cvar[which][d][jp] - n_vars] = 0; cvar[which][d][jp] = -1; }
            if (v_fixed) {
              coef_unknown[cvar[which][d][jv] - n_vars] = 0;
              cvar[which][d][jv] = cvar[which][d][jp];
            }
          };
          ends(0, 1, 0);
          ends(ncj - 1, ncj - 2, nb);
        }
        for (int k = 0; k <= nb; ++k)
          for (int q = 0; q < 2; ++q)
            for (int d = 0; d < 3; ++d) replaced[S.idx[k][q * 3 + d]] = 1;
      }
      // recovery dx_nodes = Z dc and the proximal term delta |Z dc|^2 over the free node values
      for (int which = 0; which < 2; ++which) {
        const Spline &S = which ? ang : lin;
        for (int k = 0; k <= nb; ++k) {
          const int span = span_of[std::min(k, nb - 1)] - 3;   // first coefficient of the polynomial that starts (ends, for the last node) here
          double ders[3][4];
          bspline_ders(knots, span + 3, S.node_time(k), ders);
          for (int q = 0; q < 2; ++q)
            for (int d = 0; d < 3; ++d) {
              const int v = S.idx[k][q * 3 + d];
              if (!is_free(v)) continue;
              std::vector<std::pair<int, double>> row;
              for (int a = 0; a < 4; ++a) {
                const int col = cvar[which][d][span + a];
                if (col < 0 || std::fabs(ders[q][a]) < 1e-14) continue;
                bool merged = false;
                for (auto &e : row) if (e.first == col) { e.second += ders[q][a]; merged = true; }
                if (!merged) row.push_back({col, ders[q][a]});
              }
              rec_var.push_back(v);
              for (int a = 0; a < 4; ++a) {
                rec_col.push_back(a < (int)row.size() ? row[a].first : -1);
                rec_w.push_back(a < (int)row.size() ? row[a].second : 0.0);
              }
              quad.push_back(row);
            }
        }
      }
      // projection tables
      {
        std::vector<int> poly_of_span(knots.size(), -1);
        for (int k = 0; k < nb; ++k) poly_of_span[span_of[k]] = k;
        for (int which = 0; which < 2; ++which) {
          const Spline &S = which ? ang : lin;
          for (int d = 0; d < 3; ++d)
            for (int j = 0; j < ncj; ++j) {
              int k = -1;
              for (int i : {j + 2, j + 1, j + 3, j})
                if (i >= 3 && i < (int)knots.size() && poly_of_span[i] >= 0) { k = poly_of_span[i]; break; }
              const double h = S.dur[k], t0 = S.node_time(k);
              const double s1 = knots[j + 1] - t0, s2 = knots[j + 2] - t0, s3 = knots[j + 3] - t0;
              const double m1 = (s1 + s2 + s3) / 3, m2 = (s1 * s2 + s1 * s3 + s2 * s3) / 3, m3 = s1 * s2 * s3;
              const double wq[4] = {1 - 3 * m2 / (h * h) + 2 * m3 / (h * h * h), m1 - 2 * m2 / h + m3 / (h * h),
                                    3 * m2 / (h * h) - 2 * m3 / (h * h * h), -m2 / h + m3 / (h * h)};
              const int vq[4] = {S.idx[k][d], S.idx[k][3 + d], S.idx[k + 1][d], S.idx[k + 1][3 + d]};
              for (int a = 0; a < 4; ++a) { pc_var.push_back(vq[a]); pc_w.push_back(wq[a]); }
            }
          for (int k = 0; k <= nb; ++k) {
            const int span = span_of[std::min(k, nb - 1)] - 3;
            double ders[3][4];
            bspline_ders(knots, span + 3, S.node_time(k), ders);
            for (int q = 0; q < 2; ++q)
              for (int d = 0; d < 3; ++d) {
                const int v = S.idx[k][q * 3 + d];
                if (!is_free(v)) continue;
                pz_var.push_back(v);
                for (int a = 0; a < 4; ++a) {
                  pz_col.push_back((which * 3 + d) * ncj + span + a);
                  pz_w.push_back(ders[q][a]);
                }
              }
          }
        }
      }
    }
    // ---- reduced swings: the mid node's x, y, v_x, v_y as the linear image of the neighbouring footholds ----
    // (nearest-cell terrain only: a bilinear heightfield puts the slopes dh/dx, dh/dy of the swing's terrain row on the mid
    //  node's x, y)
    reduce_swing = P.reduce_swing != 0 && P.terrain_mode == 1;
    if (reduce_swing) {
      for (int e = 0; e < NEE; ++e) {
        const Spline &S = eem[e];
        for (int node = 1; node < S.n_polys; ++node) {
          if (S.idx[node][3] < 0) continue;   // (a mid node has velocities)
          for (int d = 0; d < 2; ++d) {
            const int ip = S.idx[node - 1][d], in = S.idx[node + 1][d], ic = S.idx[node][d], iv = S.idx[node][3 + d];
            const int vv[2] = {ic, iv};
            const double ww[2][2] = {{0.5, 0.5}, {-1.0 / P.t_swing_avg, 1.0 / P.t_swing_avg}};
            for (int q = 0; q < 2; ++q) {
              replaced[vv[q]] = 1;
              psw_var.push_back(vv[q]);
              psw_src.push_back(ip); psw_src.push_back(in);
              psw_w.push_back(ww[q][0]); psw_w.push_back(ww[q][1]);
              std::vector<std::pair<int, double>> row;
              if (is_free(ip)) row.push_back({ip, ww[q][0]});
              if (is_free(in)) row.push_back({in, ww[q][1]});
              rec_var.push_back(vv[q]);
              for (int a = 0; a < 4; ++a) {
                rec_col.push_back(a < (int)row.size() ? row[a].first : -1);
                rec_w.push_back(a < (int)row.size() ? row[a].second : 0.0);
              }
              quad.push_back(row);
            }
            // the foothold in front of the swing now feeds the rows of the whole swing (it used to end with its stance): its place
            // in the elimination order moves to the end of the swing -- eliminated at the end of its stance, every unknown of the
            // swing's rows would enter the front with it (long swings: fronts of 288 slots instead of 208)
            if (ip >= 0) var_time[ip] = std::max(var_time[ip], S.node_time(node + 1));
          }
        }
      }
    }
    if (n_sol > (int)sol_diag.size()) sol_diag.resize(n_sol, 0.0);
    {
      // the proximal term delta |dx|^2 over the recovered node values in the unknowns they are recovered from
      std::vector<std::pair<std::pair<int, int>, double>> acc;
      for (auto &row : quad)
        for (size_t i = 0; i < row.size(); ++i)
          for (size_t j = 0; j <= i; ++j) {
            const double val = P.delta_x * row[i].second * row[j].second;
            if (row[i].first == row[j].first) { sol_diag[row[i].first] += val; continue; }
            const std::pair<int, int> key{std::min(row[i].first, row[j].first), std::max(row[i].first, row[j].first)};
            bool found = false;
            for (auto &e : acc) if (e.first == key) { e.second += val; found = true; break; }
            if (!found) acc.push_back({key, val});
          }
      for (auto &e : acc) sym_static.push_back({e.first.first, e.first.second, e.second});
    }

    // ---- constraint layout (logs/towr_log.out:112-129) ----
    t_dyn = time_grid(T, P.dt_dyn);
    t_rom = time_grid(T, P.dt_rom);
    int c = 0;
    for (int e = 0; e < NEE; ++e) { off_terrain[e] = c; c += eem[e].n_polys; }
    off_dyn = c; c += 6 * (int)t_dyn.size();
    off_acc_lin = c; c += 3 * (nb - 1);
    off_acc_ang = c; c += 3 * (nb - 1);
    for (int e = 0; e < NEE; ++e) { off_rom[e] = c; c += 3 * (int)t_rom.size(); }
    for (int e = 0; e < NEE; ++e) { off_force[e] = c; c += 5 * (int)fnode[e].size(); }
    for (int e = 0; e < NEE; ++e) { off_swing[e] = c; c += 4 * (P.n_phases[e] - 1) / 2; }
    n_cons = c;
    con_lo.assign(n_cons, 0.0);
    con_hi.assign(n_cons, 0.0);
    con_time.assign(n_cons, 0.0);
    row_kind.assign(n_cons, 1);

    // ---- terrain rows: nodes 1..N of each foot; stance rows are equalities, swing rows z >= h.
    //      Working set: the row of the fixed first stance is constant and the second node of a
    //      stance repeats the first, so one row per later stance is kept. ----
    for (int e = 0; e < NEE; ++e) {
      const Spline &S = eem[e];
      for (int node = 1; node <= S.n_polys; ++node) {
        int row = off_terrain[e] + node - 1;
        bool swing = S.idx[node][3] >= 0;
        con_time[row] = S.node_time(node);
        TerrInst ti;
        std::memset(&ti, 0, sizeof(ti));
        ti.vx = S.idx[node][0]; ti.vy = S.idx[node][1]; ti.vz = S.idx[node][2];
        ti.row = row;
        ColBuilder cb{{}, this};
        ti.cx = cb.add(ti.vx); ti.cy = cb.add(ti.vy); ti.cz = cb.add(ti.vz);
        ti.ncol = (int)cb.cols.size();
        bool dup = !swing && node >= 1 && S.idx[node - 1][0] == ti.vx;  // second node of a stance
        if (swing) { con_hi[row] = BIG; row_kind[row] = 2; }
        if (ti.ncol == 0 || dup) row_kind[row] = 0;
        ti.in_kkt = row_kind[row] != 0;
        ti.goff = -1;
        if (ti.in_kkt) ti.goff = add_block(swing ? 1 : 0, 1, row, cb.cols, false, nullptr);
        terr.push_back(ti);
      }
    }
    // ---- dynamics ----
    for (size_t k = 0; k < t_dyn.size(); ++k) {
      double t = t_dyn[k];
      DynInst di;
      std::memset(&di, 0, sizeof(di));
      di.r = make_in(lin, t, 0); di.a = make_in(lin, t, 2);
      di.th = make_in(ang, t, 0); di.thd = make_in(ang, t, 1); di.thdd = make_in(ang, t, 2);
      for (int e = 0; e < NEE; ++e) { di.p[e] = make_in(eem[e], t, 0); di.f[e] = make_in(eef[e], t, 0); }
      di.row0 = off_dyn + 6 * (int)k;
      BaseIn bi;
      bi.r = make_in_sol(0, lin, t, 0); bi.a = make_in_sol(0, lin, t, 2);
      bi.th = make_in_sol(1, ang, t, 0); bi.thd = make_in_sol(1, ang, t, 1); bi.thdd = make_in_sol(1, ang, t, 2);
      dyn_sol.push_back(bi);
      ColBuilder cb{{}, this};
      cb.group(bi.r, di.c_lin);
      cb.group(bi.th, di.c_ang);
      {
        std::array<VecInW, NEE> ps;
        for (int e = 0; e < NEE; ++e) ps[e] = make_foot_sol(e, t);
        dyn_psol.push_back(ps);
      }
      for (int e = 0; e < NEE; ++e) { cb.group(dyn_psol.back()[e], di.c_p[e]); cb.group(di.f[e], di.c_f[e]); }
      di.ncol = (int)cb.cols.size();
      // the grid repeats T when floor(T/dt)*dt == T: the repeated block is the same six equations
      bool dup = k > 0 && std::fabs(t - t_dyn[k - 1]) < 1e-9;
      di.in_kkt = !dup;
      // (order rule 1: the first knot's multipliers behind every variable of their block -- see order_rule)
      double t_row = t;
      if (order_guard() && k == 0)
        for (int c : cb.cols) t_row = std::max(t_row, var_time[c]);
      for (int i = 0; i < 6; ++i) {
        con_time[di.row0 + i] = t_row;
        if (dup) row_kind[di.row0 + i] = 0;
      }
      di.goff = -1;
      if (di.in_kkt) di.goff = add_block(0, 6, di.row0, cb.cols, false, nullptr);
      dyn.push_back(di);
    }
    // ---- acceleration continuity at the interior junctions (constant coefficients) ----
    for (int which = 0; which < 2; ++which) {
      const Spline &S = which ? ang : lin;
      int r0 = which ? off_acc_ang : off_acc_lin;
      for (int j = 0; j + 1 < nb; ++j) {
        VecIn prev = make_in_poly(S, j, S.dur[j], 2), next = make_in_poly(S, j + 1, 0.0, 2);
        for (int d = 0; d < 3; ++d) {
          LinRow lr;
          std::memset(&lr, 0, sizeof(lr));
          lr.row = r0 + 3 * j + d;
          auto addterm = [&](int var, double cf) {
            if (var < 0 || cf == 0.0) return;
            for (int i = 0; i < lr.n; ++i)
              if (lr.var[i] == var) { lr.coef[i] += cf; return; }
            lr.var[lr.n] = var; lr.coef[lr.n] = cf; lr.n++;
          };
          for (int a = 0; a < 4; ++a) { addterm(prev.var[3 * a + d], prev.w[a]); addterm(next.var[3 * a + d], -next.w[a]); }
          con_time[lr.row] = S.node_time(j + 1);
          linrow.push_back(lr);
        }
      }
    }
    // ---- range of motion ----
    for (int e = 0; e < NEE; ++e)
      for (size_t k = 0; k < t_rom.size(); ++k) {
        double t = t_rom[k];
        RomInst ri;
        std::memset(&ri, 0, sizeof(ri));
        ri.r = make_in(lin, t, 0); ri.th = make_in(ang, t, 0); ri.p = make_in(eem[e], t, 0);
        ri.ee = e;
        ri.row0 = off_rom[e] + 3 * (int)k;
        BaseIn bi;
        std::memset(&bi, 0, sizeof(bi));
        bi.r = make_in_sol(0, lin, t, 0); bi.th = make_in_sol(1, ang, t, 0);
        rom_sol.push_back(bi);
        ColBuilder cb{{}, this};
        rom_psol.push_back(make_foot_sol(e, t));
        cb.group(bi.r, ri.c_lin); cb.group(bi.th, ri.c_ang); cb.group(rom_psol.back(), ri.c_p);
        ri.ncol = (int)cb.cols.size();
        for (int d = 0; d < 3; ++d) {
          con_lo[ri.row0 + d] = P.nominal_stance[e][d] - P.max_dev[d];
          con_hi[ri.row0 + d] = P.nominal_stance[e][d] + P.max_dev[d];
          con_time[ri.row0 + d] = t;
          row_kind[ri.row0 + d] = 2;
        }
        ri.goff = add_block(1, 3, ri.row0, cb.cols, false, nullptr);
        rom.push_back(ri);
      }
    if (foot_sol_overflow) { err = "reduce_swing: a foot position depends on more than four footholds per dimension (VecInW has four slots): this schedule needs reduce_swing = 0"; return -1; }
    // ---- force: unilateral + friction pyramid at every optimised force node ----
    for (int e = 0; e < NEE; ++e)
      for (size_t j = 0; j < fnode[e].size(); ++j) {
        int node = fnode[e][j], sv = stance_var[e][fnode_stance[e][j]];
        ForceInst fi;
        std::memset(&fi, 0, sizeof(fi));
        for (int d = 0; d < 3; ++d) fi.vf[d] = eef[e].idx[node][d];
        fi.vsx = sv; fi.vsy = sv + 1;
        fi.row0 = off_force[e] + 5 * (int)j;
        ColBuilder cb{{}, this};
        for (int d = 0; d < 3; ++d) fi.cf[d] = cb.add(fi.vf[d]);
        fi.csx = cb.add(fi.vsx); fi.csy = cb.add(fi.vsy);
        fi.ncol = (int)cb.cols.size();
        double tt = eef[e].node_time(node);
        for (int r = 0; r < 5; ++r) { con_time[fi.row0 + r] = tt; row_kind[fi.row0 + r] = 2; }
        con_lo[fi.row0] = 0; con_hi[fi.row0] = P.f_max;
        con_lo[fi.row0 + 1] = -BIG; con_hi[fi.row0 + 1] = 0;
        con_lo[fi.row0 + 2] = 0; con_hi[fi.row0 + 2] = BIG;
        con_lo[fi.row0 + 3] = -BIG; con_hi[fi.row0 + 3] = 0;
        con_lo[fi.row0 + 4] = 0; con_hi[fi.row0 + 4] = BIG;
        fi.goff = add_block(1, 5, fi.row0, cb.cols, false, nullptr);
        force.push_back(fi);
      }
    // ---- swing: mid node xy = centre of the neighbouring stances, v_xy = distance / t_swing_avg
    for (int e = 0; e < NEE; ++e) {
      const Spline &S = eem[e];
      int row = off_swing[e];
      for (int node = 1; node < S.n_polys; ++node) {
        if (S.idx[node][3] < 0) continue;
        for (int d = 0; d < 2; ++d) {
          int ip = S.idx[node - 1][d], in = S.idx[node + 1][d], ic = S.idx[node][d], iv = S.idx[node][3 + d];
          LinRow a, b;
          std::memset(&a, 0, sizeof(a));
          std::memset(&b, 0, sizeof(b));
          a.row = row; a.n = 3;
          a.var[0] = ic; a.coef[0] = 1; a.var[1] = ip; a.coef[1] = -0.5; a.var[2] = in; a.coef[2] = -0.5;
          b.row = row + 1; b.n = 3;
          b.var[0] = iv; b.coef[0] = 1; b.var[1] = ip; b.coef[1] = 1 / P.t_swing_avg; b.var[2] = in; b.coef[2] = -1 / P.t_swing_avg;
          con_time[row] = con_time[row + 1] = S.node_time(node);
          linrow.push_back(a);
          linrow.push_back(b);
          row += 2;
        }
      }
    }
    // constant-coefficient rows become equality blocks with a shared (static) G
    for (const LinRow &lr : linrow) {
      std::vector<int> cols;
      std::vector<double> vals;
      bool on_base = false;
      for (int i = 0; i < lr.n; ++i) {
        if (replaced[lr.var[i]]) on_base = true;
        if (is_free(lr.var[i])) { cols.push_back(lr.var[i]); vals.push_back(lr.coef[i]); }
      }
      if (on_base) {
        // reduced base: the row in the coefficients' space, A Z.  It vanishes wherever the basis is C2 (the row leaves the KKT
        // system: still evaluated, its value is zero for every iterate) and stays at the two double knots.
        std::vector<int> ccols;
        std::vector<double> cvals;
        double amax = 0.0;
        for (size_t i = 0; i < cols.size(); ++i) {
          amax = std::max(amax, std::fabs(vals[i]));
          if (!replaced[cols[i]]) {   // (an unknown of its own, e.g. the footholds of a swing row)
            size_t at = 0;
            while (at < ccols.size() && ccols[at] != cols[i]) ++at;
            if (at == ccols.size()) { ccols.push_back(cols[i]); cvals.push_back(0.0); }
            cvals[at] += vals[i];
            continue;
          }
          for (size_t q = 0; q < rec_var.size(); ++q) {
            if (rec_var[q] != cols[i]) continue;
            for (int a = 0; a < 4; ++a) {
              const int col = rec_col[4 * q + a];
              if (col < 0) continue;
              size_t at = 0;
              while (at < ccols.size() && ccols[at] != col) ++at;
              if (at == ccols.size()) { ccols.push_back(col); cvals.push_back(0.0); }
              cvals[at] += vals[i] * rec_w[4 * q + a];
            }
          }
        }
        std::vector<int> kc;
        std::vector<double> kv;
        for (size_t i = 0; i < ccols.size(); ++i)
          if (std::fabs(cvals[i]) > 1e-9 * amax) { kc.push_back(ccols[i]); kv.push_back(cvals[i]); }
        if (kc.empty()) { row_kind[lr.row] = 0; continue; }
        add_block(0, 1, lr.row, kc, true, kv.data());
        continue;
      }
      if (cols.empty()) { row_kind[lr.row] = 0; continue; }
      add_block(0, 1, lr.row, cols, true, vals.data());
    }
    return 0;
  }
};

}  // namespace qtos
