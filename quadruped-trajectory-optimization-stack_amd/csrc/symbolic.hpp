// symbolic.hpp -- fixed-structure analysis of the condensed KKT system
//
//     K = [ delta I + Ji' S Ji   Je' ]        unknowns: free node variables + equality multipliers
//         [ Je                  -eps I ]
//
// The unknowns are ordered by the time stamp of their node / constraint, which makes K a
// variable-band ("skyline") matrix: long-lived unknowns (a stance foothold is coupled to every base
// node of its stance) sit at the END of their life so they only lengthen their own row.  The chain
// is cut into stages of PIV consecutive pivots; stage k's front = its pivots + every later unknown
// whose row reaches back into the eliminated range.  Each unknown is given ONE slot of the front
// (assembled entries in LDS, Schur updates in the MFMA accumulator registers of k_kkt) for its whole
// life, so Schur complements are applied in place and nothing is ever copied between stages (slots
// of eliminated pivots are recycled).
#pragma once
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <numeric>
#include <vector>

#include "env.hpp"
#include "model.hpp"

namespace qtos {

constexpr int PIV = 16;

// Which update wave holds which 16 x 16 tile of the Schur complement (k_kkt2 / k_kkt3): update index u holds the tiles
// t = u + NU i of the lower triangle, row by row.  Shared by the kernels and by the analysis.
#ifndef QTOS_NU112
#define QTOS_NU112 0
#endif
#ifndef QTOS_NU96
#define QTOS_NU96 0
#endif
constexpr int kkt_pick_nu(int ntile) {
  if (QTOS_NU112 > 0 && ntile == 28) return QTOS_NU112;   // (experiment: update waves of a 112-slot front)
  if (QTOS_NU96 > 0 && ntile == 21) return QTOS_NU96;     // (experiment: of a 96-slot front)
  // 96 slots (21 tiles; both benchmark gaits since round 6): twelve update waves -- nine with two tiles, three with one -- instead of
  // the eleven the rule below picks: -0.7 % per launch; 7, 9, 10, 13, 14 waves: +2 .. +8 % (profiles/r06_experiments/kkt96_tuning.log)
  if (ntile == 21) return 12;
  int best = 14, best_t = (ntile + 13) / 14;
  for (int nu = 14; nu >= 8; --nu) {
    const int t = (ntile + nu - 1) / nu;
    if (t < best_t || (t == best_t && nu * t - ntile <= best * best_t - ntile)) { best = nu; best_t = t; }
  }
  return ntile < 8 ? (ntile < 1 ? 1 : ntile) : best;
}
constexpr int kkt_tile_row(int t) { int R = 0; while (((R + 1) * (R + 2)) >> 1 <= t) ++R; return R; }

struct StageDesc {
  int n_active;            // pivots + border (for the algorithmic byte / flop count)
  int g_begin, g_len;      // this stage's slice of the per-problem Jacobian-block buffer G
  int ent_begin, ent_end;  // equality entries  (EqEntry)
  int rhs_begin, rhs_end;  // equality right-hand sides (EqRhs)
  int iq_begin, iq_end;    // inequality blocks (IqBlock)
};
// K[slot_r][slot_c] = G value (equality Jacobian entry = coupling multiplier <-> variable)
struct EqEntry {
  int src;   // >= 0: offset into the problem's G buffer; < 0: -(index+1) into g_static
  short slot_r, slot_c;
};
struct EqRhs {
  int row, slot;  // rhs[slot] = -g[row]
};
// inequality block: K[slot_a][slot_c] += sum_r sig_r G[r][a] G[r][c]; rhs[slot_a] -= sum_r G[r][a] w_r
struct IqBlock {
  int m, n, row0, gloc /* offset inside the stage's G slice */, slot_off, pad0, pad1, pad2;
};

// Kronecker structure of a range-of-motion block (Symbolic::kron): every column of G is a static multiple of a column of
// one of two 3 x 3 matrices -- G[i][col] = alpha_col M_mu[i][d], M_0 = R^T (base position and foot columns), M_1 = d(R^T (p - r))/d theta
// (kernels.hpp eval_rom, Symbolic::build_linear_terms) --, so G' S G [a][c] = rho_a rho_c T'[mu_a mu_c][d_a][d_c] with the 27
// entries T' = (three-term sums over a representative column per (mu, d)) and rho = alpha / alpha of the representative,
// and the right-hand side -G' w [a] = -rho_a V'[mu_a][d_a].  33 sums per block instead of 405 per block.
struct KMeta {
  bool ok = false;
  unsigned char mu[32], d[32], grp[32];   // per local column
  int rep[6];                             // representative column of (mu, d): index 3 mu + d
  double rho[12];                         // per group
  double alpha[32];                       // per local column (host-side checks)
  int n_grp = 0;
};

struct Symbolic {
  QtosEnv env;               // the environment switches of this analysis (parsed once by the caller: env.hpp)
  int n_unknowns = 0, n_free = 0, n_eq = 0, n_stages = 0, front = 0;
  bool kron = false;                 // range-of-motion blocks of the main records assembled through their Kronecker structure (QTOS_KRON)
  std::vector<KMeta> iq_kron;        // parallel to iq_blocks
  int max_kblocks = 0;               // most Kronecker blocks in one record (LDS scratch of the kernel: 33 doubles each)
  static constexpr int KRON_SM = 33, KRON_STRIDE = 22, KRON_MAXB = 15;   // (stride: two ints + ten doubles per block; + 4 ints for the two constants)
  // Short stages (round 4): stage boundaries need not fall on multiples of 16 unknowns.  Where a partition into stages of at
  // most 16 pivots exists whose largest front is a whole 16-slot group smaller than the uniform partition's at no more stages
  // (found by dynamic programming over the boundaries; the 100-knot walk: 112 slots instead of 128, two short stages), the
  // (applied to fronts above 128 slots, see shorten_stages) missing pivots of a short stage are DUMMIES -- order[] entry -1, unit pivot, no entries, zero right-hand side, a slot of
  // their own for that one stage -- and every array by position counts them: n_unknowns is then n_stages x 16 POSITIONS and
  // n_real_unknowns the unknowns of the KKT system.
  int n_real_unknowns = 0;
  // Pair mode (round 5, k_kkt5): two consecutive stages -- a PAIR (a, b) -- are eliminated behind one set of barriers.  Both
  // stages' pivots must hold slots for the whole pair: an unknown enters the front at the even stage of the pair of its first
  // coupling, the slots of a pair's pivots are released together at its end, the number of stages is even (an all-dummy stage
  // is appended if need be), and a pair has ONE record (32 pivot slots / diagonals, the entries, right-hand sides and
  // inequality blocks of both stages, one gather table whose targets are unique within the pair: two records assembled side
  // by side would race on the cells they share).  Panels, pivot-block inverses, masks and every per-position array keep the
  // 16-pivot stage as their unit: k_chord, the sweeps and k_residual do not change.
  bool pair_mode = false;
  int n_records = 0;                // records: one per stage, or one per pair (srec_off / drec_off have n_records + 1 entries)
  int rec_stages = 1;               // stages per record
  int tgt_shift = 12;               // a gather-table target word = (cell << tgt_shift) | first contribution
  std::vector<int> diag_pos;        // per position: stream position of its pivot diagonal (k_residual)
  std::vector<unsigned> pair_groups; // pair mode, per pair: bit g = some slot of the 16-slot group g is in use during the pair (k_kkt5 skips the matrix instructions of empty groups)
  std::vector<int> stage_dummies;   // per stage: dummy pivots in it
  bool short_stages = true;
  std::vector<int> order;      // position -> var index, or n_vars + row for a multiplier
  std::vector<int> var_pos;    // var -> position or -1
  std::vector<int> row_pos;    // row -> position or -1
  std::vector<int> var_slot;   // var -> front slot or -1 (fixed)
  std::vector<int> row_slot;   // equality row -> front slot or -1
  std::vector<int> piv_slot;   // n_stages*PIV: slot of each pivot (dummy pivots get free slots)
  std::vector<int> piv_unknown;// n_stages*PIV: order[] entry or -1 for a dummy pivot
  std::vector<double> piv_diag;// n_stages*PIV: delta_x, -eps_dual, or 1 (dummy)
  std::vector<StageDesc> stages;
  std::vector<EqEntry> eq_entries;
  std::vector<EqRhs> eq_rhs;
  std::vector<IqBlock> iq_blocks;
  std::vector<short> iq_slots;    // front slot of every column of every inequality block
  int max_stage_g = 0;            // longest G slice of a stage
  // Packed per-stage records consumed by k_kkt (each is ONE contiguous, coalesced read):
  //   static  srec[srec_off[k] ..]: n_ent, n_rhs, n_iq, hi, gather offset, n_tgt, number of continuation
  //           records, index of the first one in `cont`, piv_slot[16],
  //           tri index per equality entry, slot per rhs entry, gather table of the inequality blocks
  //           (target, first contribution), one packed int per contribution
  //   dynamic stream (per problem), drec_off[k] ..: piv_diag[16], equality values, -g of the rhs
  //           rows, then per inequality block G (m x n), sig (m), w (m)
  // pack_src[i] says where stream element i comes from: (kind << 28) | index, kind 0 G, 1 g_static,
  // 2 -g[row], 3 sig[row], 4 w[row], 5 piv_diag, 6 alignment padding
  std::vector<int> srec, srec_off, pack_src, drec_off, stage_hi;
  // Compact storage of the assembled entries (see compact_cells): cell 0 is a constant zero, cells
  // 1..front hold the assembled right-hand side by slot, the rest are handed out to the structural
  // entries of K for the stages between their assembly and the gathering of their pivot column.
  std::vector<unsigned short> ctab;   // n_stages x (front/16) x 64 lanes x 4: cell of panel entry (row 16R + lk + 4g, pivot column li), 0 = none
  int n_cells = 0;                    // length of the cell array (zero cell + rhs + entries)
  // cell_mode 2 (k_kkt2: records are assembled during the AB phase, next to the gathering of the previous
  // stage's columns): right-hand-side entries get cells of their own too (a slot changes hands at a stage
  // boundary, a cell does not), listed per pivot in rtab, and a cell is recycled one stage later
  std::vector<int> rhs_ptr, rhs_gpos, rhs_row;   // right-hand side of K by unknown position (DevPlan::rhs_ptr ...)
  // equality part of K by unknown position, both triangles (k_residual: K x without the factorisation): unknown p is coupled
  // with unknown kx_col[e] through the stream value at kx_pos[e], e in [kx_ptr[p], kx_ptr[p + 1])
  std::vector<int> kx_ptr, kx_col, kx_pos;
  int cell_mode = 2;
  std::vector<int> rtab;              // n_stages x 16: cell of the assembled right-hand side of every pivot (0 = none)
  std::vector<unsigned> amask;   // per stage 256 bits (8 words): front slots whose row of the factor panel V is stored (live, not a pivot of the stage)
  // substitution sweeps with a one-stage look-ahead (k_chord, backward pass of k_kkt2): the rows of V_k that belong to
  // the pivots of stage k+1 are applied by the wave that carries the chain, the other rows by everybody else
  std::vector<unsigned> amask2;  // amask without the slots of the next stage's pivots
  std::vector<int> nxt_pack;     // n_stages x 4: bytes m = 0..3 of word q: slot of pivot q + 4m of stage k+1 if its row of V_k is live, else 255
  // The linearisation kernels write straight into the stream: eq_pos maps the (virtual) G offset of
  // an equality-block entry to its stream position; inequality blocks are contiguous in the stream
  // (Block::goff = stream offset); rhs/sig/w positions per constraint row; constants (static
  // Jacobian values, pivot diagonals) are written once per problem at planner creation.
  std::vector<int> eq_pos, rhs_pos, sig_pos, w_pos, const_pos;
  std::vector<double> const_val;
  int max_srec = 0, max_drec = 0;
  long long g_doubles = 0;
  long long algorithmic_bytes = 0, flops = 0, envelope = 0;
  int max_active = 0;
  std::string err;

  // Jacobian entries of the dynamics / range-of-motion columns as linear forms over the local
  // Jacobians (see LinTerm1); the formulas are those of one 6-row (3-row) column:
  //   dyn  lin   : d g_ang / d r = -[sum f]x w0 ; d g_lin / d a = m w1 I          (loc 27..29 = sum f)
  //        ang   : rows 0-2 = A_th w0 + A_thd w1 + A_thdd w2 (loc 0.., 9.., 18..), rows 3-5 = 0
  //        foot e: d g_ang / d p_e = [f_e]x w0 (loc 30+3e), rows 3-5 = 0
  //        force e: d g_ang / d f_e = [r - p_e]x w0 (loc 42+3e) ; d g_lin / d f_e = -w0 I
  //   rom  lin / foot: -/+ R^T w0 ((R^T)[i][d] = loc[3d+i]) ; ang: loc[9+3i+d] w0
  std::vector<LinTerm1> dyn_t1, rom_t1;
  std::vector<LinTerm3> dyn_t3;
  std::vector<int> dyn_t1_off, dyn_t3_off;   // first entry of every knot chunk (+ end)
  std::vector<int> rom_t1_off;               // same for the chunks of range-of-motion instances
  int rom_chunk = 1;
  void build_linear_terms(HostModel &M) {
    const int n_dyn = (int)M.dyn.size(), n_ch = std::max(1, (n_dyn + 127) / 128);
    M.dyn_chunk = std::max(1, (n_dyn + n_ch - 1) / n_ch);
    std::vector<std::vector<LinTerm1>> t1(n_ch);
    std::vector<std::vector<LinTerm3>> t3(n_ch);
    auto skew = [](int i, int d, int &k, double &sgn) {   // [v]x[i][d] = sgn * v[k] (i != d)
      k = 3 - i - d;
      sgn = ((d - i + 3) % 3 == 1) ? -1.0 : 1.0;
    };
    for (const ColDesc &c : M.dyn_cols) {
      const int ch = c.inst / M.dyn_chunk, base = (c.inst - ch * M.dyn_chunk) * DYN_LOC, d = c.dim;
      for (int i = 0; i < 6; ++i) {
        const int pos = eq_pos[c.gbase + i * c.ncol];
        double cst = 0.0;
        bool is_const = true;
        if (i < 3 && c.kind == 1) {
          LinTerm3 t;
          t.pos = pos;
          t.off[0] = base + 3 * i + d; t.off[1] = base + 9 + 3 * i + d; t.off[2] = base + 18 + 3 * i + d;
          t.a[0] = c.w0; t.a[1] = c.w1; t.a[2] = c.w2;
          t3[ch].push_back(t);
          is_const = false;
        } else if (i < 3 && i != d) {
          int k; double sgn;
          skew(i, d, k, sgn);
          const int vec = c.kind == 0 ? 27 : (c.kind < 6 ? 30 + 3 * (c.kind - 2) : 42 + 3 * (c.kind - 6));
          t1[ch].push_back({pos, base + vec + k, (c.kind == 0 ? -sgn : sgn) * c.w0});
          is_const = false;
        } else if (i >= 3 && i - 3 == d) {
          if (c.kind == 0) cst = M.P.mass * c.w1;
          else if (c.kind >= 6) cst = -c.w0;
        }
        if (is_const) { const_pos.push_back(pos); const_val.push_back(cst); }
      }
    }
    dyn_t1.clear(); dyn_t3.clear(); dyn_t1_off.assign(1, 0); dyn_t3_off.assign(1, 0);
    for (int ch = 0; ch < n_ch; ++ch) {
      std::sort(t1[ch].begin(), t1[ch].end(), [](const LinTerm1 &a, const LinTerm1 &b) { return a.pos < b.pos; });
      std::sort(t3[ch].begin(), t3[ch].end(), [](const LinTerm3 &a, const LinTerm3 &b) { return a.pos < b.pos; });
      dyn_t1.insert(dyn_t1.end(), t1[ch].begin(), t1[ch].end());
      dyn_t3.insert(dyn_t3.end(), t3[ch].begin(), t3[ch].end());
      dyn_t1_off.push_back((int)dyn_t1.size());
      dyn_t3_off.push_back((int)dyn_t3.size());
    }
    // range-of-motion instances go through the LDS scratch in chunks of at most 512 (one chunk up to a 10 s
    // horizon; `-duration 20` has 1008 of them)
    const int n_rom = (int)M.rom.size(), n_chr = std::max(1, (n_rom + 511) / 512);
    rom_chunk = std::max(1, (n_rom + n_chr - 1) / n_chr);
    std::vector<std::vector<LinTerm1>> r1(n_chr);
    for (const ColDesc &c : M.rom_cols) {
      const int ch = c.inst / rom_chunk, base = (c.inst - ch * rom_chunk) * ROM_LOC, d = c.dim;
      for (int i = 0; i < 3; ++i) {
        const int pos = c.gbase + i * c.ncol;   // inequality block: contiguous in the stream
        if (c.kind == 1) r1[ch].push_back({pos, base + 9 + 3 * i + d, c.w0});
        else r1[ch].push_back({pos, base + 3 * d + i, (c.kind == 2 ? 1.0 : -1.0) * c.w0});
      }
    }
    rom_t1.clear();
    rom_t1_off.assign(1, 0);
    for (int ch = 0; ch < n_chr; ++ch) {
      std::sort(r1[ch].begin(), r1[ch].end(), [](const LinTerm1 &a, const LinTerm1 &b) { return a.pos < b.pos; });
      rom_t1.insert(rom_t1.end(), r1[ch].begin(), r1[ch].end());
      rom_t1_off.push_back((int)rom_t1.size());
    }
  }

  // record limits: a stage record travels through the prefetch registers of k_kkt (2 x 16 B of doubles
  // and 3 x 16 B of ints per thread, 512 threads) and its gather codes address 4096 doubles
  static constexpr int SHDR_INTS = 8;
  // (set in build once the front is known: a front of up to 128 slots leaves LDS for 6144 ints / 2048 doubles
  //  per record; larger fronts -- three panels of up to 85 KB -- get 4096 / 1280 and spill the rest of a heavy
  //  stage into continuation records)
  int REC_MAX_DOUBLES = 2048, REC_MAX_INTS = 6144;
  int rec_cap_ints = 0;   // optional upper limit of a record's ints (qtos_planner_create retries with smaller records if the LDS budget is exceeded)
  std::vector<int> cont;   // continuation records: {srec offset, ints, stream offset, doubles} each
  // Kronecker structure of block bi if it belongs to a range-of-motion instance (the instances hold their block id in .goff
  // until HostModel::finalize_goff)
  KMeta kron_meta(const HostModel &M, int bi) const {
    KMeta K;
    const Block &b = M.blocks[bi];
    if (b.kind != 1 || b.m != 3 || b.n > 32 || M.reduce_swing) return K;   // (reduced swings: the foot's columns carry per-dimension weights)
    int inst = -1;
    for (size_t k = 0; k < M.rom.size(); ++k)
      if (M.rom[k].goff == bi) { inst = (int)k; break; }
    if (inst < 0) return K;
    const RomInst &I = M.rom[inst];
    const HostModel::BaseIn &Bi = M.rom_sol[inst];
    double alpha[32];
    int kind[32];
    for (int c = 0; c < b.n; ++c) { alpha[c] = 0.0; kind[c] = -1; K.d[c] = 0; K.mu[c] = 0; K.grp[c] = 0; }
    auto add = [&](const short cmap[12], const VecIn &in, int kd, double sgn) {
      for (int sidx = 0; sidx < 4; ++sidx)
        for (int d = 0; d < 3; ++d) {
          const int c = cmap[3 * sidx + d];
          if (c < 0 || c >= b.n) continue;
          if (kind[c] >= 0 && (kind[c] != kd || K.d[c] != d)) { kind[c] = 99; continue; }   // a column with two roles: no structure
          kind[c] = kd; K.d[c] = (unsigned char)d; K.mu[c] = kd == 1 ? 1 : 0;
          alpha[c] += sgn * in.w[sidx];
        }
    };
    add(I.c_lin, Bi.r, 0, -1.0);
    add(I.c_ang, Bi.th, 1, 1.0);
    add(I.c_p, I.p, 2, 1.0);
    for (int c = 0; c < b.n; ++c)
      if (kind[c] < 0 || kind[c] == 99 || alpha[c] == 0.0) return K;
    // groups: columns of one kind with one weight
    double galpha[12];
    int gkind[12];
    for (int c = 0; c < b.n; ++c) {
      int g = -1;
      for (int j = 0; j < K.n_grp; ++j)
        if (gkind[j] == kind[c] && galpha[j] == alpha[c]) { g = j; break; }
      if (g < 0) {
        if (K.n_grp == 12) return K;
        g = K.n_grp++;
        gkind[g] = kind[c]; galpha[g] = alpha[c];
      }
      K.grp[c] = (unsigned char)g;
    }
    // representatives: per mu the group of largest |alpha| with all three dimensions among the block's columns
    double ralpha[2] = {0.0, 0.0};
    for (int i = 0; i < 6; ++i) K.rep[i] = -1;
    for (int mu = 0; mu < 2; ++mu) {
      int best = -1;
      for (int g = 0; g < K.n_grp; ++g) {
        if ((gkind[g] == 1 ? 1 : 0) != mu) continue;
        bool have[3] = {false, false, false};
        for (int c = 0; c < b.n; ++c)
          if (K.grp[c] == g) have[K.d[c]] = true;
        if (have[0] && have[1] && have[2] && (best < 0 || std::fabs(galpha[g]) > std::fabs(galpha[best]))) best = g;
      }
      bool any = false;
      for (int g = 0; g < K.n_grp; ++g) any |= (gkind[g] == 1 ? 1 : 0) == mu;
      if (best < 0) { if (any) return K; continue; }   // (columns of this mu without a full representative: old path)
      ralpha[mu] = galpha[best];
      for (int c = 0; c < b.n; ++c)
        if (K.grp[c] == best) K.rep[3 * mu + K.d[c]] = c;
    }
    // the representatives' groups have rho = 1 and no table entry (index 15); the others are renumbered 0 .. 9
    int repg[2] = {-1, -1};
    for (int mu = 0; mu < 2; ++mu)
      if (K.rep[3 * mu] >= 0) repg[mu] = K.grp[K.rep[3 * mu]];
    int renum[12], n_tab = 0;
    for (int g = 0; g < K.n_grp; ++g) renum[g] = (g == repg[0] || g == repg[1]) ? 15 : n_tab++;
    if (n_tab > 10) return K;
    for (int g = 0; g < 12; ++g) K.rho[g] = 0.0;
    for (int g = 0; g < K.n_grp; ++g)
      if (renum[g] != 15) K.rho[renum[g]] = galpha[g] / ralpha[gkind[g] == 1 ? 1 : 0];
    for (int c = 0; c < b.n; ++c) K.grp[c] = (unsigned char)renum[K.grp[c]];
    K.n_grp = n_tab;
    for (int c = 0; c < 32; ++c) K.alpha[c] = c < b.n ? alpha[c] : 0.0;
    K.ok = true;
    return K;
  }
  // host-side check of the structure (QTOS_DEBUG_KRON): random matrices and weights, every entry of G' S G and of G' w through
  // the 33 sums against the direct three-term sums; returns the largest relative difference
  double check_kron(int *n_blocks) const {
    double worst = 0.0;
    unsigned rng = 12345u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 65536.0 - 0.5; };
    *n_blocks = 0;
    for (size_t q = 0; q < iq_blocks.size(); ++q) {
      const KMeta &K = iq_kron[q];
      if (!K.ok) continue;
      ++*n_blocks;
      const int n = iq_blocks[q].n;
      double Mm[2][3][3], sg[3], w[3], G[3][32];
      for (auto &m2 : Mm) for (auto &r : m2) for (double &v : r) v = rnd();
      for (int i = 0; i < 3; ++i) { sg[i] = 1.0 + rnd(); w[i] = rnd(); }
      for (int i = 0; i < 3; ++i) for (int c = 0; c < n; ++c) G[i][c] = K.alpha[c] * Mm[K.mu[c]][i][K.d[c]];
      double T[33];
      for (int t = 0; t < 33; ++t) {
        int mu, nu, d, e;
        if (t < 9) { mu = nu = 0; d = t / 3; e = t % 3; } else if (t < 18) { mu = 0; nu = 1; d = (t - 9) / 3; e = (t - 9) % 3; }
        else if (t < 27) { mu = nu = 1; d = (t - 18) / 3; e = (t - 18) % 3; } else { mu = nu = (t - 27) / 3; d = e = (t - 27) % 3; }
        const int c1 = K.rep[3 * mu + d], c2 = K.rep[3 * nu + e];
        double acc = 0.0;
        for (int i = 0; i < 3; ++i) acc += t < 27 ? (c1 >= 0 && c2 >= 0 ? sg[i] * G[i][c1] * G[i][c2] : 0.0) : (c1 >= 0 ? G[i][c1] * w[i] : 0.0);
        T[t] = acc;
      }
      for (int a = 0; a < n; ++a)
        for (int c = 0; c <= a; ++c) {
          double direct = 0.0;
          for (int i = 0; i < 3; ++i) direct += sg[i] * G[i][a] * G[i][c];
          int t;
          if (K.mu[a] == K.mu[c]) t = (K.mu[a] ? 18 : 0) + 3 * K.d[a] + K.d[c];
          else t = 9 + (K.mu[a] == 0 ? 3 * K.d[a] + K.d[c] : 3 * K.d[c] + K.d[a]);
          const double viaK = (K.grp[a] == 15 ? 1.0 : K.rho[K.grp[a]]) * (K.grp[c] == 15 ? 1.0 : K.rho[K.grp[c]]) * T[t];
          worst = std::max(worst, std::fabs(viaK - direct) / (std::fabs(direct) + 1e-30));
        }
      for (int a = 0; a < n; ++a) {
        double direct = 0.0;
        for (int i = 0; i < 3; ++i) direct += G[i][a] * w[i];
        const double viaK = (K.grp[a] == 15 ? 1.0 : K.rho[K.grp[a]]) * T[27 + 3 * K.mu[a] + K.d[a]];
        worst = std::max(worst, std::fabs(viaK - direct) / (std::fabs(direct) + 1e-30));
      }
    }
    return worst;
  }
  // dynamic part (per block G, sig, w) and gather table of the blocks `blks` of stage k, appended to the
  // record that starts at srec[s0] / pack_src[d0]; patches the record's header ints [4], [5]
  // the longest prefix of `blks` that fits a record which already holds `dyn` doubles and `fixed` ints
  template <class TRS>
  int split_blocks(const StageDesc &S, const std::vector<int> &blks, int dyn, int fixed, TRS trs,
                   std::vector<int> &mine, std::vector<int> &rest, int reserve = 0) {
    // (reserve: static contributions emitted with this record -- one double, one contribution and at most one target each)
    std::set<int> targets;
    int contrib = reserve;
    dyn += reserve;
    fixed += reserve;
    for (int q : blks) {
      const IqBlock &Q = iq_blocks[S.iq_begin + q];
      if (Q.m > 5 || Q.n > 32) { err = "inequality block too large for the packed gather records"; return -1; }
      const int d = Q.m * Q.n + 2 * Q.m, c = Q.n * (Q.n + 1) / 2 + Q.n;
      std::set<int> t2 = targets;
      for (int a = 0; a < Q.n; ++a) {
        const int sa = iq_slots[Q.slot_off + a];
        for (int cc = 0; cc <= a; ++cc) t2.insert(trs(sa, iq_slots[Q.slot_off + cc]));
        t2.insert(front * (front + 1) / 2 + sa);
      }
      const int kints = kron ? 6 + KRON_STRIDE * ((int)mine.size() + 1) : 0;   // (the Kronecker section, if every block had one)
      bool fits = rest.empty() && dyn < 4096 && dyn + d <= REC_MAX_DOUBLES - 2 && contrib + c < (1 << tgt_shift) &&
                  fixed + (int)t2.size() + 1 + contrib + c + kints <= REC_MAX_INTS - 8;
      if (fits) { mine.push_back(q); dyn += d; contrib += c; targets.swap(t2); }
      else rest.push_back(q);
    }
    return 0;
  }
  std::vector<std::array<int, 3>> sym_pp;   // static contributions as emitted: {position a, position b, stream position of the value}
  template <class TRS>
  int emit_blocks(int k, const StageDesc &S, const std::vector<int> &blks, int s0, int d0, TRS trs,
                  const std::vector<std::array<int, 6>> *syms = nullptr) {
    (void)k;
    std::vector<int> blk_goff(blks.size());
    for (size_t bi = 0; bi < blks.size(); ++bi) {
      const IqBlock &Q = iq_blocks[S.iq_begin + blks[bi]];
      blk_goff[bi] = (int)pack_src.size() - d0;
      if (blk_goff[bi] >= 4096) { err = "inequality block beyond the reach of the packed gather records"; return -1; }
      for (int i = 0; i < Q.m * Q.n; ++i) pack_src.push_back(S.g_begin + Q.gloc + i);
      for (int r = 0; r < Q.m; ++r) pack_src.push_back((3 << 28) | (Q.row0 + r));
      for (int r = 0; r < Q.m; ++r) pack_src.push_back((4 << 28) | (Q.row0 + r));
    }
    // gather table: target entry -> contributions (bi << 16 | a << 8 | c; c = 255: right-hand-side
    // contribution of column a).  One thread owns one target, so the blocks of a record are assembled
    // in ONE pass without conflicts and in a fixed order.
    std::map<int, std::vector<int>> tmap;
    for (size_t bi = 0; bi < blks.size(); ++bi) {
      const IqBlock &Q = iq_blocks[S.iq_begin + blks[bi]];
      for (int a = 0; a < Q.n; ++a) {
        const int sa = iq_slots[Q.slot_off + a];
        for (int c = 0; c <= a; ++c) tmap[trs(sa, iq_slots[Q.slot_off + c])].push_back(((int)bi << 16) | (a << 8) | c);
        tmap[front * (front + 1) / 2 + sa].push_back(((int)bi << 16) | (a << 8) | 255);
      }
    }
    // static contributions: code = offset of the value in the dynamic record | 62 << 18 (a one-by-one "block")
    if (syms)
      for (const auto &e : *syms) {
        const int off = (int)pack_src.size() - d0;
        if (off >= 4096) { err = "static entry beyond the reach of the packed gather records"; return -1; }
        sym_pp.push_back({e[3], e[4], (int)pack_src.size()});
        pack_src.push_back((1 << 28) | e[2]);
        tmap[trs(e[0], e[1])].push_back(-(off + 1));   // (marked: rewritten below)
      }
    srec[s0 + 4] = (int)srec.size() - s0;
    srec[s0 + 5] = (int)tmap.size();
    // one int per target (tri << 12 | first contribution), then one self-contained int per
    // contribution: offset of the block's G in the dynamic record (12 bits) | a << 12 | c << 18
    // (c = 63: right-hand side) | (n - 1) << 24 | (m - 1) << 29
    int cpos = 0;
    std::vector<int> codes;
    // targets in ascending order of their number of contributions: the 64 targets of a wave then loop equally long (a
    // wave takes as long as its longest lane), and the waves that take the low item indices -- in k_kkt2 the ones
    // without Schur tiles, which share the factor wave's SIMD -- get the short ones (measured: unsorted 1.136 ms per
    // launch, descending 1.130, ascending 1.110).  The order of the sums of a target does not change.
    std::vector<std::pair<int, std::vector<int>>> tlist(tmap.begin(), tmap.end());
    std::stable_sort(tlist.begin(), tlist.end(), [](const auto &a, const auto &b) { return a.second.size() < b.second.size(); });
    // Kronecker blocks of this record (main records only): index among them, or -1
    std::vector<int> kidx(blks.size(), -1);
    int n_k = 0;
    if (kron && syms)
      for (size_t bi = 0; bi < blks.size(); ++bi)
        if (iq_kron[S.iq_begin + blks[bi]].ok && n_k < KRON_MAXB) kidx[bi] = n_k++;
    for (auto &kv : tlist) {
      srec.push_back((kv.first << tgt_shift) | cpos);
      cpos += (int)kv.second.size();
      std::vector<int> kc, oc;   // a target's Kronecker contributions come first (the kernel runs them as a loop of their own)
      for (int code : kv.second) {
        if (code < 0) { oc.push_back((-code - 1) | (62 << 18)); continue; }   // static contribution
        const int bi = code >> 16, a = (code >> 8) & 255, c = code & 255;
        const IqBlock &Q = iq_blocks[S.iq_begin + blks[bi]];
        if (kidx[bi] >= 0) {
          // [0:6) entry of the block's 33 sums | [6:10) block | [12:18) group of a | 61 << 18 | [24:28) group of c
          const KMeta &K = iq_kron[S.iq_begin + blks[bi]];
          int tidx, gc = 0;
          if (c == 255) tidx = 27 + 3 * K.mu[a] + K.d[a];
          else {
            gc = K.grp[c];
            if (K.mu[a] == K.mu[c]) tidx = (K.mu[a] ? 18 : 0) + 3 * K.d[a] + K.d[c];
            else tidx = 9 + (K.mu[a] == 0 ? 3 * K.d[a] + K.d[c] : 3 * K.d[c] + K.d[a]);
          }
          // [0:9) the sum | [9:17) rho of a | 61 << 18 | [24:32) rho of c -- indices into the record's array of weights: block
          // kb's ten at 10 kb, then +1 (a representative's weight) and -1 (as "rho of c" of a right-hand-side entry: -rho_a V')
          const int one = 10 * n_k, ra = K.grp[a] == 15 ? one : 10 * kidx[bi] + K.grp[a];
          const int rc = c == 255 ? one + 1 : (gc == 15 ? one : 10 * kidx[bi] + gc);
          kc.push_back((KRON_SM * kidx[bi] + tidx) | (ra << 9) | (61 << 18) | (int)((unsigned)rc << 24));
          continue;
        }
        oc.push_back(blk_goff[bi] | (a << 12) | ((c == 255 ? 63 : c) << 18) | ((Q.n - 1) << 24) | (int)((unsigned)(Q.m - 1) << 29));
      }
      codes.insert(codes.end(), kc.begin(), kc.end());
      codes.insert(codes.end(), oc.begin(), oc.end());
    }
    if (cpos >= (1 << tgt_shift)) { err = "gather table overflow"; return -1; }
    srec.push_back(cpos);   // sentinel: end of the last target's contributions
    srec.insert(srec.end(), codes.begin(), codes.end());
    if (n_k > 0) {
      // Kronecker section: per block KRON_STRIDE ints: offset of G in the dynamic record | (n - 1) << 12, the six representative
      // columns (5 bits each), ten doubles rho (group 15 = a representative's: 1).
      if ((srec.size() - s0) & 1) srec.push_back(0);
      const int koff = (int)srec.size() - s0;
      // two ints per block (offset of G in the dynamic record | (n - 1) << 12; the six representative columns, 5 bits each),
      // then the record's weights: ten doubles per block, +1, -1
      for (size_t bi = 0; bi < blks.size(); ++bi) {
        if (kidx[bi] < 0) continue;
        const KMeta &K = iq_kron[S.iq_begin + blks[bi]];
        const IqBlock &Q = iq_blocks[S.iq_begin + blks[bi]];
        srec.push_back(blk_goff[bi] | ((Q.n - 1) << 12));
        int reps = 0;
        for (int i = 0; i < 6; ++i) reps |= (K.rep[i] < 0 ? 0 : K.rep[i]) << (5 * i);
        srec.push_back(reps);
      }
      auto push_double = [&](double v) { int w[2]; std::memcpy(w, &v, 8); srec.push_back(w[0]); srec.push_back(w[1]); };
      for (size_t bi = 0; bi < blks.size(); ++bi) {
        if (kidx[bi] < 0) continue;
        const KMeta &K = iq_kron[S.iq_begin + blks[bi]];
        for (int g = 0; g < 10; ++g) push_double(K.rho[g]);
      }
      push_double(1.0); push_double(-1.0);
      kron_off_of_record[s0] = (n_k << 5) | (koff << 9);   // (header int [2]: blocks of the record | Kronecker blocks << 5 | offset of their section << 9)
      max_kblocks = std::max(max_kblocks, n_k);
    }
    return 0;
  }
  std::map<int, int> kron_off_of_record;   // record start -> offset of its Kronecker section (the header int [2] is set by the caller)

  // The assembled (original) entries of K waiting in LDS for their pivot column used to live in a dense
  // lower triangle over the front's slots ((F+1)(F+2)/2 doubles: 67 KB at F = 128), of which only a few
  // thousand are structurally non-zero at any time.  Here every structural entry (u, v) gets a cell for
  // the stages between its first assembly (record k <= stage of u) and the gathering of the pivot columns
  // of its earlier unknown u (AB phase of stage(u) - 1, which comes before the assembly of record
  // stage(u) + 1: cells are recycled from then on).  The records' targets are rewritten to cell numbers
  // and a per-stage table tells every lane of the panel construction which cell (if any) feeds its entry.
  // An entry between two pivots of one stage is delivered once, to the row with the larger pivot index.
  int compact_cells(const std::vector<int> &first, const std::vector<int> &slot_of) {
    const int F = front, ntri_rhs = F * (F + 1) / 2;
    std::vector<int> occ((size_t)n_stages * F, -1);
    for (int j = 0; j < n_unknowns; ++j) {
      // (pair mode: a slot is held from the even stage of the pair of the first coupling to the end of the pivot's pair)
      const int k0 = pair_mode ? (first[j] / PIV) & ~1 : first[j] / PIV, k1 = pair_mode ? (j / PIV) | 1 : j / PIV;
      for (int k = k0; k <= k1; ++k) occ[(size_t)k * F + slot_of[j]] = j;
    }
    std::map<std::pair<int, int>, int> cell;
    std::vector<std::vector<int>> retire(n_stages);
    std::vector<int> free_cells;
    n_cells = cell_mode == 2 ? 1 : 1 + F;
    const int RHS = 1 << 30;
    auto target = [&](int k, int t) -> int {
      if (t >= ntri_rhs && cell_mode != 2) return 1 + (t - ntri_rhs);
      if (t >= ntri_rhs) {
        const int u = occ[(size_t)k * F + (t - ntri_rhs)];
        if (u < 0) { err = "assembled right-hand side on a free slot"; return -1; }
        auto it = cell.find({u, RHS});
        if (it != cell.end()) return it->second;
        int id;
        if (!free_cells.empty()) { id = free_cells.back(); free_cells.pop_back(); }
        else id = n_cells++;
        cell[{u, RHS}] = id;
        retire[u / PIV].push_back(id);
        return id;
      }
      int r = 0;
      while ((r + 1) * (r + 2) / 2 <= t) ++r;
      const int c = t - r * (r + 1) / 2;
      int u = occ[(size_t)k * F + r], v = occ[(size_t)k * F + c];
      if (u < 0 || v < 0) { err = "assembled entry on a free slot"; return -1; }
      if (u > v) std::swap(u, v);
      if (u / PIV < k) { err = "entry assembled after its pivot column was gathered"; return -1; }
      auto it = cell.find({u, v});
      if (it != cell.end()) return it->second;
      int id;
      if (!free_cells.empty()) { id = free_cells.back(); free_cells.pop_back(); }
      else id = n_cells++;
      cell[{u, v}] = id;
      retire[u / PIV].push_back(id);
      return id;
    };
    auto rewrite = [&](int k, int s0) -> int {
      const int n_ent = srec[s0], n_rhs = srec[s0 + 1];
      int *e = &srec[s0 + SHDR_INTS + PIV * rec_stages];
      for (int i = 0; i < n_ent; ++i) if ((e[i] = target(k, e[i])) < 0) return -1;
      for (int i = 0; i < n_rhs; ++i) if ((e[n_ent + i] = target(k, ntri_rhs + e[n_ent + i])) < 0) return -1;
      int *tg = &srec[s0 + srec[s0 + 4]];
      const int tmask = (1 << tgt_shift) - 1;
      for (int t = 0; t < srec[s0 + 5]; ++t) {
        const int c = target(k, tg[t] >> tgt_shift);
        if (c < 0) return -1;
        tg[t] = (c << tgt_shift) | (tg[t] & tmask);
      }
      return 0;
    };
    ctab.assign((size_t)n_stages * F * PIV, 0);
    for (int rec = 0; rec < n_records; ++rec) {
      const int k = rec * rec_stages;
      const int lag = cell_mode == 2 ? 2 : 1;
      if (pair_mode) {
        // k_kkt5 assembles the record of pair q in the second half of pair q - 2; the columns of pair r were gathered -- their
        // cells retired -- in the first phase of pair r - 1: the record may re-use the cells of every pair before its own
        if (k >= 2) { for (int s = k - 2; s < k; ++s) for (int id : retire[s]) free_cells.push_back(id); }
      } else
      if (k >= lag) { for (int id : retire[k - lag]) free_cells.push_back(id); }
      std::sort(free_cells.begin(), free_cells.end(), std::greater<int>());   // lowest cell first
      if (rewrite(k, srec_off[rec])) return -1;
      const int n_cont = srec[srec_off[rec] + 6], c_first = srec[srec_off[rec] + 7];
      for (int c = 0; c < n_cont; ++c)
        if (rewrite(k, cont[4 * (c_first + c)])) return -1;
    }
    if (n_cells >= 65536) { err = "too many live entries for the 16-bit cell table"; return -1; }
    const int NT = F / 16;
    rtab.assign((size_t)n_stages * PIV, 0);
    for (auto &kv : cell) {
      if (kv.first.second == RHS) { rtab[kv.first.first] = kv.second; continue; }
      const int u = kv.first.first, v = kv.first.second, k = u / PIV, col = u - k * PIV, r = slot_of[v];
      const int R = r >> 4, rr = r & 15, lk = rr & 3, g = rr >> 2;
      ctab[((((size_t)k * NT + R) * 64) + lk * 16 + col) * 4 + g] = (unsigned short)kv.second;
    }
    n_cells = (n_cells + 1) & ~1;
    if (env.debug) fprintf(stderr, "qtos: %d cells for the assembled entries (dense triangle: %d)\n", n_cells, (F + 1) * (F + 2) / 2);
    return 0;
  }


  // see short_stages.  first: position -> smallest coupled position (rewritten in positions that count the dummies).
  void shorten_stages(std::vector<int> &first, std::vector<int> &block_minpos, int n_sol, int n_cons) {
    const int N = n_unknowns, NS0 = (N + PIV - 1) / PIV;
    std::vector<int> sf(first);
    std::sort(sf.begin(), sf.end());
    // unknowns alive in a stage of the positions [a, b): everyone coupled to a position below b, minus those eliminated before a
    auto alive = [&](int a, int b) { return (int)(std::lower_bound(sf.begin(), sf.end(), b) - sf.begin()) - a; };
    int F0 = 0;
    for (int k = 0; k < NS0; ++k) {
      const int a = k * PIV, b = std::min(N, a + PIV);
      F0 = std::max(F0, alive(a, b) + (PIV - (b - a)));
    }
    F0 = ((F0 + PIV - 1) / PIV) * PIV;
    // Measured: fronts above 128 slots gain even at 2 % more stages (round 4, profiles/r04_short_stages.txt: `-duration 20` 176 ->
    // 160 slots, 2.63 -> 1.81 ms per launch).  Up to 128 slots a smaller front pays only where it costs NO stage (round 4: the
    // walk 128 -> 112 slots at two more stages, 0.643 -> 0.656 ms; round 5, with the reduced swings: the trot 112 -> 96 slots at
    // the same 113 stages, 0.668 -> 0.592 ms per launch, the walk 128 -> 112 at the same 100: equal).  QTOS_SHORT_STAGES=1: the
    // 2 % rule for every front size; 0: never.
    if (env.short_stages == 0) return;
    const bool no_extra_stage = env.short_stages < 0 && F0 <= 128;
    std::vector<int> cut;   // boundaries of the chosen partition
    for (int target = F0 - PIV; target >= 2 * PIV; target -= PIV) {
      const int INF = 1 << 29;
      std::vector<int> best(N + 1, INF), prev(N + 1, -1);
      best[0] = 0;
      for (int a = 0; a < N; ++a) {
        if (best[a] >= INF) continue;
        for (int r = 1; r <= PIV && a + r <= N; ++r) {
          if (alive(a, a + r) + (PIV - r) > target) continue;
          // (fewest stages; among those the partition found first, i.e. the shortest stages as early as possible)
          if (best[a] + 1 < best[a + r]) { best[a + r] = best[a] + 1; prev[a + r] = a; }
        }
      }
      if (env.debug) fprintf(stderr, "qtos: short stages: a front of %d slots takes %d stages (%d at %d slots)\n", target, best[N], NS0, F0);
      if (best[N] > (no_extra_stage ? NS0 : NS0 + NS0 / 50)) break;
      cut.clear();
      for (int b = N; b > 0; b = prev[b]) cut.push_back(b);
      cut.push_back(0);
      std::reverse(cut.begin(), cut.end());
    }
    if (cut.empty()) return;
    const int NSv = (int)cut.size() - 1;
    std::vector<int> vmap(N), order_v((size_t)NSv * PIV, -1), first_v((size_t)NSv * PIV);
    stage_dummies.assign(NSv, 0);
    for (int k = 0; k < NSv; ++k) {
      for (int p = cut[k]; p < cut[k + 1]; ++p) vmap[p] = k * PIV + (p - cut[k]);
      stage_dummies[k] = PIV - (cut[k + 1] - cut[k]);
    }
    for (size_t v = 0; v < first_v.size(); ++v) first_v[v] = (int)v;   // (a dummy is coupled to nothing)
    for (int p = 0; p < N; ++p) {
      order_v[vmap[p]] = order[p];
      first_v[vmap[p]] = vmap[first[p]];
    }
    for (int v = 0; v < n_sol; ++v) if (var_pos[v] >= 0) var_pos[v] = vmap[var_pos[v]];
    for (int r = 0; r < n_cons; ++r) if (row_pos[r] >= 0) row_pos[r] = vmap[row_pos[r]];
    for (int &mp : block_minpos) if (mp >= 0 && mp < N) mp = vmap[mp];   // (position of an inequality block's earliest column)
    order.swap(order_v);
    first.swap(first_v);
    n_unknowns = NSv * PIV;
  }

  int build(HostModel &M) {
    // (solver variables: the model's variables, then -- QtosParams.reduce_base -- the B-spline coefficients that replace the
    //  base node values inside the solve, model.hpp)
    const int n = M.n_sol, m = M.n_cons;
    struct Key { double t; int id; };
    std::vector<Key> keys;
    for (int v = 0; v < n; ++v)
      if (M.is_unknown(v)) keys.push_back({M.var_time[v], v});
    n_free = (int)keys.size();
    // only rows that ended up in an equality block are multipliers
    std::vector<char> is_eq(m, 0);
    for (const Block &b : M.blocks)
      if (b.kind == 0)
        for (int r = 0; r < b.m; ++r) is_eq[b.row0 + r] = 1;
    for (int r = 0; r < m; ++r)
      if (is_eq[r]) keys.push_back({M.con_time[r] + 1e-7, n + r});
    n_unknowns = (int)keys.size();
    n_eq = n_unknowns - n_free;
    std::stable_sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
      return a.t < b.t || (a.t == b.t && a.id < b.id);
    });
    order.resize(n_unknowns);
    var_pos.assign(n, -1);
    row_pos.assign(m, -1);
    for (int i = 0; i < n_unknowns; ++i) {
      order[i] = keys[i].id;
      if (keys[i].id < n) var_pos[keys[i].id] = i;
      else row_pos[keys[i].id - n] = i;
    }
    // envelope: first[j] = smallest position coupled with j
    std::vector<int> first(n_unknowns);
    std::iota(first.begin(), first.end(), 0);
    // Pattern of K: an inequality block couples all of its columns with each other (Ji' S Ji);
    // an equality block couples each multiplier with the columns of its row and nothing else.
    std::vector<int> block_minpos(M.blocks.size(), 0);
    for (size_t bi = 0; bi < M.blocks.size(); ++bi) {
      const Block &b = M.blocks[bi];
      if (b.kind == 1) {
        int mn = n_unknowns;
        for (int a = 0; a < b.n; ++a) mn = std::min(mn, var_pos[M.block_cols[b.col_off + a]]);
        block_minpos[bi] = mn;
        for (int a = 0; a < b.n; ++a) {
          int p = var_pos[M.block_cols[b.col_off + a]];
          first[p] = std::min(first[p], mn);
        }
      } else {
        for (int r = 0; r < b.m; ++r) {
          const int pr = row_pos[b.row0 + r];
          for (int a = 0; a < b.n; ++a) {
            const int pa = var_pos[M.block_cols[b.col_off + a]];
            const int lo = std::min(pr, pa), hi = std::max(pr, pa);
            first[hi] = std::min(first[hi], lo);
          }
        }
      }
    }
    for (const HostModel::SymEntry &e : M.sym_static) {
      const int pa = var_pos[e.a], pb = var_pos[e.b];
      if (pa < 0 || pb < 0) continue;
      const int lo = std::min(pa, pb), hi = std::max(pa, pb);
      first[hi] = std::min(first[hi], lo);
    }
    n_real_unknowns = n_unknowns;
    if (short_stages) shorten_stages(first, block_minpos, n, m);
    n_stages = (n_unknowns + PIV - 1) / PIV;
    if (pair_mode) {
      // whole pairs: dummy positions fill the last stage and, if the number of stages is odd, one more stage
      const int want = ((n_stages + 1) & ~1) * PIV, have = n_unknowns;
      for (int v = n_unknowns; v < want; ++v) { order.push_back(-1); first.push_back(v); }
      n_unknowns = want;
      n_stages = n_unknowns / PIV;
      stage_dummies.resize(n_stages, 0);
      // (the fill of the last real stage counts as before -- active, like the dummies that filled it without pair mode --, an
      //  appended all-dummy stage is no work of the algorithm)
      for (int v = have; v < want; ++v) if (v / PIV > (have - 1) / PIV) stage_dummies[v / PIV]++;
      rec_stages = 2;
      tgt_shift = 13;
    }
    n_records = n_stages / rec_stages;
    stage_dummies.resize(n_stages, 0);
    // the stage at which an unknown enters the front
    auto enter_stage = [&](int j) { const int e = first[j] / PIV; return pair_mode ? (e & ~1) : e; };
    if (!env.dump_first.empty()) {   // diagnostic: envelope of the ordered matrix (position -> first coupled position, unknown id)
      if (FILE *f = fopen(env.dump_first.c_str(), "w")) {
        for (int j = 0; j < n_unknowns; ++j) fprintf(f, "%d %d %d\n", j, first[j], order[j]);
        fclose(f);
      }
    }
    // slot allocation
    var_slot.assign(n, -1);
    row_slot.assign(m, -1);
    std::vector<int> slot_of(n_unknowns, -1);
    piv_slot.assign((size_t)n_stages * PIV, -1);
    piv_unknown.assign((size_t)n_stages * PIV, -1);
    piv_diag.assign((size_t)n_stages * PIV, 1.0);
    stages.resize(n_stages);
    // unknowns sorted by entry stage: an unknown enters the front at stage first[j] / PIV
    std::vector<std::vector<int>> enter(n_stages);
    for (int j = 0; j < n_unknowns; ++j) enter[enter_stage(j)].push_back(j);
    std::vector<int> free_slots;  // kept sorted descending so pop_back gives the smallest
    std::vector<char> in_use;
    std::vector<int> slot_stage;   // pivot stage of the unknown occupying a slot, -1 if free
    stage_hi.assign(n_stages, 0);
    amask.assign((size_t)n_stages * 8, 0u);
    int n_slots = 0;
    max_active = 0;
    envelope = 0;
    for (int j = 0; j < n_unknowns; ++j) envelope += j - first[j] + 1;
    std::vector<int> active_count(n_stages, 0);
    int active = 0;
    // The front size is the largest number of simultaneously live unknowns whatever the placement rule
    // (a free slot is always reused), so it is known up front and ALL its slots are on offer from the
    // first stage on: that lets the rule below keep the pivots of a stage together.
    {
      int live = 0, peak = 0, held = 0;
      for (int k = 0; k < n_stages; ++k) {
        live += (int)enter[k].size();
        const int members = std::min(n_unknowns, (k + 1) * PIV) - k * PIV;
        peak = std::max(peak, live + (PIV - members));   // + the dummy pivots of a short last stage
        // (pair mode: the pivots of the even stage keep their slots until the pair is over)
        if (pair_mode && !(k & 1)) held = members; else { live -= members + held; held = 0; }
      }
      n_slots = ((peak + PIV - 1) / PIV) * PIV;
      for (int t = n_slots - 1; t >= 0; --t) free_slots.push_back(t);
      in_use.assign(n_slots, 0);
      slot_stage.assign(n_slots, -1);
    }
    for (int k = 0; k < n_stages; ++k) {
      for (int j : enter[k]) {
        // The kernel moves a stage's pivot columns out of the Schur tiles tile by tile: the fewer 16-slot
        // groups its pivots are spread over, the fewer tiles to visit.  Placement of an entering unknown:
        // a group that already hosts unknowns of its pivot stage (the one with most of them); else a
        // completely free group; else the group with the most free slots (room for the siblings still
        // to come); ties to the lower group, lowest free slot of the group.  (1.9 groups per stage on
        // the 100-knot problem; lowest-free-slot placement gives 3.7, same-stage preference alone 2.6.)
        int s = -1;
        if (!free_slots.empty()) {
          const int stage_j = pair_mode ? j / (2 * PIV) : j / PIV, n_grp = (n_slots + 15) / 16;   // (pair mode: the pair's 32 pivots together)
          int best_grp = -1, best_score = -1;
          for (int grp = 0; grp < n_grp; ++grp) {
            int same = 0, used = 0, nfree = 0;
            for (int t = grp * 16; t < grp * 16 + 16 && t < n_slots; ++t) {
              if (in_use[t]) { used++; if (slot_stage[t] == stage_j) same++; }
              else nfree++;
            }
            if (!nfree) continue;
            const int place = env.place;
            int score = same > 0 ? 2000 + same : (used == 0 ? 1000 : nfree);
            if (place == 1) score = 1000 - grp;                                   // lowest free slot
            if (place == 2) score = same > 0 ? 2000 + same : 1000 - grp;           // siblings, else lowest group
            if (place == 3) score = same > 0 ? 2000 + same : (used > 0 ? 1000 + nfree : 500 - grp);   // siblings, else the used group with most room, else lowest empty group
            if (place == 4) score = same > 0 ? 2000 + same : (used > 0 ? 1000 + 16 * (16 - grp) + nfree : 500 - grp);
            if (score > best_score) { best_score = score; best_grp = grp; }
          }
          for (int t = best_grp * 16; t < best_grp * 16 + 16; ++t)
            if (!in_use[t]) { s = t; break; }
          free_slots.erase(std::find(free_slots.begin(), free_slots.end(), s));
        }
        if (s < 0) s = n_slots++;
        slot_of[j] = s;
        if ((int)in_use.size() <= s) in_use.resize(s + 1, 0);
        if ((int)slot_stage.size() <= s) slot_stage.resize(s + 1, -1);
        in_use[s] = 1;
        slot_stage[s] = pair_mode ? j / (2 * PIV) : j / PIV;
        active++;
      }
      // dummy pivots of the (short) last stage need distinct unused slots
      int lo = k * PIV, hi = std::min(n_unknowns, lo + PIV);
      std::vector<int> dummies;
      for (int i = hi; i < lo + PIV; ++i) {
        int s;
        if (!free_slots.empty()) { s = free_slots.back(); free_slots.pop_back(); }
        else s = n_slots++;
        dummies.push_back(s);
        if ((int)in_use.size() <= s) in_use.resize(s + 1, 0);
        in_use[s] = 1;
      }
      {
        int hi = 0;
        for (int t = 0; t < (int)in_use.size(); ++t)
          if (in_use[t]) hi = t + 1;
        stage_hi[k] = ((hi + 3) / 4) * 4;
      }
      active_count[k] = active + (int)dummies.size();
      max_active = std::max(max_active, active_count[k]);
      for (int i = lo; i < lo + PIV; ++i) {
        size_t q = (size_t)k * PIV + (i - lo);
        if (i < hi) {
          piv_slot[q] = slot_of[i];
          piv_unknown[q] = order[i];
          piv_diag[q] = order[i] < 0 ? 1.0 : (order[i] < n ? M.sol_diag[order[i]] : -M.P.eps_dual);   // (a dummy pivot of a short stage: 1)
        } else {
          piv_slot[q] = dummies[i - hi];
          piv_unknown[q] = -1;
          piv_diag[q] = 1.0;
        }
      }
      // rows of the stage's factor panel that can be non-zero: the slots in use, without its own pivots
      for (int t = 0; t < (int)in_use.size() && t < 256; ++t)
        if (in_use[t]) amask[(size_t)k * 8 + (t >> 5)] |= 1u << (t & 31);
      for (int i = 0; i < PIV; ++i) {
        const int t = piv_slot[(size_t)k * PIV + i];
        if (t < 256) amask[(size_t)k * 8 + (t >> 5)] &= ~(1u << (t & 31));
      }
      if (pair_mode && (k & 1)) {   // (the even stage's pivots still hold their slots: not rows of this stage's panel)
        for (int i = 0; i < PIV; ++i) {
          const int t = piv_slot[(size_t)(k - 1) * PIV + i];
          if (t < 256) amask[(size_t)k * 8 + (t >> 5)] &= ~(1u << (t & 31));
        }
      }
      // release pivots and dummies (pair mode: both stages' at the end of the pair)
      if (!pair_mode || (k & 1)) {
        for (int kk = pair_mode ? k - 1 : k; kk <= k; ++kk)
          for (int i = 0; i < PIV; ++i) {
            const int t = piv_slot[(size_t)kk * PIV + i];
            free_slots.push_back(t);
            in_use[t] = 0;
            if ((int)slot_stage.size() > t) slot_stage[t] = -1;
          }
        std::sort(free_slots.begin(), free_slots.end(), std::greater<int>());
      }
      active -= (hi - lo);
    }
    if (pair_mode) {
      // the populated front of a stage as the algorithm counts it: its pivots and the later unknowns coupled to
      // an eliminated one -- not the slots the pair holds
      std::vector<int> fs(first.begin(), first.end());
      for (int &v : fs) v /= PIV;
      std::sort(fs.begin(), fs.end());
      for (int k = 0; k < n_stages; ++k)
        active_count[k] = (int)(std::upper_bound(fs.begin(), fs.end(), k) - fs.begin()) - k * PIV;
    }
    pair_groups.assign(pair_mode ? n_stages / 2 : 1, 0u);
    if (pair_mode)
      for (int k = 0; k < n_stages; ++k) {
        unsigned g = 0;
        for (int t = 0; t < 256; ++t) if ((amask[(size_t)k * 8 + (t >> 5)] >> (t & 31)) & 1u) g |= 1u << (t >> 4);
        for (int i = 0; i < PIV; ++i) g |= 1u << (piv_slot[(size_t)k * PIV + i] >> 4);
        pair_groups[k >> 1] |= g;
      }
    amask2 = amask;
    nxt_pack.assign((size_t)n_stages * 4, -1);
    for (int k = 0; k + 1 < n_stages; ++k)
      for (int i = 0; i < PIV; ++i) {
        const int t = piv_slot[(size_t)(k + 1) * PIV + i];
        const bool live = t < 256 && ((amask[(size_t)k * 8 + (t >> 5)] >> (t & 31)) & 1u);
        if (live) {
          amask2[(size_t)k * 8 + (t >> 5)] &= ~(1u << (t & 31));
          unsigned &w = (unsigned &)nxt_pack[(size_t)k * 4 + (i & 3)];
          w = (w & ~(0xffu << (8 * (i >> 2)))) | ((unsigned)t << (8 * (i >> 2)));
        }
      }
    front = ((n_slots + PIV - 1) / PIV) * PIV;
    if (front > 128 && cell_mode == 2) { REC_MAX_INTS = 4096; REC_MAX_DOUBLES = 1280; }
    // (the caller found the records too large for the LDS it has left: heavy stages spill into continuation records earlier)
    if (rec_cap_ints > 0) { REC_MAX_INTS = std::min(REC_MAX_INTS, rec_cap_ints); REC_MAX_DOUBLES = std::min(REC_MAX_DOUBLES, rec_cap_ints / 3); }
    for (int j = 0; j < n_unknowns; ++j) {
      if (order[j] < 0) continue;
      if (order[j] < n) var_slot[order[j]] = slot_of[j];
      else row_slot[order[j] - n] = slot_of[j];
    }
    // Inequality blocks are assembled at the stage of their earliest column; their G blocks are laid
    // out in that order so that each stage reads ONE contiguous slice of G.  An equality entry
    // K[multiplier][variable] is assembled at the stage that eliminates the earlier of the two
    // (both have a slot by then); equality G blocks follow the inequality slices in the buffer.
    std::vector<std::vector<int>> owned(n_stages);
    for (size_t bi = 0; bi < M.blocks.size(); ++bi)
      if (M.blocks[bi].kind == 1) owned[block_minpos[bi] / PIV].push_back((int)bi);
    g_doubles = 0;
    for (int k = 0; k < n_stages; ++k) {
      StageDesc &S = stages[k];
      S.n_active = active_count[k];
      S.g_begin = (int)g_doubles;
      S.iq_begin = (int)iq_blocks.size();
      for (int bi : owned[k]) {
        Block &b = M.blocks[bi];
        b.goff = (int)g_doubles;
        IqBlock q;
        q.m = b.m; q.n = b.n; q.row0 = b.row0; q.gloc = (int)(g_doubles - (pair_mode ? stages[k & ~1].g_begin : S.g_begin));   // (relative to the record's first stage)
        q.slot_off = (int)iq_slots.size();
        q.pad0 = q.pad1 = q.pad2 = 0;
        for (int a = 0; a < b.n; ++a) {
          iq_slots.push_back((short)var_slot[M.block_cols[b.col_off + a]]);
        }
        iq_blocks.push_back(q);
        iq_kron.push_back(kron ? kron_meta(M, bi) : KMeta());
        g_doubles += (long long)b.m * b.n;
      }
      S.g_len = (int)(g_doubles - S.g_begin);
      S.iq_end = (int)iq_blocks.size();
      max_stage_g = std::max(max_stage_g, S.g_len);
    }
    std::vector<std::vector<EqEntry>> ent(n_stages);
    std::vector<std::vector<std::pair<int, int>>> ent_pp(n_stages);   // (multiplier position, variable position) of every entry
    std::vector<std::vector<EqRhs>> rhs(n_stages);
    for (Block &b : M.blocks) {
      if (b.kind != 0) continue;
      if (!b.gstatic) { b.goff = (int)g_doubles; g_doubles += (long long)b.m * b.n; }
      for (int r = 0; r < b.m; ++r) {
        const int pr = row_pos[b.row0 + r], sr = row_slot[b.row0 + r];
        rhs[pr / PIV].push_back({b.row0 + r, sr});
        for (int a = 0; a < b.n; ++a) {
          const int var = M.block_cols[b.col_off + a];
          EqEntry e;
          e.src = b.gstatic ? -(b.goff + r * b.n + a + 1) : b.goff + r * b.n + a;
          e.slot_r = (short)sr;
          e.slot_c = (short)var_slot[var];
          ent[std::min(pr, var_pos[var]) / PIV].push_back(e);
          ent_pp[std::min(pr, var_pos[var]) / PIV].push_back({pr, var_pos[var]});
        }
      }
    }
    // Order of a stage's entries in its record (and with it in the problem's stream): the values the evaluation kernels
    // write in every linearisation first and grouped by the pass that writes them -- one-term dynamics entries, three-term
    // dynamics entries, terrain rows --, then everything that is written once (structural zeros and constants of the dynamics
    // blocks, static rows).  Interleaved as the blocks list them, the entries of one pass are runs of two among constants: 8-byte
    // stores scattered over every cache line of the stream (25 of the 39 k cycles of that pass).  Which
    // cell an entry is added to does not depend on its place in the list.
    {
      std::vector<char> cls((size_t)std::max<long long>(g_doubles, 1), 3);
      for (const DynInst &I : M.dyn) {
        if (!I.in_kkt || I.goff < 0) continue;
        const Block &b = M.blocks[I.goff];          // (instances hold their block id until finalize_goff)
        if (b.kind != 0 || b.gstatic) continue;
        auto mark = [&](const short cmap[12], int kind) {
          for (int sl = 0; sl < 4; ++sl)
            for (int d = 0; d < 3; ++d) {
              const int c = cmap[3 * sl + d];
              if (c < 0) continue;
              for (int i = 0; i < 3; ++i)           // (rows 3..5 of a column are constants: build_linear_terms)
                if (kind == 1) cls[b.goff + c + i * b.n] = 1;
                else if (i != d) cls[b.goff + c + i * b.n] = 0;
            }
        };
        mark(I.c_lin, 0);
        mark(I.c_ang, 1);
        for (int e = 0; e < NEE; ++e) { mark(I.c_p[e], 2 + e); mark(I.c_f[e], 6 + e); }
      }
      for (const TerrInst &T : M.terr) {
        if (!T.in_kkt || T.goff < 0) continue;
        const Block &b = M.blocks[T.goff];
        if (b.kind != 0 || b.gstatic) continue;
        for (int a = 0; a < b.n; ++a) cls[b.goff + a] = 2;
      }
      for (int k = 0; k < n_stages; ++k) {
        std::vector<int> perm(ent[k].size());
        std::iota(perm.begin(), perm.end(), 0);
        auto key = [&](int i) { return ent[k][i].src >= 0 ? (int)cls[ent[k][i].src] : 4; };
        std::stable_sort(perm.begin(), perm.end(), [&](int a, int b2) { return key(a) < key(b2); });
        std::vector<EqEntry> e2(ent[k].size());
        std::vector<std::pair<int, int>> p2(ent[k].size());
        for (size_t i = 0; i < perm.size(); ++i) { e2[i] = ent[k][perm[i]]; p2[i] = ent_pp[k][perm[i]]; }
        ent[k].swap(e2);
        ent_pp[k].swap(p2);
      }
    }
    // static entries between two variable unknowns (the proximal term of the reduced base: delta_x Z'Z off the diagonal).
    // They share cells with the entries the inequality blocks assemble (J' S J of the same coefficient pairs), so they are
    // NOT equality-type entries (added by other threads of the same assembly pass: a race on the cell): each becomes one more
    // contribution of its target in the gather table of the stage that eliminates the earlier of the two -- the target's
    // thread sums all of them in a fixed order.  {slot a, slot b, index into g_static, position a, position b}
    std::vector<std::vector<std::array<int, 6>>> sym_of(n_stages);   // (.. , tile type of the pair: classes 0+0, 0+1, 1+1)
    for (const HostModel::SymEntry &se : M.sym_static) {
      const int pa = var_pos[se.a], pb = var_pos[se.b];
      if (pa < 0 || pb < 0) continue;
      sym_of[std::min(pa, pb) / PIV].push_back({var_slot[se.a], var_slot[se.b], (int)M.g_static.size(), pa, pb, 0});
      M.g_static.push_back(se.val);
    }
    sym_pp.clear();
    for (int k = 0; k < n_stages; ++k) {
      stages[k].ent_begin = (int)eq_entries.size();
      eq_entries.insert(eq_entries.end(), ent[k].begin(), ent[k].end());
      stages[k].ent_end = (int)eq_entries.size();
      stages[k].rhs_begin = (int)eq_rhs.size();
      eq_rhs.insert(eq_rhs.end(), rhs[k].begin(), rhs[k].end());
      stages[k].rhs_end = (int)eq_rhs.size();
    }
    M.g_doubles = g_doubles;
    // ---- packed records ----
    auto trs = [](int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; };
    srec_off.assign(n_records + 1, 0);
    drec_off.assign(n_records + 1, 0);
    diag_pos.assign((size_t)n_stages * PIV, -1);
    std::vector<std::vector<int>> ent_spos(n_stages);   // stream position of every equality entry of a stage
    std::vector<std::vector<int>> pending;   // per record: inequality blocks left for continuation records
    std::vector<StageDesc> rec_desc(n_records);
    if (pair_mode) { REC_MAX_INTS = std::max(REC_MAX_INTS, 12288); REC_MAX_DOUBLES = 4094; if (rec_cap_ints > 0) { REC_MAX_INTS = std::min(REC_MAX_INTS, rec_cap_ints); REC_MAX_DOUBLES = std::min(REC_MAX_DOUBLES, rec_cap_ints / 3); } }
    for (int rec = 0; rec < n_records; ++rec) {
      const int k0 = rec * rec_stages, k1 = k0 + rec_stages;   // the record's stages
      // (the entries, right-hand sides and inequality blocks of consecutive stages are consecutive in their lists)
      StageDesc S = stages[k0];
      S.ent_end = stages[k1 - 1].ent_end; S.rhs_end = stages[k1 - 1].rhs_end; S.iq_end = stages[k1 - 1].iq_end;
      rec_desc[rec] = S;
      // 16-byte aligned records: k_kkt moves them with 128-bit loads
      while (srec.size() & 3) srec.push_back(0);
      while (pack_src.size() & 1) pack_src.push_back(6 << 28);
      srec_off[rec] = (int)srec.size();
      drec_off[rec] = (int)pack_src.size();
      const int n_ent = S.ent_end - S.ent_begin, n_rhs = S.rhs_end - S.rhs_begin, n_iq = S.iq_end - S.iq_begin;
      int hi_rec = 0;
      for (int k = k0; k < k1; ++k) hi_rec = std::max(hi_rec, stage_hi[k]);
      srec.push_back(n_ent); srec.push_back(n_rhs); srec.push_back(n_iq); srec.push_back(hi_rec);
      srec.push_back(0); srec.push_back(0); srec.push_back(0); srec.push_back(0);   // [4] gather table offset, [5] n_tgt
      for (int k = k0; k < k1; ++k)
        for (int i = 0; i < PIV; ++i) {
          srec.push_back(piv_slot[(size_t)k * PIV + i]);
          diag_pos[(size_t)k * PIV + i] = (int)pack_src.size();
          pack_src.push_back((5 << 28) | (k * PIV + i));
        }
      for (int k = k0; k < k1; ++k)
        for (int i = stages[k].ent_begin; i < stages[k].ent_end; ++i) {
          const EqEntry &e = eq_entries[i];
          srec.push_back(trs(e.slot_r, e.slot_c));
          ent_spos[k].push_back((int)pack_src.size());
          pack_src.push_back(e.src >= 0 ? e.src : ((1 << 28) | (-e.src - 1)));
        }
      for (int i = S.rhs_begin; i < S.rhs_end; ++i) {
        srec.push_back(eq_rhs[i].slot);
        pack_src.push_back((2 << 28) | eq_rhs[i].row);
      }
      // inequality blocks of the record: as many as the record limits allow go into the record itself,
      // the rest into continuation records (same layout, no equality part) that the kernel
      // fetches and assembles one after the other; they are emitted behind all stage records
      std::vector<std::array<int, 6>> syms;
      for (int k = k0; k < k1; ++k) syms.insert(syms.end(), sym_of[k].begin(), sym_of[k].end());
      std::vector<int> mine, rest;
      {
        std::vector<int> all(n_iq);
        std::iota(all.begin(), all.end(), 0);
        if (split_blocks(S, all, (int)pack_src.size() - drec_off[rec], (int)srec.size() - srec_off[rec], trs, mine, rest, (int)syms.size())) return -1;
      }
      if (emit_blocks(k0, S, mine, srec_off[rec], drec_off[rec], trs, &syms)) return -1;
      srec[srec_off[rec] + 2] = (int)mine.size() | (kron_off_of_record.count(srec_off[rec]) ? kron_off_of_record[srec_off[rec]] : 0);
      pending.push_back(rest);
      max_srec = std::max(max_srec, (int)srec.size() - srec_off[rec]);
      max_drec = std::max(max_drec, (int)pack_src.size() - drec_off[rec]);
    }
    while (srec.size() & 3) srec.push_back(0);
    while (pack_src.size() & 1) pack_src.push_back(6 << 28);
    srec_off[n_records] = (int)srec.size();
    drec_off[n_records] = (int)pack_src.size();
    // continuation records
    cont.clear();
    for (int rec = 0; rec < n_records; ++rec) {
      const StageDesc &S = rec_desc[rec];
      const int k = rec * rec_stages;
      std::vector<int> rest = pending[rec];
      int first_cont = (int)cont.size() / 4, n_cont = 0;
      while (!rest.empty()) {
        while (srec.size() & 3) srec.push_back(0);
        while (pack_src.size() & 1) pack_src.push_back(6 << 28);
        const int s0 = (int)srec.size(), d0 = (int)pack_src.size();
        srec.push_back(0); srec.push_back(0); srec.push_back(0); srec.push_back(stage_hi[k]);
        srec.push_back(0); srec.push_back(0); srec.push_back(0); srec.push_back(0);
        for (int i = 0; i < PIV * rec_stages; ++i) { srec.push_back(0); pack_src.push_back(6 << 28); }
        std::vector<int> mine, later;
        if (split_blocks(S, rest, PIV * rec_stages, SHDR_INTS + PIV * rec_stages, trs, mine, later)) return -1;
        if (mine.empty()) { err = "inequality block does not fit a record"; return -1; }
        if (emit_blocks(k, S, mine, s0, d0, trs)) return -1;
        srec[s0 + 2] = (int)mine.size();
        while (srec.size() & 3) srec.push_back(0);
        while (pack_src.size() & 1) pack_src.push_back(6 << 28);
        cont.push_back(s0); cont.push_back((int)srec.size() - s0); cont.push_back(d0); cont.push_back((int)pack_src.size() - d0);
        max_srec = std::max(max_srec, (int)srec.size() - s0);
        max_drec = std::max(max_drec, (int)pack_src.size() - d0);
        rest = later;
        ++n_cont;
      }
      srec[srec_off[rec] + 6] = n_cont;
      srec[srec_off[rec] + 7] = first_cont;
    }
    if (env.debug) fprintf(stderr, "qtos: %d stages, %d continuation records, max record %d ints / %d doubles\n", n_stages, (int)cont.size() / 4, max_srec, max_drec);
    if (env.debug >= 2) {
      // lifetimes
      std::map<int,int> hist;
      for (int j = 0; j < n_unknowns; ++j) hist[j / PIV - first[j] / PIV]++;
      fprintf(stderr, "lifetime(stages) histogram:");
      for (auto &kv : hist) fprintf(stderr, " %d:%d", kv.first, kv.second);
      fprintf(stderr, "\n");
      // live original entries in A per stage: target tri index -> (assembly stage, retire stage)
      // an entry (r,c) is assembled at stage s_a (record of that stage) and retired when min(pivot stage of slot r, of slot c)
      // here: count distinct targets per record and total
      long long tot_t = 0; int max_t = 0;
      for (int k = 0; k < n_records; ++k) { int nt = srec[srec_off[k] + 5] + srec[srec_off[k]] ; tot_t += nt; max_t = std::max(max_t, nt); }
      fprintf(stderr, "targets total %lld max/stage %d ; srec ints %zu, stream doubles %zu\n", tot_t, max_t, srec.size(), pack_src.size());
      fprintf(stderr, "active per stage:");
      for (int k = 0; k < n_stages; ++k) fprintf(stderr, " %d", active_count[k]);
      fprintf(stderr, "\nhi per stage:");
      for (int k = 0; k < n_stages; ++k) fprintf(stderr, " %d", stage_hi[k]);
      {
        double tiles = 0, pgs = 0, xt = 0;
        for (int k = 0; k < n_stages; ++k) {
          int g = 0, pg = 0;
          unsigned pgm = 0, gm = 0;
          for (int grp = 0; grp < 16; ++grp) {
            const unsigned w = (amask[(size_t)k * 8 + (grp >> 1)] >> ((grp & 1) * 16)) & 0xffffu;
            if (w) { g++; gm |= 1u << grp; }
          }
          for (int i = 0; i < PIV; ++i) pgm |= 1u << (piv_slot[(size_t)k * PIV + i] >> 4);
          pg = __builtin_popcount(pgm);
          tiles += g * (g + 1) / 2.0;
          pgs += pg;
          // tiles touched by the extraction of this stage's columns (two stages earlier: use this stage's occupancy)
          int t = 0;
          for (int R = 0; R < 16; ++R) for (int Cc = 0; Cc <= R; ++Cc)
            if (((gm | pgm) >> R & 1) && ((gm | pgm) >> Cc & 1) && ((pgm >> R & 1) || (pgm >> Cc & 1))) t++;
          xt += t;
        }
        fprintf(stderr, "\nmean update tiles %.2f, pivot groups %.2f, extraction tiles %.2f", tiles / n_stages, pgs / n_stages, xt / n_stages);
        if (front == 128) {   // tiles per SIMD if the tiles of empty groups were skipped (k_kkt2's static tile -> wave map: 12 update waves)
          double mx = 0;
          for (int k = 0; k < n_stages; ++k) {
            unsigned gm = 0;
            for (int grp = 0; grp < 8; ++grp)
              if ((amask[(size_t)k * 8 + (grp >> 1)] >> ((grp & 1) * 16)) & 0xffffu) gm |= 1u << grp;
            int per[4] = {0, 0, 0, 0};
            for (int uw = 0; uw < 12; ++uw)
              for (int i = 0; i < 3; ++i) {
                const int t = uw + 12 * i;
                int R = 0;
                while ((R + 1) * (R + 2) / 2 <= t) ++R;
                const int Cc = t - R * (R + 1) / 2;
                if (((gm >> R) & 1) && ((gm >> Cc) & 1)) per[(uw + 1 + uw / 3) & 3]++;
              }
            mx += std::max(per[1], std::max(per[2], per[3]));
          }
          fprintf(stderr, "\nmean over stages of the busiest SIMD's tiles with empty groups skipped: %.2f of 12", mx / n_stages);
        }
      }
      {
        std::map<std::pair<int,int>, int> bh;
        long long con = 0, conw = 0;
        for (const Block &b : M.blocks) if (b.kind == 1) { bh[{b.m, b.n}]++; con += (long long)b.n * (b.n + 1) / 2 + b.n; conw += ((long long)b.n * (b.n + 1) / 2 + b.n) * b.m; }
        fprintf(stderr, "\ninequality blocks (rows x cols: count):");
        for (auto &kv : bh) fprintf(stderr, " %dx%d:%d", kv.first.first, kv.first.second, kv.second);
        fprintf(stderr, "\ncontributions %lld, row-weighted %lld", con, conw);
      }
      {   // inequality rows and touched 16-slot groups per stage record (MFMA condensation study)
        fprintf(stderr, "\niq rows per stage (rows/chunks4/groups/tiles):");
        long long tot_mf = 0, tot_ent = 0; int mxr = 0;
        for (int k = 0; k < n_stages; ++k) {
          const StageDesc &S = stages[k];
          int rows = 0, ch = 0; long long mf = 0; unsigned gall = 0;
          for (int q = S.iq_begin; q < S.iq_end; ++q) {
            const IqBlock &Q = iq_blocks[q];
            unsigned gm = 0;
            for (int a = 0; a < Q.n; ++a) gm |= 1u << (iq_slots[Q.slot_off + a] >> 4);
            const int g = __builtin_popcount(gm), c4 = (Q.m + 3) / 4;
            rows += Q.m; ch += c4; mf += (long long)c4 * g * (g + 1) / 2; gall |= gm; tot_ent += Q.m * Q.n;
          }
          mxr = std::max(mxr, rows);
          tot_mf += mf;
          fprintf(stderr, " %d/%d/%d/%lld", rows, ch, __builtin_popcount(gall), mf);
        }
        fprintf(stderr, "\niq MFMAs total %lld (per stage %.1f), G entries %lld, max rows/stage %d", tot_mf, (double)tot_mf / n_stages, tot_ent, mxr);
      }
      fprintf(stderr, "\nenter per stage:");
      for (int k = 0; k < n_stages; ++k) fprintf(stderr, " %d", (int)enter[k].size());
      fprintf(stderr, "\n");
    }
    if (cont.empty()) cont.assign(4, 0);
    while (srec.size() & 3) srec.push_back(0);
    while (pack_src.size() & 1) pack_src.push_back(6 << 28);
    max_srec = (max_srec + 7) & ~3;   // a stage's record may end with alignment padding
    max_drec = (max_drec + 3) & ~1;
    if (compact_cells(first, slot_of)) return -1;
    // ---- direct-write maps derived from the stream layout ----
    eq_pos.assign((size_t)std::max<long long>(g_doubles, 1), -1);
    rhs_pos.assign(m, -1);
    sig_pos.assign(m, -1);
    w_pos.assign(m, -1);
    std::vector<int> iq_first((size_t)std::max<long long>(g_doubles, 1), -1);  // virtual G offset -> stream pos
    for (int i = 0; i < (int)pack_src.size(); ++i) {
      const int kind = pack_src[i] >> 28, idx = pack_src[i] & 0x0fffffff;
      if (kind == 0) { eq_pos[idx] = i; iq_first[idx] = i; }
      else if (kind == 1) { const_pos.push_back(i); const_val.push_back(M.g_static[idx]); }
      else if (kind == 2) rhs_pos[idx] = i;
      else if (kind == 3) sig_pos[idx] = i;
      else if (kind == 4) w_pos[idx] = i;
      else if (kind == 5) { const_pos.push_back(i); const_val.push_back(piv_diag[idx]); }
    }
    // inequality blocks: contiguous in the stream -> their goff becomes the stream offset itself
    for (Block &b : M.blocks)
      if (b.kind == 1) b.goff = iq_first[b.goff];
    M.finalize_goff();
    build_linear_terms(M);
    {   // right-hand side lists for the chord step: multipliers -g[row]; variables -sum_r G[r][a] w_r over their blocks
      std::vector<std::vector<std::pair<int, int>>> terms(n_unknowns);
      for (const Block &b : M.blocks) {
        if (b.kind != 1) continue;
        for (int a = 0; a < b.n; ++a) {
          const int p = var_pos[M.block_cols[b.col_off + a]];
          if (p < 0) continue;
          for (int r = 0; r < b.m; ++r) terms[p].push_back({b.goff + r * b.n + a, b.row0 + r});
        }
      }
      rhs_ptr.assign(1, 0);
      rhs_gpos.clear(); rhs_row.clear();
      for (int p = 0; p < n_unknowns; ++p) {
        if (order[p] >= n) { rhs_gpos.push_back(-1); rhs_row.push_back(order[p] - n); }
        else
          for (auto &t : terms[p]) { rhs_gpos.push_back(t.first); rhs_row.push_back(t.second); }
        rhs_ptr.push_back((int)rhs_gpos.size());
      }
    }
    {   // equality part of K by unknown (the entries of stage k sit behind the 16 pivot diagonals of its dynamic record)
      std::vector<std::vector<std::pair<int, int>>> rows(n_unknowns);
      for (int k = 0; k < n_stages; ++k)
        for (size_t i = 0; i < ent_pp[k].size(); ++i) {
          const int pr = ent_pp[k][i].first, pv = ent_pp[k][i].second, spos = ent_spos[k][i];
          rows[pr].push_back({pv, spos});
          rows[pv].push_back({pr, spos});
        }
      for (const auto &e : sym_pp) {
        rows[e[0]].push_back({e[1], e[2]});
        rows[e[1]].push_back({e[0], e[2]});
      }
      kx_ptr.assign(1, 0);
      kx_col.clear(); kx_pos.clear();
      for (int p = 0; p < n_unknowns; ++p) {
        for (auto &e : rows[p]) { kx_col.push_back(e.first); kx_pos.push_back(e.second); }
        kx_ptr.push_back((int)kx_col.size());
      }
    }
    // bytes = w * [ sum_k (p + c_k) * p  +  2 M ]   (matrix once, rhs in, solution out)
    algorithmic_bytes = 0;
    flops = 0;
    for (int k = 0; k < n_stages; ++k) {
      stages[k].n_active -= stage_dummies[k];   // (the unknowns of the system: dummies of short stages are not work of the algorithm)
      long long a = stages[k].n_active;
      algorithmic_bytes += 8LL * a * PIV;
      flops += 2LL * PIV * a * a;
    }
    algorithmic_bytes += 8LL * 2 * n_real_unknowns;
    return 0;
  }

  // ---- two-ended elimination: ANALYSIS ONLY (round 6, DESIGN.md section 5 "Two chains per compute unit") --------------------
  // The KKT matrix is block-banded in time: a chain L that eliminates from t = 0 forward and a chain R that eliminates from t = T
  // backward meet nothing of each other until they reach the unknowns that are alive across a split time -- the SEPARATOR, which
  // receives the Schur complements of both and is eliminated last.  This function answers what such an order would look like
  // for this model (it emits no kernel tables): for every stage boundary s of the current order
  //   S  = positions >= s whose envelope reaches below s (alive across the split),
  //   L  = positions < s in the current order (its fronts are the current ones, cut at s),
  //   R  = the other positions >= s, re-ordered for the mirrored problem -- by time stamp descending, a long-lived unknown
  //        (one whose envelope starts more than `long_life` seconds before its own time stamp: footholds, spline coefficients)
  //        at the START of its life instead of its end --, fronts by the mirrored envelope rule;
  // the split with the fewest serial steps max(stages L, stages R) + stages S is reported.
  struct TwoEnded {
    int split_stage = 0, stages_now = 0, front_now = 0;
    int stages_left = 0, stages_right = 0, stages_sep = 0, sep_unknowns = 0;
    int front_left = 0, front_right = 0, front_sep = 0;     // slots (multiples of 16)
    int serial_steps = 0;
    int peak_left = 0, peak_right = 0;                      // populated slots at the largest front of either chain
  };
  TwoEnded analyze_two_ended(const HostModel &M, double long_life = 0.3) const {
    TwoEnded best;
    const int N = n_unknowns, n = M.n_sol;
    best.stages_now = n_stages; best.front_now = front;
    // adjacency by position (pattern of K: an inequality block is a clique of its columns, an equality row couples its
    // multiplier with its columns, static entries couple two variables)
    std::vector<std::vector<int>> adj(N);
    auto link = [&](int a, int b) { if (a >= 0 && b >= 0 && a != b) { adj[a].push_back(b); adj[b].push_back(a); } };
    for (const Block &b : M.blocks) {
      if (b.kind == 1) {
        for (int a = 0; a < b.n; ++a)
          for (int c = 0; c < a; ++c) link(var_pos[M.block_cols[b.col_off + a]], var_pos[M.block_cols[b.col_off + c]]);
      } else {
        for (int r = 0; r < b.m; ++r)
          for (int a = 0; a < b.n; ++a) link(row_pos[b.row0 + r], var_pos[M.block_cols[b.col_off + a]]);
      }
    }
    for (const HostModel::SymEntry &e : M.sym_static) link(var_pos[e.a], var_pos[e.b]);
    std::vector<double> tp(N, 0.0);
    for (int p = 0; p < N; ++p) tp[p] = order[p] < 0 ? (p ? tp[p - 1] : 0.0) : (order[p] < n ? M.var_time[order[p]] : M.con_time[order[p] - n] + 1e-7);
    std::vector<int> lo(N), hi(N);
    for (int p = 0; p < N; ++p) {
      lo[p] = hi[p] = p;
      for (int q : adj[p]) { lo[p] = std::min(lo[p], q); hi[p] = std::max(hi[p], q); }
    }
    // (envelope: an unknown is in the front from the stage of the smallest position it is coupled with; fill stays inside)
    best.serial_steps = 1 << 30;
    for (int ks = n_stages / 4; ks <= (3 * n_stages) / 4; ++ks) {
      const int s = ks * PIV;
      TwoEnded t;
      t.split_stage = ks; t.stages_now = n_stages; t.front_now = front;
      std::vector<char> inS(N, 0);
      for (int j = s; j < N; ++j) if (lo[j] < s) { inS[j] = 1; t.sep_unknowns++; }
      // chain L: the current order cut at s
      t.stages_left = ks;
      for (int k = 0; k < ks; ++k) {
        int live = PIV;
        for (int j = (k + 1) * PIV; j < N; ++j) live += lo[j] < (k + 1) * PIV && (j < s || inS[j]);
        t.peak_left = std::max(t.peak_left, live);
      }
      // chain R: the mirrored order
      std::vector<int> R;
      for (int j = s; j < N; ++j) if (!inS[j] && order[j] >= 0) R.push_back(j);
      std::vector<double> key(N, 0.0);
      for (int j : R) {
        const double t_start = tp[lo[j]];
        key[j] = (tp[j] - t_start > long_life) ? t_start : tp[j];
      }
      std::stable_sort(R.begin(), R.end(), [&](int a, int b) { return key[a] > key[b] || (key[a] == key[b] && a > b); });
      std::vector<int> posr(N, 1 << 29);      // mirrored position; the separator's members are never eliminated by this chain
      for (size_t i = 0; i < R.size(); ++i) posr[R[i]] = (int)i;
      std::vector<int> firstr(N, 1 << 29);
      for (int j = s; j < N; ++j) {
        if (order[j] < 0) continue;
        firstr[j] = posr[j];
        for (int q : adj[j]) if (q >= s) firstr[j] = std::min(firstr[j], posr[q]);
      }
      t.stages_right = ((int)R.size() + PIV - 1) / PIV;
      for (int k = 0; k < t.stages_right; ++k) {
        int live = PIV;
        for (int j = s; j < N; ++j) live += order[j] >= 0 && posr[j] >= (k + 1) * PIV && firstr[j] < (k + 1) * PIV;
        t.peak_right = std::max(t.peak_right, live);
      }
      t.stages_sep = (t.sep_unknowns + PIV - 1) / PIV;
      t.front_left = ((t.peak_left + PIV - 1) / PIV) * PIV;
      t.front_right = ((t.peak_right + PIV - 1) / PIV) * PIV;
      t.front_sep = t.stages_sep * PIV;
      t.serial_steps = std::max(t.stages_left, t.stages_right) + t.stages_sep;
      const int fb = std::max(best.front_left, best.front_right), ft = std::max(t.front_left, t.front_right);
      if (t.serial_steps < best.serial_steps || (t.serial_steps == best.serial_steps && ft < fb)) best = t;
    }
    return best;
  }
};

}  // namespace qtos
