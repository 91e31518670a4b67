// kkt2.hpp -- the KKT kernels: k_kkt2 (factor + solve, 16 waves = 1024 threads per problem) and k_chord (solve with the
// factorisation of the preceding k_kkt2 launch and a new right-hand side).
//
// Chain of fronts, 16 pivots per stage: the assembled original entries in LDS cells (Symbolic::compact_cells), the
// Schur updates in f64 MFMA accumulator registers, the factor panels V_k = P_k B_k^-1 and w_k to HBM for the sweeps.
// Per stage two phases and two LDS-only barriers, no wave with two jobs in a row:
//
//   AB(k)  waves 0 .. NT-1   one 16-row tile of the panel each: V = P (L D L^T)^-1, next pivot columns, and the two
//                             operands of the Schur update into LDS: -V and P with the rows of the next pivots
//                             blanked (U -= V P^T = Y D^-1 Y^T without forming Y; fronts above 128 slots have no
//                             room for the second panel and form Y = P L^-T on service waves)
//          wave NT            the whole right-hand-side row: w = (L D L^T)^-1 p_F (to HBM), rhs -= P w, the right-hand
//                             side of the next pivots.  (y_F D^-1 Y[r]^T = P[r] w: the row needs nothing this phase makes.)
//   C(k)   wave 0             LDL^T + L^-1 of the next pivot block in registers, B^-1 = L^-T D^-1 L^-1 on the matrix core
//          update waves       (those not on the factor wave's SIMD: a SIMD has one vector ALU, and it stands still while
//                             an f64 matrix instruction runs) U -= V P^T on MAXT tiles each, extraction of stage k+2's
//                             columns, their share of the assembly of stage k+2's records
//          waves 4, 8, 12     assembly, then LDS-DMA of the records of stage k+3; wave 12: header of stage k+3 (pivot
//                             slots, slot map, masks, diagonals)
//   backward substitution and both sweeps of k_chord: one-stage look-ahead, one barrier per stage (sweep_backward)
//
// Fronts up to 208 slots: 13 row tiles, 91 Schur tiles = 7 per update wave (56 registers); the LDS budget holds
// because the assembled entries live in cells; records through prefetch registers instead of LDS-DMA there.
#pragma once
#include "kernels.hpp"

namespace qtos {

constexpr int KT2 = 1024;

template <int F>
struct Kkt2Cfg {
  static constexpr int NT = F / 16;
  static constexpr int NTILE = NT * (NT + 1) / 2;
  // Update waves: the f64 matrix instructions and the f64 vector instructions of a SIMD share one pipe, and
  // waves w, w+4, w+8, w+12 of a workgroup share a SIMD: the twelve waves that do not sit on the factor wave's
  // SIMD come first (update index u -> wave u + 1 + u / 3), waves 4 and 8 join only when a large front needs
  // them (u = 12, 13); wave 12 publishes headers.  As many update waves as divide the tiles evenly.
  static constexpr int NU = kkt_pick_nu(NTILE);
  static constexpr int MAXT = (NTILE + NU - 1) / NU;
  static constexpr int NSV = 16 - NT;            // service waves of the AB phase
  static constexpr int NH = NT <= 8 ? 2 : 1;     // backward pass: waves per row tile
  static constexpr int FR = (F + 63) & ~63;      // by-slot arrays padded to whole waves
};

// LDS layout (doubles)
template <int F>
struct Kkt2Layout {
  using CF = Kkt2Cfg<F>;
  static constexpr int PSZ = (F + 1) * PLD;
  static constexpr int LIB = 0;                          // 2 x 16 x PLD   L^-1 (current / next)
  static constexpr int DVB = LIB + 2 * PIV * PLD;        // 2 x 16         1 / d
  static constexpr int DGB = DVB + 2 * PIV;              // 3 x 16         pivot diagonals (ring)
  static constexpr int UF = DGB + 3 * PIV;               // FR             accumulated rhs updates
  static constexpr int XS = UF + CF::FR;                 // FR             solution by slot (backward)
  static constexpr int RED = XS + CF::FR;                // 2 x 16 x 16 partial sums + 64 dummy slots
  static constexpr int PSB = RED + 2 * 16 * PIV + 64;    // 3 x 16 ints    pivot slots (ring)
  static constexpr int HIB = PSB + 3 * PIV / 2;          // 4 ints
  static constexpr int JM = HIB + 2;                     // 2 x FR ints    slot -> pivot index
  static constexpr int PM = JM + CF::FR;                 // 2 x 8 ints     pivot-slot bit masks
  static constexpr int MIV = PM + 8;                     // 16 x PLD       (L D L^T)^-1 of the current pivot block
  static constexpr int PB = MIV + PIV * PLD;             // 3 panels of (F+1) x PLD: P_k / P_k+1 alternate in 0 and 2, 1 = operand A of the update
  static constexpr bool VP = F <= 128;                   // update as  U -= V P^T  (operand B in a fourth panel: no room for it on larger fronts)
  static constexpr int PMB = PB + 3 * PSZ;               // F x PLD: rows of P_k with those of the next pivots blanked
  static constexpr int VAR = (PMB + (VP ? F * PLD : 0) + 1) & ~1;   // dbuf, then (ints) sbuf, hiall, then the cells A
};
// record buffers: a front of up to 128 slots has two of them, filled by LDS-DMA (1 KB per wave instruction: sizes
// in 1 KB granules); larger fronts have one, filled through prefetch registers
__host__ __device__ inline size_t kkt2_dbuf_doubles(int F, int max_drec) {
  return F <= 128 ? (((size_t)max_drec * 8 + 1023) & ~(size_t)1023) / 8 : (((size_t)max_drec + 1) & ~(size_t)1);
}
__host__ __device__ inline size_t kkt2_sbuf_ints(int F, int max_srec) {
  return F <= 128 ? (((size_t)max_srec * 4 + 1023) & ~(size_t)1023) / 4 : (((size_t)max_srec + 3) & ~(size_t)3);
}
// what the backward sweep of k_kkt2 / k_kkt3 keeps in LDS in front of the helper waves' tables (sweep_ds_lds_bytes)
inline size_t kkt2_sweep_base_bytes(int F, int NS) {
  const int FR = (F + 63) & ~63;
  const size_t fixed = 2 * PIV * PLD + 2 * PIV + 3 * PIV + 2 * (size_t)FR + 2 * 16 * PIV + 64 + 3 * PIV / 2 + 2 + FR + 8 + PIV * PLD;
  return fixed * sizeof(double) + (size_t)NS * 12 * sizeof(int);
}
inline size_t kkt2_lds_bytes(int F, int NS, int max_srec, int max_drec, int n_cells) {
  const int FR = (F + 63) & ~63, PSZ = (F + 1) * PLD;
  const size_t fixed = 2 * PIV * PLD + 2 * PIV + 3 * PIV + 2 * (size_t)FR + 2 * 16 * PIV + 64 + 3 * PIV / 2 + 2 + FR + 8 + PIV * PLD;
  size_t o = (fixed + 3 * (size_t)PSZ + (F <= 128 ? (size_t)F * PLD : 0) + 1) & ~(size_t)1;
  const size_t nbuf = F <= 128 ? 2 : 1;
  o += nbuf * kkt2_dbuf_doubles(F, max_drec);
  size_t oi = 2 * o + nbuf * kkt2_sbuf_ints(F, max_srec) + ((((size_t)NS + 1) + 3) & ~(size_t)3);
  oi += 2 * (((size_t)n_cells + 1) & ~(size_t)1);
  // (the backward pass keeps NS x 12 ints of sweep tables where the panels were)
  const size_t sweep = fixed * sizeof(double) + (size_t)NS * 12 * sizeof(int);
  return oi * sizeof(int) > sweep ? oi * sizeof(int) : sweep;
}

// pivot-slot mask of up to 256 bits: lane l holds word l & 7
struct Mask256 {
  int v;
};
__device__ __forceinline__ Mask256 load_mask8(const unsigned *pm8, int lane) {
  Mask256 m;
  m.v = (int)pm8[lane & 7];
  return m;
}
__device__ __forceinline__ unsigned grp16(const Mask256 &m, int grp) {
  return ((unsigned)__builtin_amdgcn_readlane(m.v, grp >> 1) >> ((grp & 1) * 16)) & 0xffffu;
}
// acc += bcast_K(w) * b : the K-th lane of every 16-lane row of w, one DP-ALU DPP instruction
template <int K>
__device__ __forceinline__ void fma_bc(double &acc, double w, double b) {
  asm volatile("v_fmac_f64 %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(b), "n"(K));
}
template <int K>
__device__ __forceinline__ void dot16_steps(double &acc, double w, const double (&p)[PIV]) {
  if constexpr (K < PIV) {
    fma_bc<K>(acc, w, p[K]);
    dot16_steps<K + 1>(acc, w, p);
  }
}

// two consecutive ints by a scalar load (the compiler reads plan arrays reached through the by-value DevPlan
// with vector loads and a full vmcnt(0) drain: it cannot prove them read-only)
__device__ __forceinline__ void sload2(const int *p, int &a, int &b) {
  typedef int i2_t __attribute__((ext_vector_type(2)));
  i2_t r;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  a = r[0];
  b = r[1];
}

// ---- substitution sweeps with a one-stage look-ahead ------------------------------------------------------------
// A triangular solve with the chain of fronts is a chain of NS dependent steps: x_k needs x_{k+1} (backward), p_k needs
// p_{k-1} (forward).  Per stage only a 16 x 16 block carries that dependence -- the rows of V_k that belong to the
// pivots of stage k+1 (Symbolic::nxt_pack) --; every other row of V_k meets values that are one stage older.  Wave 0
// carries the chain (the block product, in registers), waves 1..NT apply the other rows one stage behind it
// (Symbolic::amask2) and hand their partial sums over through LDS: one barrier per stage, and between two barriers
// the chain costs one reduction of 16 partials and 16 row-broadcast multiply-adds.

// elementwise sum over the four 16-lane rows of the wave, the same bits on every lane: (r0 + r2) + (r1 + r3)
__device__ __forceinline__ double rowsum4(double p) {
  int lo = __double2loint(p), hi = __double2hiint(p);
  auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double s = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
  lo = __double2loint(s);
  hi = __double2hiint(s);
  auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(h2[0], l2[0]) + __hiloint2double(h2[1], l2[1]);
}
// acc += (lane K of the 16-lane row of x) * b on the rows of the wave selected by RM
template <int K, int RM>
__device__ __forceinline__ void fma_bc_rows(double &acc, double x, double b) {
  asm volatile("v_fmac_f64 %0, %1, %2 row_newbcast:%3 row_mask:%4 bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(b), "n"(K), "n"(RM));
}
// on the lanes of row q:  sum_m x[q + 4 m] c[m]   (x[i] lives on lane i of every row; four independent chains)
__device__ __forceinline__ double dot4_by_row(double x, const double (&c)[4]) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  asm volatile("s_nop 1" : "+v"(x), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));   // DPP read of a VALU result: two wait states
  fma_bc_rows<0, 1>(a0, x, c[0]); fma_bc_rows<1, 2>(a0, x, c[0]); fma_bc_rows<2, 4>(a0, x, c[0]); fma_bc_rows<3, 8>(a0, x, c[0]);
  fma_bc_rows<4, 1>(a1, x, c[1]); fma_bc_rows<5, 2>(a1, x, c[1]); fma_bc_rows<6, 4>(a1, x, c[1]); fma_bc_rows<7, 8>(a1, x, c[1]);
  fma_bc_rows<8, 1>(a2, x, c[2]); fma_bc_rows<9, 2>(a2, x, c[2]); fma_bc_rows<10, 4>(a2, x, c[2]); fma_bc_rows<11, 8>(a2, x, c[2]);
  fma_bc_rows<12, 1>(a3, x, c[3]); fma_bc_rows<13, 2>(a3, x, c[3]); fma_bc_rows<14, 4>(a3, x, c[3]); fma_bc_rows<15, 8>(a3, x, c[3]);
  return (a0 + a1) + (a2 + a3);
}
#ifndef QTOS_SWD
#define QTOS_SWD 4
#endif
constexpr int SWD = QTOS_SWD;   // stages of factor panel in flight per wave (prefetch ring of the sweeps)

// backward substitution  x_k = w_k - V_k^T x  over the whole chain.  xs (solution by slot) zeroed, red = 2 x 16 x 16
// doubles, nxp = LDS copy of Symbolic::nxt_pack (NS x 4 ints) followed by Symbolic::amask2 (NS x 8): they are addresses of
// the panel loads, a copy in global memory would put a second memory round trip in front of every one of them.  The
// caller has synchronised the workgroup.
// The slack steps ds = Ji dx + (g - s) on the waves that idle in the sweep (P.sw_on; Symbolic-independent tables built by
// the planner: sw_tasks, sw_cpos).  An inequality block belongs to the stage of its earliest column: once the chain has
// solved that stage every column of the block is known.  Waves 13..15 (three SIMDs that do not hold the chain wave) take
// the rows of the blocks in turns, one round of 16 rows per step of the chain, four lanes per row with k_step's own order
// of summation (the same bits), G and the column positions prefetched three steps ahead; xp = the solution by position in LDS.
struct SweepDs {
  const double *G;     // the problem's stream
  double *ds;
  const double *g, *s;
  double *dbg;         // diagnostic build: cycles of the helper waves' turns and of the chain wave's step parts (sweep_backward_early)
};
constexpr int SW_W0 = 13, SW_NW = 3, SW_ROUND = 16, SW_RU = 8;
typedef int swi4_t __attribute__((ext_vector_type(4)));
typedef int swi2_t __attribute__((ext_vector_type(2)));
#ifdef QTOS_STAMPS
#define HSTAMP(v) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); v = t_; } while (0)
#else
#define HSTAMP(v) do {} while (0)
#endif
struct SwSlot {
  double gv[SW_RU], gr[3], g_row, s_row;
  swi4_t c;    // positions of the lane's entries, 16 bits each (sw_c16)
  swi2_t cr;
};
// LDS behind the sweep tables: the solution by position (NS x 16 doubles), then the rounds' rows (first four ints of a SwTask)
__host__ __device__ inline size_t sweep_ds_lds_bytes(int NS, int sw_steps) {
  return sw_steps > 0 ? 16 + (size_t)NS * PIV * sizeof(double) + (size_t)sw_steps * SW_ROUND * 16 : 0;
}
// T = the first four ints of a SwTask: goff, n, row, c16_off
__device__ __forceinline__ void sw_load(const DevPlan &P, const SweepDs &sd, const swi4_t &T, SwSlot &S, int q) {
  const double *Gr = sd.G + T[0];
  const int n = T[1], n4 = n & ~3, row = max(T[2], 0);
  S.g_row = sd.g[row];
  S.s_row = sd.s[row];
  S.c = *(const swi4_t *)(P.sw_c16 + T[3] + 4 * q);
  S.cr = *(const swi2_t *)(P.sw_c16 + T[3] + 16);
#pragma unroll
  for (int u = 0; u < SW_RU; ++u) S.gv[u] = Gr[max(min(q + 4 * u, n4 - 4 + q), 0)];
#pragma unroll
  for (int u = 0; u < 3; ++u) S.gr[u] = Gr[min(n4 + u, n - 1)];
}
// sw_load with the row's entries unclamped (sweep_backward_early)
__device__ __forceinline__ void sw_load2(const DevPlan &P, const SweepDs &sd, const swi4_t &T, SwSlot &S, int q) {
  const double *Gr = sd.G + T[0];
  const int n = T[1], n4 = n & ~3, row = max(T[2], 0);
  S.g_row = sd.g[row];
  S.s_row = sd.s[row];
  S.c = *(const swi4_t *)(P.sw_c16 + T[3] + 4 * q);
  S.cr = *(const swi2_t *)(P.sw_c16 + T[3] + 16);
  // (entries q + 4 u and n4 + u unclamped: one address for the eight, one for the three; what lies behind the row's end --
  //  the stream goes on, the allocation has 64 doubles to spare -- is loaded and never used: sw_row selects by the counts)
  const double *Gq = Gr + q, *Gt = Gr + n4;
#pragma unroll
  for (int u = 0; u < SW_RU; ++u) S.gv[u] = Gq[4 * u];
#pragma unroll
  for (int u = 0; u < 3; ++u) S.gr[u] = Gt[u];
  (void)n;
}
__device__ __forceinline__ void sw_row(const DevPlan &P, const SweepDs &sd, const swi4_t &T, const SwSlot &S, const double *xp, int q, const SwTask *full) {
  const int n = T[1], n4 = n & ~3;
  double ev[SW_RU], er[3];
#pragma unroll
  for (int u = 0; u < SW_RU; ++u) ev[u] = xp[((unsigned)S.c[u >> 1] >> (16 * (u & 1))) & 0xffffu];
#pragma unroll
  for (int u = 0; u < 3; ++u) er[u] = xp[((unsigned)S.cr[u >> 1] >> (16 * (u & 1))) & 0xffffu];
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < SW_RU; ++u) acc = q + 4 * u < n4 ? fma(S.gv[u], ev[u], acc) : acc;
  if (n4 > 4 * SW_RU) {   // (rows of more than 35 entries: the rest of the whole groups, at memory latency)
    const double *Gr = sd.G + T[0];
    const int *cols = P.sw_cpos + full->cpos_off;
    for (int a = q + 4 * SW_RU; a < n4; a += 4) acc = fma(Gr[a], xp[cols[a]], acc);
  }
  // (no branch around the uses of gr: behind a branch the compiler must take the loads for pending at the next turn's
  //  address arithmetic -- their registers are recycled -- and waits for everything, the row's store included)
#pragma unroll
  for (int u = 0; u < 3; ++u) acc = (q == 0 && n4 + u < n) ? fma(S.gr[u], er[u], acc) : acc;
  acc = quadsum(acc);
  if (q == 0 && T[2] >= 0) sd.ds[T[2]] = acc + (S.g_row - S.s_row);
}

// The rounds the schedule could not place inside the sweep (a transcription with more rows than 16 per stage): the solution
// is complete, every wave of the workgroup takes rounds (the helper waves alone go through them one memory latency at a time).
// k_chord only (sweep_backward_early): any change to the text of sweep_backward has moved k_kkt2 / k_kkt3 by +1 % per launch
// in the A/B of libraries (profiles/r04_experiments/sweep_variants.log), this one included.
template <bool UNCLAMPED>
__device__ __forceinline__ void sw_tail(const DevPlan &P, const SweepDs &sd, const swi4_t *swt, const double *xp, int nstep, int wv, int lane) {
  const int hq = lane & 3, hr = lane >> 2;
  for (int r = nstep + wv; r < P.sw_steps; r += 16) {
    const swi4_t T = swt[r * SW_ROUND + hr];
    SwSlot S;
    if (UNCLAMPED) sw_load2(P, sd, T, S, hq); else sw_load(P, sd, T, S, hq);
    sw_row(P, sd, T, S, xp, hq, P.sw_tasks + (size_t)r * SW_ROUND + hr);
  }
}

// NTHR / HW0: threads of the workgroup and the first of the three helper waves (k_kkt5 runs twelve waves: helpers 9 .. 11)
template <int F, int NTHR = KT2, int HW0 = SW_W0>
__device__ __forceinline__ void sweep_backward(const DevPlan &P, const double *__restrict__ panel, double *__restrict__ dx,
                                               double *__restrict__ sol, double *xs, double *red, const int *nxp, int wv, int lane,
                                               double *xp, const SweepDs &sd) {
  constexpr int NT = Kkt2Cfg<F>::NT, pstride = (F + 1) * PIV;
  constexpr bool HELP = NT < HW0;   // (fronts of 208 slots and more have no idle helper waves: the planner leaves sw_on off)
  const int NS = P.n_stages, n = P.n_sol;
  const int j = lane & 15, q = lane >> 4;
  const int vcol = 4 * (j & 3) + (j >> 2);   // where column j of a V row sits
  const bool owner = wv >= 1 && wv <= NT;
  const int R = owner ? wv - 1 : 0;
  // ring slot of a stage: four doubles -- the chain block on wave 0, the rows of the wave's tile on waves 1..NT --, and on
  // wave 0 the stage's w, pivot slots and unknowns
  double bv[SWD][4], bw[SWD];
  int bps[SWD], bun[SWD];
  unsigned bam[SWD];
  auto load = [&](int s, double (&v)[4], unsigned &am, double &wj, int &psj, int &unkj) __attribute__((always_inline)) {
    const int kk = max(s, 0);
    const double *pk = panel + (size_t)kk * pstride;
    // (the masks and slots below are addresses of the loads: from LDS)
    am = owner ? ((unsigned)nxp[NS * 4 + kk * 8 + (R >> 1)] >> ((R & 1) * 16)) & 0xffffu : 0u;
    const unsigned np = (unsigned)nxp[kk * 4 + q];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = q + 4 * i;
      const unsigned nr = (np >> (8 * i)) & 255u;
      const int off = wv == 0 ? (nr != 255u ? PIV + (int)nr * PIV + vcol : j) : (((am >> row) & 1u) ? PIV + (16 * R + row) * PIV + vcol : j);
      v[i] = pk[off];
    }
    if (wv == 0) {
      wj = pk[j];
      psj = P.piv_slot[kk * PIV + j];
      unkj = P.piv_unknown[kk * PIV + j];
    }
  };
  if (lane < PIV) { red[wv * PIV + lane] = 0.0; red[256 + wv * PIV + lane] = 0.0; }   // (the last stage has no rows besides its pivots)
  swi4_t *swt = (swi4_t *)(xp + NS * PIV);   // (16-byte aligned: the callers' xp is)
  if (HELP && P.sw_on)
    for (int i = wv * 64 + lane; i < P.sw_steps * SW_ROUND; i += NTHR) swt[i] = *(const swi4_t *)(P.sw_tasks + i);
  // waves NT+1 .. 15 have no rows: they only meet the barriers (a SIMD has one vector ALU: what they would execute on
  // dummies is time the working waves of their SIMD do not get)
  const bool active = wv <= NT;
  if (!active) {
    if (!HELP || wv < HW0 || !P.sw_on) {
      lds_barrier();
      for (int k0 = NS - 1; k0 >= 0; k0 -= SWD)
#pragma unroll
        for (int d = 0; d < SWD; ++d) lds_barrier();
      return;
    }
    // round i (16 rows) runs in step i of the chain (stage NS - 1 - i); the planner's schedule puts a block's rows behind
    // its stage.  The three waves take turns: a wave's loads have three steps to arrive, one set of prefetch registers.
    const int hq = lane & 3, hr = lane >> 2, hw = wv - HW0;
    const int nstep = ((NS + SWD - 1) / SWD) * SWD;   // (steps of the chain loop below)
    lds_barrier();
    // (the rows of a round come from the LDS copy of the schedule: a descriptor carried from turn to turn in registers is
    //  copied into the carried register right behind its load -- a wait for the memory in front of the step's barrier)
    auto task = [&](int i) __attribute__((always_inline)) { return swt[min(i, P.sw_steps - 1) * SW_ROUND + hr]; };
    swi4_t T = task(hw);
    SwSlot S;
    sw_load(P, sd, T, S, hq);
    int turn = hw;
    for (int i = 0; i < nstep; ++i) {
      if (i == turn) {
        sw_row(P, sd, T, S, xp, hq, P.sw_tasks + (size_t)turn * SW_ROUND + hr);
        turn += SW_NW;
        T = task(turn);
        sw_load(P, sd, T, S, hq);
      }
      lds_barrier();
    }
    for (; turn < P.sw_steps; turn += SW_NW) {   // rounds the schedule could not place earlier (x is complete)
      sw_row(P, sd, T, S, xp, hq, P.sw_tasks + (size_t)turn * SW_ROUND + hr);
      T = task(turn + SW_NW);
      sw_load(P, sd, T, S, hq);
    }
    return;
  }
#pragma unroll
  for (int d = 0; d < SWD; ++d) load(NS - 1 - d, bv[d], bam[d], bw[d], bps[d], bun[d]);
  double corr = 0.0;
  lds_barrier();
  for (int k0 = NS - 1; k0 >= 0; k0 -= SWD) {
#pragma unroll
    for (int d = 0; d < SWD; ++d) {
      const int t = k0 - d, dn = (d + 1) % SWD;   // stage of the chain; ring slot of stage t - 1
      if (wv == 0) {
        double r16[NT];                           // (the partial sums of waves 1 .. NT)
#pragma unroll
        for (int w2 = 0; w2 < NT; ++w2) r16[w2] = red[(t & 1) * 256 + (w2 + 1) * PIV + j];
        double sm = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NT; ++w2) sm += r16[w2];
        const double x = t >= 0 ? bw[d] - sm - corr : 0.0;
        if (lane < PIV && t >= 0) {
          xs[bps[d]] = x;
          xp[t * PIV + lane] = x;        // by position, for the helper waves (ds = Ji dx)
          sol[t * PIV + lane] = x;       // by unknown position, multipliers included (k_residual)
          if (bun[d] >= 0 && bun[d] < n) dx[bun[d]] = x;
        }
        // the block of stage t - 1 against the entries just found
        const unsigned np = (unsigned)nxp[max(t - 1, 0) * 4 + q];
        double c[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) c[m] = ((np >> (8 * m)) & 255u) != 255u && t >= 1 ? bv[dn][m] : 0.0;
        corr = rowsum4(dot4_by_row(x, c));
      }
      // the other rows of stage t - 1 (they meet entries that are at least one barrier old)
      if (wv >= 1) {
        double pp = 0.0;
        const unsigned am = bam[dn];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = q + 4 * i;
          const double xv = xs[16 * R + row];
          pp = ((am >> row) & 1u) ? fma(bv[dn][i], xv, pp) : pp;
        }
        pp = rowsum4(pp);
        if (lane < PIV) red[((t - 1) & 1) * 256 + wv * PIV + j] = t >= 1 ? pp : 0.0;
      }
      lds_barrier();
      load(t - SWD, bv[d], bam[d], bw[d], bps[d], bun[d]);
    }
  }
}


// sweep_backward with the prefetch of a ring slot IN FRONT of the step's barrier and its table word read with the step's
// other LDS reads (round 4).  k_chord's backward sweep: 1.74 k -> 1.51 k cycles per stage (0.156 -> 0.145 ms per launch); inside
// k_kkt2 / k_kkt3 the same text measured 5 - 10 us per launch SLOWER than sweep_backward above (A/B of two libraries on one
// box, profiles/r04_experiments/sweep_variants.log), so they keep that one.  Helper waves: every wave without rows but wave 12.
template <int F, bool EARLY>
__device__ __forceinline__ void sweep_backward_early(const DevPlan &P, const double *__restrict__ panel, double *__restrict__ dx,
                                               double *__restrict__ sol, double *xs, double *red, const int *nxp, int wv, int lane,
                                               double *xp, const SweepDs &sd) {
  constexpr int NT = Kkt2Cfg<F>::NT, pstride = (F + 1) * PIV;
  // helper waves (ds = Ji dx): the waves without rows, except wave 12 on the chain wave's SIMD; fronts of 208 slots and more
  // have fewer than three: the planner leaves sw_on off
  constexpr int HN = (15 - NT) - (NT < 12 ? 1 : 0);
  constexpr bool HELP = NT < SW_W0;
  const int NS = P.n_stages, n = P.n_sol;
  const int j = lane & 15, q = lane >> 4;
  const int vcol = 4 * (j & 3) + (j >> 2);   // where column j of a V row sits
  const bool owner = wv >= 1 && wv <= NT;
  const int R = owner ? wv - 1 : 0;
  // ring slot of a stage: four doubles -- the chain block on wave 0, the rows of the wave's tile on waves 1..NT --, and on
  // wave 0 the stage's w, pivot slots and unknowns
  double bv[SWD][4], bw[SWD];
  int bps[SWD], bun[SWD];
  unsigned bam[SWD];
  // The sweep tables are addresses of the panel loads.  A stage's word (wave 0: the slots of the next stage's pivots, nxt_pack;
  // row waves: the row mask of their tile, amask2) is read from LDS together with the step's other LDS reads and handed to
  // fetch(): read inside it, behind the step's barrier, every word was a round trip of its own in front of the chain's reads
  // of the partial sums (round 4: 1.74 k -> cycles per stage).  am: row waves the tile's row mask, wave 0 the rows of the
  // chain block that exist (bit m: pivot q + 4 m of the next stage has a slot at this stage).
  auto word_of = [&](int s) __attribute__((always_inline)) {
    const int kk = max(s, 0);
    return (unsigned)(wv == 0 ? nxp[kk * 4 + q] : nxp[NS * 4 + kk * 8 + (R >> 1)]);
  };
  auto fetch = [&](int s, unsigned word, double (&v)[4], unsigned &am, double &wj, int &psj, int &unkj) __attribute__((always_inline)) {
    const int kk = max(s, 0);
    const double *pk = panel + (size_t)kk * pstride;
    if (wv == 0) {
      unsigned present = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned nr = (word >> (8 * i)) & 255u;
        present |= (nr != 255u ? 1u : 0u) << i;
        v[i] = pk[nr != 255u ? PIV + (int)nr * PIV + vcol : j];
      }
      am = present;
      wj = pk[j];
      psj = P.piv_slot[kk * PIV + j];
      unkj = P.piv_unknown[kk * PIV + j];
    } else {
      am = (word >> ((R & 1) * 16)) & 0xffffu;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = q + 4 * i;
        v[i] = pk[((am >> row) & 1u) ? PIV + (16 * R + row) * PIV + vcol : j];
      }
    }
  };
  if (lane < PIV) { red[wv * PIV + lane] = 0.0; red[256 + wv * PIV + lane] = 0.0; }   // (the last stage has no rows besides its pivots)
  swi4_t *swt = (swi4_t *)(xp + NS * PIV);   // (16-byte aligned: the callers' xp is)
  if (HELP && P.sw_on)
    for (int i = wv * 64 + lane; i < P.sw_steps * SW_ROUND; i += KT2) swt[i] = *(const swi4_t *)(P.sw_tasks + i);
  // waves NT+1 .. 15 have no rows: they only meet the barriers (a SIMD has one vector ALU: what they would execute on
  // dummies is time the working waves of their SIMD do not get)
  const bool active = wv <= NT;
  if (!active) {
    if (!HELP || wv == 12 || !P.sw_on) {
      lds_barrier();
      for (int k0 = NS - 1; k0 >= 0; k0 -= SWD)
#pragma unroll
        for (int d = 0; d < SWD; ++d) lds_barrier();
      if (HELP && P.sw_on) sw_tail<true>(P, sd, swt, xp, ((NS + SWD - 1) / SWD) * SWD, wv, lane);
      return;
    }
    // round i (16 rows) runs in step i of the chain (stage NS - 1 - i); the planner's schedule puts a block's rows behind
    // its stage.  The helper waves take turns: a wave's loads have HN steps to arrive (a load from the stream takes longer
    // than two steps of the chain: with three waves the sweep waited for them), one set of prefetch registers.
    const int hq = lane & 3, hr = lane >> 2, hw = wv - (NT + 1) - (wv > 12 && NT < 12 ? 1 : 0);
    const int nstep = ((NS + SWD - 1) / SWD) * SWD;   // (steps of the chain loop below)
    lds_barrier();
    // (the rows of a round come from the LDS copy of the schedule: a descriptor carried from turn to turn in registers is
    //  copied into the carried register right behind its load -- a wait for the memory in front of the step's barrier)
    auto task = [&](int i) __attribute__((always_inline)) { return swt[min(i, P.sw_steps - 1) * SW_ROUND + hr]; };
    // (sums and loads of a turn in ONE step: split over two steps the compiler puts a wait for the row's store in front of
    //  the loads -- it cannot know that the sums' branch has waited for the registers they recycle; the next round's rows are
    //  read from LDS ahead of the sums, their latency under the sums' chain)
    swi4_t T = task(hw);
    SwSlot S;
    sw_load2(P, sd, T, S, hq);
    int turn = hw;
#ifdef QTOS_STAMPS
    unsigned long long hs0 = 0, hs1 = 0, hs2 = 0, hn = 0;
#endif
    for (int i = 0; i < nstep; ++i) {
      if (i == turn) {
#ifdef QTOS_STAMPS
        unsigned long long a0, a1, a2;
#endif
        HSTAMP(a0);
        const swi4_t Tn = task(turn + HN);
        sw_row(P, sd, T, S, xp, hq, P.sw_tasks + (size_t)turn * SW_ROUND + hr);
        HSTAMP(a1);
        turn += HN;
        T = Tn;
        sw_load2(P, sd, T, S, hq);
        HSTAMP(a2);
#ifdef QTOS_STAMPS
        hs0 += a1 - a0; hs1 += a2 - a1; hn += 1;
#endif
      }
#ifdef QTOS_STAMPS
      unsigned long long b0, b1;
      HSTAMP(b0);
#endif
      lds_barrier();
#ifdef QTOS_STAMPS
      HSTAMP(b1);
      if (i + 1 == turn || true) hs2 += b1 - b0;
#endif
    }
#ifdef QTOS_STAMPS
    if (sd.dbg && wv == 13 && lane == 0) { sd.dbg[0] = (double)hs0 / (double)hn; sd.dbg[1] = (double)hs1 / (double)hn; sd.dbg[2] = (double)hs2 / (double)nstep; sd.dbg[3] = (double)hn; }
#endif
    sw_tail<true>(P, sd, swt, xp, nstep, wv, lane);
    return;
  }
#pragma unroll
  for (int d = 0; d < SWD; ++d) fetch(NS - 1 - d, word_of(NS - 1 - d), bv[d], bam[d], bw[d], bps[d], bun[d]);
  double corr = 0.0;
#ifdef QTOS_STAMPS
  unsigned long long cs[4] = {0, 0, 0, 0}, wk = 0;
#endif
  lds_barrier();
  for (int k0 = NS - 1; k0 >= 0; k0 -= SWD) {
#pragma unroll
    for (int d = 0; d < SWD; ++d) {
      const int t = k0 - d, dn = (d + 1) % SWD;   // stage of the chain; ring slot of stage t - 1
#ifdef QTOS_STAMPS
      unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
      HSTAMP(c0);
#endif
      const unsigned wnext = EARLY ? word_of(t - SWD) : 0u;    // (for the prefetch at the end of the step)
      if (wv == 0) {
        double r16[NT];                           // (the partial sums of waves 1 .. NT)
#pragma unroll
        for (int w2 = 0; w2 < NT; ++w2) r16[w2] = red[(t & 1) * 256 + (w2 + 1) * PIV + j];
        double sm = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NT; ++w2) sm += r16[w2];
        const double x = t >= 0 ? bw[d] - sm - corr : 0.0;
#ifdef QTOS_STAMPS
        HSTAMP(c1);
#endif
        if (lane < PIV && t >= 0) {
          xs[bps[d]] = x;
          xp[t * PIV + lane] = x;        // by position, for the helper waves (ds = Ji dx)
#ifndef QTOS_EXP_CHAIN
          sol[t * PIV + lane] = x;       // by unknown position, multipliers included (k_residual)
          if (bun[d] >= 0 && bun[d] < n) dx[bun[d]] = x;
#endif
        }
        // the block of stage t - 1 against the entries just found
        double c[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) c[m] = ((bam[dn] >> m) & 1u) && t >= 1 ? bv[dn][m] : 0.0;
        corr = rowsum4(dot4_by_row(x, c));
#ifdef QTOS_STAMPS
        HSTAMP(c2);
#endif
      }
      // the other rows of stage t - 1 (they meet entries that are at least one barrier old)
      if (wv >= 1) {
        double pp = 0.0;
        const unsigned am = bam[dn];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = q + 4 * i;
          const double xv = xs[16 * R + row];
          pp = ((am >> row) & 1u) ? fma(bv[dn][i], xv, pp) : pp;
        }
        pp = rowsum4(pp);
        if (lane < PIV) red[((t - 1) & 1) * 256 + wv * PIV + j] = t >= 1 ? pp : 0.0;
      }
      // ring slot d is free: wave 0 has used the stage's w, slots and unknowns above, the row waves its rows a step ago
#ifdef QTOS_EXP_CHAIN
      if (wv != 0)
#endif
      if (EARLY) fetch(t - SWD, wnext, bv[d], bam[d], bw[d], bps[d], bun[d]);
#ifdef QTOS_STAMPS
      HSTAMP(c3);
      wk += c3 - c0;
#endif
      lds_barrier();
      if (!EARLY) fetch(t - SWD, word_of(t - SWD), bv[d], bam[d], bw[d], bps[d], bun[d]);
#ifdef QTOS_STAMPS
      if (wv == 0) { HSTAMP(c4); cs[0] += c1 - c0; cs[1] += c2 - c1; cs[2] += c3 - c2; cs[3] += c4 - c3; }
#endif
    }
  }
#ifdef QTOS_STAMPS
  if (sd.dbg && wv == 0 && lane == 0) for (int i = 0; i < 4; ++i) sd.dbg[4 + i] = (double)cs[i] / (double)NS;
  if (sd.dbg && lane == 0) sd.dbg[8 + wv] = (double)wk / (double)NS;
#endif
  if (HELP && P.sw_on) sw_tail<true>(P, sd, swt, xp, ((NS + SWD - 1) / SWD) * SWD, wv, lane);
}



template <int F, bool CONT, bool KRON = false>
__global__ __launch_bounds__(KT2) void k_kkt2(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x;
  if (b >= B || W.done[b] || W.chord[b] == 1) return;   // (a problem flagged for a chord step is k_chord's)
  extern __shared__ double lds[];
  using CF = Kkt2Cfg<F>;
  using LY = Kkt2Layout<F>;
  constexpr int NT = CF::NT, NU = CF::NU, MAXT2 = CF::MAXT, NSV = CF::NSV, NH = CF::NH, FR = CF::FR, PSZ = LY::PSZ;
  const int tid = threadIdx.x, NS = P.n_stages, n = P.n_sol;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
  double *Lib = lds + LY::LIB, *dvb = lds + LY::DVB, *dgb = lds + LY::DGB, *UF = lds + LY::UF, *xs = lds + LY::XS;
  double *red = lds + LY::RED, *PB = lds + LY::PB, *dbuf = lds + LY::VAR;
  int *psb = (int *)(lds + LY::PSB), *hib = (int *)(lds + LY::HIB), *jm = (int *)(lds + LY::JM);
  unsigned *pm = (unsigned *)(lds + LY::PM);
  // slot -> pivot index of stages k+2 / k+3 as bytes; within a 16-slot group the rows lk, lk+4, lk+8, lk+12 are adjacent:
  // one 32-bit read gives a lane the four rows of a tile it holds
  unsigned char *jmb = (unsigned char *)jm;
  double *Minv = lds + LY::MIV;
  double *ksm = (double *)((char *)lds + (KRON ? P.kron_lds_off : 0));   // KRON: the 33 sums of the Kronecker blocks of the record about to be assembled
  // record buffers: [dbuf 0][dbuf 1][sbuf 0][sbuf 1] with DMA (record s lives in buffer s & 1), one of each without
  constexpr bool DMA = F <= 128;
  constexpr int NBUF = DMA ? 2 : 1;
  const int dstride = (int)kkt2_dbuf_doubles(F, P.max_drec), sstride = (int)kkt2_sbuf_ints(F, P.max_srec);
  double *const dbuf0 = dbuf;
  int *const sbuf0 = (int *)(dbuf + NBUF * dstride);
  int *sbuf = sbuf0;
  int *hiall = sbuf0 + NBUF * sstride;
  double *A = (double *)(hiall + ((NS + 4) & ~3));   // cells of the assembled entries
  const double *stream = W.stream + (size_t)b * P.stream_len;
  double *panel = W.panel + (size_t)b * P.panel_stride;
  double *dx = W.dx + (size_t)b * n;
  const int pstride = (F + 1) * PIV;   // per stage: w (16), V (F x 16)

  // ---- U tiles of this wave: update index uw (Kkt2Cfg), tile t = uw + NU i of the lower triangle ---------
  const int uw = (wv & 3) ? wv - 1 - (wv >> 2) : (wv == 4 ? 12 : (wv == 8 ? 13 : (wv == 12 ? 14 : 99)));
  const bool is_upd = uw < NU;
  d4_t U[MAXT2];
  int tRC[MAXT2];   // (R << 8) | C, or -1
#pragma unroll
  for (int i = 0; i < MAXT2; ++i) {
    U[i] = d4_t{0.0, 0.0, 0.0, 0.0};
    int t = uw + NU * i;
    bool tv = t < CF::NTILE;
    int R = 0;
    while (is_upd && tv && ((R + 1) * (R + 2)) >> 1 <= t) ++R;
    const bool valid = is_upd && tv;
    tRC[i] = valid ? (R << 8) | (t - ((R * (R + 1)) >> 1)) : -1;
  }

  int tile_lane = (li * PLD + lk) * 8;   // (li, lk): row li, column lk of a 16 x 16 tile of a panel, in bytes
  asm volatile("" : "+v"(tile_lane));
  unsigned ge4_keep = 0u, gt4_keep = 0u;
#pragma unroll
  for (int g = 0; g < 4; ++g) { ge4_keep |= (lk + 4 * g >= li ? 1u : 0u) << (4 * g); gt4_keep |= (lk + 4 * g > li ? 1u : 0u) << (4 * g); }
  for (int i = tid; i < P.n_cells; i += KT2) A[i] = 0.0;
  for (int i = tid; i < 3 * PSZ; i += KT2) PB[i] = 0.0;
  for (int i = tid; i < FR; i += KT2) { UF[i] = 0.0; xs[i] = 0.0; }
  for (int v = tid; v < n; v += KT2) dx[v] = 0.0;
  __syncthreads();

  auto header_from_lds = [&](int s) __attribute__((always_inline)) {
    if (tid < 8) pm[(s & 1) * 8 + tid] = 0u;
    if (tid < PIV) {
      const int slot = sbuf[SHDR + tid];
      psb[(s % 3) * PIV + tid] = slot;
      jmb[(s & 1) * FR + (slot & ~15) + (slot & 3) * 4 + ((slot >> 2) & 3)] = (unsigned char)tid;
      dgb[(s % 3) * PIV + tid] = dbuf[tid];
      atomicOr(&pm[(s & 1) * 8 + (slot >> 5)], 1u << (slot & 31));
    }
    if (tid == 0) { hib[s % 3] = sbuf[3]; hiall[s] = (sbuf[3] + 15) & ~15; }
  };
  auto load_records = [&](int s) __attribute__((always_inline)) {
    const int s0 = P.srec_off[s], s1 = P.srec_off[s + 1], d0 = P.drec_off[s], d1 = P.drec_off[s + 1];
    for (int i = tid; i < s1 - s0; i += KT2) sbuf[i] = P.srec[s0 + i];
    for (int i = tid; i < d1 - d0; i += KT2) dbuf[i] = stream[d0 + i];
  };
  // wave 0: LDL^T + L^-1 + (L D L^T)^-1 of the pivot block of the panel Pn, then the pivot rows leave the panel
  double *minv_g = W.minv + (size_t)b * NS * (PIV * PIV);   // inverse of every pivot block, kept for chord steps
  auto factor_block = [&](double *Pn, const int myps, const int *ps, double *Lin, double *dvn, int ks) __attribute__((always_inline)) {
    // split layout (ldlt16s): lane (li, lk) holds row li of the block, columns c = 4 g + lk.  The pivot rows of the panel
    // carry the block's lower triangle (by pivot index): the entries above the diagonal are read from the mirrored
    // position, so the block that is factored is exactly symmetric (ps = the stage's pivot slots in LDS)
    double a[4], wi[4], myinv;
    double *prow_p = Pn + myps * PLD + lk;
    int pc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) pc[g] = ps[4 * g + lk];
#pragma unroll
    for (int g = 0; g < 4; ++g) a[g] = 4 * g + lk > li ? Pn[pc[g] * PLD + li] : prow_p[4 * g];
    ldlt16s(a, wi, myinv, li, lk);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      Lin[li * PLD + 4 * g + lk] = wi[g];
      prow_p[4 * g] = 0.0;   // the pivot rows leave the panel
    }
    if (lk == (li & 3)) dvn[li] = myinv;
    {
      double zero = 0.0;
      asm volatile("" : "+v"(zero));
      d4_t mi = {zero, zero, zero, zero};
      double lt[4], ld[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) { lt[s4] = Lin[(lk + 4 * s4) * PLD + li]; ld[s4] = dvn[lk + 4 * s4]; }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) mi = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[s4], lt[s4] * ld[s4], mi, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        Minv[(lk + 4 * g) * PLD + li] = mi[g];
        minv_g[(size_t)ks * (PIV * PIV) + (lk + 4 * g) * PIV + li] = mi[g];
      }
    }
  };
  auto assemble_continuations = [&](int t0, int nth) __attribute__((always_inline)) {
    if constexpr (!CONT) return;
    // (continuation records are fetched by the whole workgroup: only used from the prologue and from
    //  a phase in which every wave takes part in the assembly)
    const int n_cont = __builtin_amdgcn_readfirstlane(sbuf[6]), c_first = __builtin_amdgcn_readfirstlane(sbuf[7]);
    for (int c = 0; c < n_cont; ++c) {
      lds_barrier();
      const int *co = P.cont + 4 * (c_first + c);
      const int so = co[0], sl = co[1], dof = co[2], dl = co[3];
      for (int i = threadIdx.x; i < dl; i += KT2) dbuf[i] = stream[dof + i];
      for (int i = threadIdx.x; i < sl; i += KT2) sbuf[i] = P.srec[so + i];
      lds_barrier();
      if (t0 >= 0) assemble_stage(A, F, sbuf, dbuf, t0, nth);
    }
  };

  // ---- prologue: assemble stages 0 and 1, gather and factor the pivot block of stage 0, leave the
  //      records of stage 2 in LDS ----------------------------------------------------------------
  auto point_buffers = [&](int s) __attribute__((always_inline)) {   // the buffer that holds the records of stage s
    if constexpr (DMA) { dbuf = dbuf0 + (s & 1) * dstride; sbuf = sbuf0 + (s & 1) * sstride; }
  };
  point_buffers(0);
  load_records(0);
  __syncthreads();
  header_from_lds(0);
  if constexpr (KRON) kron_sums(sbuf, dbuf, ksm, wv, 16, lane);
  __syncthreads();
  if constexpr (KRON) assemble_stage_kron(A, F, sbuf, dbuf, ksm, tid, KT2); else
  assemble_stage(A, F, sbuf, dbuf, tid, KT2);
  assemble_continuations(tid, KT2);
  __syncthreads();
  {
    double *P0 = PB;
    const int *ps0 = psb;
    auto cell0 = [&](int r, int j) __attribute__((always_inline)) {
      return r < F ? (int)P.ctab[(((r >> 4) * 64) + (r & 3) * 16 + j) * 4 + ((r & 15) >> 2)] : P.rtab[j];
    };
    for (int i = tid; i < (F + 1) * PIV; i += KT2) {
      const int r = i >> 4, j = i & 15, c = ps0[j];
      P0[r * PLD + j] = A[cell0(r, j)] + (r == c ? dgb[j] : 0.0);
    }
    __syncthreads();
    for (int i = tid; i < (F + 1) * PIV; i += KT2) {
      const int r = i >> 4, j = i & 15;
      A[cell0(r, j)] = 0.0;
    }
    __syncthreads();
    if (wv == 0) factor_block(P0, ps0[li], ps0, Lib, dvb, 0);
  }
  for (int s = 1; s < 3 && s < NS; ++s) {
    point_buffers(s);
    load_records(s);
    __syncthreads();
    header_from_lds(s);
    if constexpr (KRON) { if (s == 1) kron_sums(sbuf, dbuf, ksm, wv, 16, lane); }
    __syncthreads();
    if (s == 1) {
      if constexpr (KRON) assemble_stage_kron(A, F, sbuf, dbuf, ksm, tid, KT2); else
      assemble_stage(A, F, sbuf, dbuf, tid, KT2);
      assemble_continuations(tid, KT2);
    }
    __syncthreads();
  }

#ifdef QTOS_STAMPS
  // diagnostic build: per wave, cycles spent in each part of a stage (accumulated in LDS by lane 0)
  // (the counters live in the trace rows themselves -- fire-and-forget atomics of lane 0 --: no LDS, so that the diagnostic
  //  build has the product's LDS layout and fits the variants that fill it)
  unsigned long long *st2g = (unsigned long long *)(W.trace + ((size_t)b * (P.max_iter + 1) + 16) * 4);
  unsigned long long ts_ = 0;
  if (tid < 192) st2g[tid] = 0ull;
  __syncthreads();
#define KS2_START() do { if (lane == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory"); } while (0)
#define KS2(i) do { if (lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); atomicAdd(st2g + wv * 12 + (i), t_ - ts_); ts_ = t_; } } while (0)
  KS2_START();
#else
#define KS2_START() do {} while (0)
#define KS2(i) do {} while (0)
#endif
  // per-thread prefetch registers: one 128-bit load of doubles and two of ints cover the longest record
  d2_t pfd;
  i4_t pfs0, pfs1;
  int pf_nd2 = 0, pf_ns4 = 0;
  int prow_next = NS > 1 ? psb[PIV + li] : 0;   // pivot slot li of stage k+1
  typedef unsigned short us4_t __attribute__((ext_vector_type(4)));
  const us4_t *ctab4 = (const us4_t *)P.ctab;
  us4_t ct_cur = ctab4[((size_t)min(1, NS - 1) * NT + min(wv, NT - 1)) * 64 + lane];
  int rc_cur = P.rtab[min(1, NS - 1) * PIV + li];   // cell of the assembled rhs of pivot li of stage k+1
  const int tid_outer = tid, lane_outer = lane;
  for (int k = 0; k < NS; ++k) {
    int tid = tid_outer, lane = lane_outer;
    asm volatile("" : "+v"(tid), "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    const int pb = (k & 1) ? 2 : 0;
    double *Pk = PB + pb * PSZ, *Yk = PB + PSZ, *Xn = PB + (2 - pb) * PSZ;
    const double *Lik = Lib + (k & 1) * PIV * PLD, *dik = dvb + (k & 1) * PIV;
    const bool has_next = k + 1 < NS;
    KS2(7);
    // ---- install the records of stage k+2 (prefetched during phase C of the previous stage; the records of
    //      stage 2 are in LDS since the prologue).  Wave w holds elements 64 rank(w) ..: the record's header sits
    //      in the registers of wave 12 (rank 0), which publishes it. ------------------------------------
    // (rank of a wave = how early it can spare the time: header wave, the waves without Schur tiles, update
    //  waves oldest first, factor wave last; high ranks lie behind the end of most records and skip the loads)
    const int pidx = (int)((0xEDC0BA928761543Full >> (4 * wv)) & 15u) * 64 + lane;
    point_buffers(k + 2);      // (with LDS-DMA the records of stage k+2 landed during the previous stage: nothing to install)
    if (!DMA && k >= 1 && k + 2 < NS) {
      const int wbase = __builtin_amdgcn_readfirstlane(pidx);
      if (wbase < pf_nd2) ((d2_t *)dbuf)[min(pidx, pf_nd2)] = pfd;
      if (wbase < pf_ns4) ((i4_t *)sbuf)[min(pidx, pf_ns4)] = pfs0;
      if (wbase + KT2 < pf_ns4) ((i4_t *)sbuf)[min(pidx + KT2, pf_ns4)] = pfs1;
    }
    const us4_t ct_nxt = ctab4[((size_t)min(k + 2, NS - 1) * NT + min(wv, NT - 1)) * 64 + lane];
    const int rc_nxt = P.rtab[min(k + 2, NS - 1) * PIV + li];
    // ---- AB(k) ----------------------------------------------------------------------------------------
    const Mask256 m1 = load_mask8(pm + ((k + 1) & 1) * 8, lane);   // pivot slots of stage k+1
    if (wv < NT) {
      const int R = wv;
      const unsigned am_word = P.amask[k * 8 + (R >> 1)];
      const int prow = has_next ? prow_next : 0;
      // operands in the order they are needed (register budget: 128 per lane with sixteen waves)
      double pr[4], lm[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        pr[s4] = Pk[(16 * R + li) * PLD + lk + 4 * s4];
        lm[s4] = Minv[li * PLD + lk + 4 * s4];
      }
      double zero = 0.0;
      asm volatile("" : "+v"(zero));   // (a loop-invariant zero pair would be kept across the loop -- and spilled)
      // V = P (L D L^T)^-1 in accumulator layout: vt[g] = V[16R+li][lk+4g] -- V itself as the A operand of the next product
      d4_t vt = {zero, zero, zero, zero};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) vt = __builtin_amdgcn_mfma_f64_16x16x4f64(lm[s4], pr[s4], vt, 0, 0, 0);
      // next pivot columns: assembled entries (cell table: 0 = the zero cell), extracted Schur updates, pivot
      // diagonal, minus V P[piv]^T (= Y D^-1 Y[piv]^T: the raw rows of the next pivots are the B operand)
      double npp[4], xv[4], av[4];
      int aidx[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int r = 16 * R + lk + 4 * s4;
        npp[s4] = Pk[prow * PLD + lk + 4 * s4];
        xv[s4] = Xn[r * PLD + li];
        aidx[s4] = ct_cur[s4];
        av[s4] = A[aidx[s4]];
      }
      const double dgn = dgb[((k + 1) % 3) * PIV + li];
      d4_t acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = 16 * R + lk + 4 * g;
        acc[g] = xv[g] + av[g] + (r == prow ? dgn : 0.0);
        npp[g] = -npp[g];
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vt[s4], npp[s4], acc, 0, 0, 0);   // acc -= V P[piv]^T
#pragma unroll
      for (int g = 0; g < 4; ++g) A[aidx[g]] = 0.0;   // retired (the zero cell stays zero)
      if constexpr (LY::VP) {
        // operands of the Schur update: -V (accumulator layout -> row-major) and the raw rows, next pivots' rows blanked
        const bool myrowpiv = has_next && ((grp16(m1, R) >> li) & 1u);
        double *Pm = lds + LY::PMB;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Yk[(16 * R + li) * PLD + lk + 4 * g] = myrowpiv ? 0.0 : -vt[g];
          Pm[(16 * R + li) * PLD + lk + 4 * g] = myrowpiv ? 0.0 : pr[g];
        }
      } else
      // Y = P L^-T of this tile: here for the tiles without a partner wave, else on service wave R (below)
      if (R == 0 || R >= NSV) {
        double la[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) la[s4] = Lik[li * PLD + lk + 4 * s4];
        d4_t yt = {zero, zero, zero, zero};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          yt = __builtin_amdgcn_mfma_f64_16x16x4f64(la[s4], pr[s4], yt, 0, 0, 0);   // yt[g] = Y[16R+li][lk+4g]
        const bool myrowpiv = has_next && ((grp16(m1, R) >> li) & 1u);
#pragma unroll
        for (int g = 0; g < 4; ++g) Yk[(16 * R + li) * PLD + lk + 4 * g] = myrowpiv ? 0.0 : yt[g];
      }
      if (has_next) {
#pragma unroll
        for (int g = 0; g < 4; ++g) Xn[(16 * R + lk + 4 * g) * PLD + li] = acc[g];
      }
      const unsigned am16 = (am_word >> ((R & 1) * 16)) & 0xffffu;
      if ((am16 >> li) & 1u) {
        double *pv = panel + (size_t)k * pstride + PIV;
        *(d4_t *)(pv + (16 * R + li) * PIV + 4 * lk) = vt;
      }
    } else {
      // ---- service waves ------------------------------------------------------------------------------
      const int sv = wv - NT;
      if (sv == 0) {
        // right-hand-side row: w = (L D L^T)^-1 p_F; rhs -= P w; right-hand side of the next pivots
        double part = 0.0;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) part = fma(Minv[li * PLD + lk + 4 * s4], Pk[F * PLD + lk + 4 * s4], part);
        part = rowsum4(part);                    // w[li] on every lane
        if (lane < PIV) panel[(size_t)k * pstride + lane] = part;
        if (has_next) {
          asm volatile("s_nop 4" : "+v"(part));    // DPP hazard distance for the broadcast reads below
#pragma unroll
          for (int c = 0; c < FR / 64; ++c) {
            const int r = c * 64 + lane;
            const int rr = min(r, F - 1);
            double pq[PIV];
#pragma unroll
            for (int q = 0; q < PIV; ++q) pq[q] = Pk[rr * PLD + q];
            const double uf = UF[r];
            double a0 = 0.0;
            dot16_steps<0>(a0, part, pq);
            UF[r] = r < F ? uf - a0 : 0.0;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane < PIV) {
            const int c = prow_next;
            Xn[F * PLD + lane] = A[rc_cur] + UF[c];
            A[rc_cur] = 0.0;
            UF[c] = 0.0;
          }
        }
      }
      else if (KRON && LY::VP) {
        // the waves that idle in this phase: the 33 sums of the Kronecker blocks of the records of stage k+2 (assembled in C(k))
        if (k + 2 < NS) kron_sums(sbuf, dbuf, ksm, sv - 1, NSV - 1, lane);
      }
      else if (!LY::VP && sv < NT) {
        // Y = P L^-T of row tile R = sv for its tile wave
        const int R = sv;
        double la[4], pr[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          la[s4] = Lik[li * PLD + lk + 4 * s4];
          pr[s4] = Pk[(16 * R + li) * PLD + lk + 4 * s4];
        }
        double zero = 0.0;
        asm volatile("" : "+v"(zero));
        d4_t yt = {zero, zero, zero, zero};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) yt = __builtin_amdgcn_mfma_f64_16x16x4f64(la[s4], pr[s4], yt, 0, 0, 0);
        const bool myrowpiv = has_next && ((grp16(m1, R) >> li) & 1u);
#pragma unroll
        for (int g = 0; g < 4; ++g) Yk[(16 * R + li) * PLD + lk + 4 * g] = myrowpiv ? 0.0 : yt[g];
      }
    }
    KS2(0);
    lds_barrier();
    KS2(1);
    // ---- C(k) ---------------------------------------------------------------------------------------
    // prefetch of the records of stage k+3 (installed at the top of the next stage): issued by every wave when
    // its urgent work of the phase is done (48 KB-wide loads at once keep the CU's memory pipe busy for ~800 cycles);
    // a wave skips the loads that lie wholly behind the record's end
    auto prefetch_records = [&]() __attribute__((always_inline)) {
      if (!DMA && k + 3 < NS) {
        const int s = k + 3;
        int d0, d1, s0, s1;
        sload2(P.drec_off + s, d0, d1);
        sload2(P.srec_off + s, s0, s1);
        pf_nd2 = (d1 - d0) >> 1;
        pf_ns4 = (s1 - s0) >> 2;
        const d2_t *dsrc = (const d2_t *)(stream + d0);
        const i4_t *ssrc = (const i4_t *)(P.srec + s0);
        const int wbase = __builtin_amdgcn_readfirstlane(pidx);   // first element of this wave
        if (wbase < pf_nd2) pfd = dsrc[min(pidx, pf_nd2 - 1)];
        if (wbase < pf_ns4) pfs0 = ssrc[min(pidx, pf_ns4 - 1)];
        if (wbase + KT2 < pf_ns4) pfs1 = ssrc[min(pidx + KT2, pf_ns4 - 1)];
      }
    };
    prefetch_records();
    KS2(8);
    if (wv == 0) {
      __builtin_amdgcn_s_setprio(3);
      if (has_next) factor_block(Xn, prow_next, psb + ((k + 1) % 3) * PIV, Lib + ((k + 1) & 1) * PIV * PLD, dvb + ((k + 1) & 1) * PIV, k + 1);
      __builtin_amdgcn_s_setprio(0);
    } else if (is_upd) {
      const Mask256 m2 = load_mask8(pm + (k & 1) * 8, lane);   // pivot slots of stage k+2
      const bool extract = k + 2 < NS;
      const unsigned char *jm2 = jmb + (k & 1) * FR;
      double *Xnn = Pk;   // the panel of stage k is dead: it receives the columns of stage k+2
      double dv4[4];
      if constexpr (!LY::VP) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) dv4[s4] = -dik[lk + 4 * s4];
      }
      const double *Bop = LY::VP ? lds + LY::PMB : Yk;
      int rcs[MAXT2];
#pragma unroll
      for (int t = 0; t < MAXT2; ++t) { rcs[t] = tRC[t]; asm volatile("" : "+s"(rcs[t])); }
      double wa[2][4], pbv[2][4];
      // operand addresses: a lane part that never changes (tile_lane, bytes) plus a wave-uniform tile offset formed on the
      // scalar unit
      auto tile_loads = [&](int rc, double (&w)[4], double (&pq)[4]) __attribute__((always_inline)) {
        const int R = rc < 0 ? 0 : rc >> 8, C = rc < 0 ? 0 : rc & 255;
        const int oR = __builtin_amdgcn_readfirstlane(R * (16 * PLD * 8)), oC = __builtin_amdgcn_readfirstlane(C * (16 * PLD * 8));
        const char *wrow = (const char *)Yk + (tile_lane + oR), *prow2 = (const char *)Bop + (tile_lane + oC);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { w[s4] = *(const double *)(wrow + 32 * s4); pq[s4] = *(const double *)(prow2 + 32 * s4); }
      };
      // pivot indices of the columns (li) and of the four rows (lk + 4g) this lane holds in each tile: fetched ahead of
      // the products (the extraction behind them starts with no LDS round trip of its own: -5.5 % per launch)
      int jc8[MAXT2], jr32[MAXT2];
#pragma unroll
      for (int t = 0; t < MAXT2; ++t) {
        const int rc = rcs[t] < 0 ? 0 : rcs[t];
        jc8[t] = jm2[16 * (rc & 255) + (li & 3) * 4 + (li >> 2)];
        jr32[t] = *(const int *)(jm2 + 16 * (rc >> 8) + 4 * lk);
      }
      tile_loads(rcs[0], wa[0], pbv[0]);
#pragma unroll
      for (int t = 0; t < MAXT2; ++t) {
        if (t + 1 < MAXT2) tile_loads(rcs[t + 1], wa[(t + 1) & 1], pbv[(t + 1) & 1]);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          U[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(LY::VP ? wa[t & 1][s4] : wa[t & 1][s4] * dv4[s4], pbv[t & 1][s4], U[t], 0, 0, 0);
      }
      KS2(5);
      if (extract) {
        int jcs[MAXT2], jrs[MAXT2][4];
#pragma unroll
        for (int t = 0; t < MAXT2; ++t) {
          jcs[t] = jc8[t];
#pragma unroll
          for (int g = 0; g < 4; ++g) jrs[t][g] = (jr32[t] >> (8 * g)) & 255;
        }
        double *dummy = red + 2 * 16 * PIV + lane;
        // bit 4g of ge4 / gt4: row lk + 4g of a diagonal tile lies on or below / strictly below column li
        const unsigned ge4 = ge4_keep, gt4 = gt4_keep;
#pragma unroll
        for (int t = 0; t < MAXT2; ++t) {
          const int rc = rcs[t];
          if (rc < 0) continue;
          const int R = rc >> 8, C = rc & 255;
          const unsigned cw2 = grp16(m2, C), rw2 = grp16(m2, R);
          if ((cw2 | rw2) == 0u) continue;
          const unsigned cm = ((cw2 >> li) & 1u) ? (R > C ? 0x1111u : ge4) : 0u;
          const unsigned rmk = (rw2 >> lk) & (R > C ? 0x1111u : gt4);
          double *xr = Xnn + (16 * R + lk) * PLD + jcs[t], *xc = Xnn + (16 * C + li) * PLD;
          if (cw2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) *(((cm >> (4 * g)) & 1u) ? xr + g * 4 * PLD : dummy) = U[t][g];
          }
          if (rw2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) *(((rmk >> (4 * g)) & 1u) ? xc + jrs[t][g] : dummy) = U[t][g];
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int z = ~__builtin_amdgcn_sbfe((int)(cm | rmk), 4 * g, 1);
            U[t][g] = __hiloint2double(__double2hiint(U[t][g]) & z, __double2loint(U[t][g]) & z);
          }
        }
      }
    }
    KS2(2);
    // every wave but the factor wave ends the phase with its share of the assembly of stage k+2's records
    // (the waves without a Schur tile come first and take the low item indices)
    const int apos = is_upd ? (15 - NU) + uw : uw - NU;   // (wave 12, the header wave, is the last of the free ones)
#ifndef QTOS_ASM_SKIP
#define QTOS_ASM_SKIP 3
#endif
    // one thread per target, low item indices to the waves that get here first (handing the waves without Schur tiles
    // more of it was tried -- they share the factor wave's SIMD and are the slowest at it).  The youngest update wave of
    // every SIMD gets its matrix instructions last and ends the phase: it takes no part in the assembly.
    constexpr int NASM = 15 - (NU >= 12 ? QTOS_ASM_SKIP : 0);
#ifndef QTOS_ASM_FREEW
#define QTOS_ASM_FREEW 1
#endif
    // the waves without Schur tiles (they share the factor wave's SIMD, which runs no matrix instructions in this phase)
    // take FREEW shares each
    constexpr int FREEW = NU == 12 ? QTOS_ASM_FREEW : 1, NFREE = 15 - NU, NVIRT = NASM + NFREE * (FREEW - 1);
    if (wv >= 1 && k + 2 < NS && apos < NASM) {
      if constexpr (KRON) {
        if (apos < NFREE) {
#pragma unroll
          for (int f = 0; f < FREEW; ++f) assemble_stage_kron(A, F, sbuf, dbuf, ksm, (apos * FREEW + f) * 64 + lane, NVIRT * 64);
        } else assemble_stage_kron(A, F, sbuf, dbuf, ksm, (apos + NFREE * (FREEW - 1)) * 64 + lane, NVIRT * 64);
      } else
      if (apos < NFREE) {
#pragma unroll
        for (int f = 0; f < FREEW; ++f) assemble_stage(A, F, sbuf, dbuf, (apos * FREEW + f) * 64 + lane, NVIRT * 64);
      } else assemble_stage(A, F, sbuf, dbuf, (apos + NFREE * (FREEW - 1)) * 64 + lane, NVIRT * 64);
    }
    if constexpr (CONT) {
      if (k + 2 < NS) assemble_continuations(wv >= 1 ? apos * 64 + lane : -1, 15 * 64);
    }
    // LDS-DMA of the records of stage k+3 into the other buffer, by the three waves without Schur tiles once
    // their assembly is done (1 KB per instruction, chunk c of a record by wave c mod 3; wave 12 takes the
    // chunks with the header it publishes below); the loads are waited for before the phase's barrier
    if constexpr (DMA) {
      if (!(wv & 3) && wv != 0 && k + 3 < NS) {
        const int s = k + 3, wi = wv == 12 ? 0 : wv >> 2;
        int d0, d1, s0, s1;
        sload2(P.drec_off + s, d0, d1);
        sload2(P.srec_off + s, s0, s1);
        const int nbd = (d1 - d0) * 8, nbs = (s1 - s0) * 4;
        const char *gd = (const char *)(stream + d0), *gs = (const char *)(P.srec + s0);
        typedef __attribute__((address_space(3))) char lds_char;
        lds_char *ld = (lds_char *)(dbuf0 + (s & 1) * dstride), *ls = (lds_char *)(sbuf0 + (s & 1) * sstride);
        for (int c = wi; c * 1024 < nbd; c += 3)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gd + min(c * 1024 + lane * 16, nbd - 16)), (__attribute__((address_space(3))) void *)(ld + c * 1024), 16, 0, 0);
        for (int c = wi; c * 1024 < nbs; c += 3)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gs + min(c * 1024 + lane * 16, nbs - 16)), (__attribute__((address_space(3))) void *)(ls + c * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    if (wv == 12) {
      // header of stage k+3, published to the LDS rings (their slots have no reader left in this phase:
      // stage k's pivot slots / diagonals, stage k+1's slot map and mask) from this wave's share of the
      // prefetched record: lane l holds ints 4l .. 4l+3 (static header 0..7, pivot slots 8..23) and doubles
      // 2l, 2l+1 (pivot diagonals 0..15)
      const int hs = k + 3;
      if (hs < NS) {
        if (lane < 8) pm[(hs & 1) * 8 + lane] = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (DMA) {   // the header sits in the first chunks, which this wave has just waited for
          const int *hb = sbuf0 + (hs & 1) * sstride;
          const double *hdb = dbuf0 + (hs & 1) * dstride;
          if (lane == 0) { const int hv = hb[3]; hib[hs % 3] = hv; hiall[hs] = (hv + 15) & ~15; }
          if (lane < PIV) {
            const int hv = hb[SHDR + lane];
            dgb[(hs % 3) * PIV + lane] = hdb[lane];
            psb[(hs % 3) * PIV + lane] = hv;
            jmb[(hs & 1) * FR + (hv & ~15) + (hv & 3) * 4 + ((hv >> 2) & 3)] = (unsigned char)lane;
            atomicOr(&pm[(hs & 1) * 8 + (hv >> 5)], 1u << (hv & 31));
          }
        } else {
        if (lane == 0) { hib[hs % 3] = pfs0[3]; hiall[hs] = (pfs0[3] + 15) & ~15; }
        if (lane < 8) { dgb[(hs % 3) * PIV + 2 * lane] = pfd[0]; dgb[(hs % 3) * PIV + 2 * lane + 1] = pfd[1]; }
        if (lane >= 2 && lane < 6) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int jidx = 4 * (lane - 2) + c, hv = pfs0[c];
            psb[(hs % 3) * PIV + jidx] = hv;
            jmb[(hs & 1) * FR + (hv & ~15) + (hv & 3) * 4 + ((hv >> 2) & 3)] = (unsigned char)jidx;
            atomicOr(&pm[(hs & 1) * 8 + (hv >> 5)], 1u << (hv & 31));
          }
        }
        }
      }
    }
    KS2(3);
    lds_barrier();
    KS2(4);
    if (k + 2 < NS) prow_next = psb[((k + 2) % 3) * PIV + li];
    ct_cur = ct_nxt;
    rc_cur = rc_nxt;
  }
  // ---- backward substitution (sweep_backward below: one barrier per stage, one-stage look-ahead) ---------------
  __syncthreads();  // drains the factor-panel stores: they are read back below
  KS2(7);
  {
    int *nxp = (int *)PB;   // (the panels are dead)
    for (int i = tid; i < NS * 4; i += KT2) nxp[i] = P.nxt_pack[i];
    for (int i = tid; i < NS * 8; i += KT2) nxp[NS * 4 + i] = (int)P.amask2[i];
    __syncthreads();
    const SweepDs sd = {W.stream + (size_t)b * P.stream_len, W.ds + (size_t)b * P.n_cons, W.g + (size_t)b * P.n_cons, W.s + (size_t)b * P.n_cons, W.trace ? W.trace + ((size_t)b * (P.max_iter + 1) + 64) * 4 : nullptr};
    sweep_backward<F>(P, panel, dx, W.sol + (size_t)b * P.n_stages * PIV, xs, red, nxp, wv, lane, lds + ((LY::PB + NS * 6 + 1) & ~1), sd);
  }
#ifdef QTOS_STAMPS
  KS2(6);
  __syncthreads();
  // (stamps2.py reads the 192 counters as 64-bit integers)
#endif
}
#ifdef QTOS_STAMPS
// (k_kkt3 / k_kkt4 keep their counters in LDS)
#undef KS2
#define KS2(i) do { if (lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st2[wv][i] += t_ - ts_; ts_ = t_; } } while (0)
#endif



// =================================================================================================
// k_chord: one KKT solve with the factorisation k_kkt2 left behind and a new right-hand side (QtosParams.chord_tol).
// Per stage the factor panel holds V_k = P_k B_k^-1 (rows of the live slots) and k_kkt2 also keeps B_k^-1:
//   forward   p_k = rhs_k + u[piv_k];   w_k = B_k^-1 p_k  (to the panel);   u -= V_k p_k
//   backward  x_k = w_k - V_k^T x       (sweep_backward)
// with the right-hand side in elimination order from k_step.  Both sweeps run with the one-stage look-ahead described
// above: one barrier per stage; the launch streams the panels twice and is bound by that.
//   wave 0          the chain: p_k, and the block of V_k that feeds p_{k+1}
//   waves 1 .. NT   u -= V_{k-1} p_{k-1} on the other rows of row tile wv - 1, one stage behind
//   wave NT + 1     w_{k-1} = B_{k-1}^-1 p_{k-1}
constexpr int KTC = 1024;
inline size_t chord_lds_bytes(int NS, int sw_steps) { return sizeof(int) * (size_t)NS * 12 + sweep_ds_lds_bytes(NS, sw_steps); }   // sweep tables; solution by position and rounds of the helper waves
template <int F>
__global__ __launch_bounds__(KTC) void k_chord(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x;
  if (b >= B || W.done[b] || W.chord[b] != 1) return;
  using CF = Kkt2Cfg<F>;
  constexpr int NT = CF::NT, FR = CF::FR;
  static_assert(NT + 1 < 16, "k_chord: waves 1..NT own the row tiles, wave NT + 1 solves with the pivot blocks");
  __shared__ double UF[FR], xs[FR], pf[2 * PIV], red[2 * 16 * PIV];
  extern __shared__ int nxp[];   // Symbolic::nxt_pack, then Symbolic::amask2
  const int tid = threadIdx.x, NS = P.n_stages, n = P.n_sol;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, q = lane >> 4;
  double *panel = W.panel + (size_t)b * P.panel_stride;
  const double *minv = W.minv + (size_t)b * NS * (PIV * PIV);
  const double *rhs = W.rhs + (size_t)b * P.n_unknowns;
  double *dx = W.dx + (size_t)b * n;
  constexpr int pstride = (F + 1) * PIV;
  if (tid == 0) atomicAdd(W.n_active + 1, -1);   // flag consumed
  for (int i = tid; i < FR; i += KTC) { UF[i] = 0.0; xs[i] = 0.0; }
  for (int i = tid; i < NS * 4; i += KTC) nxp[i] = P.nxt_pack[i];
  for (int i = tid; i < NS * 8; i += KTC) nxp[NS * 4 + i] = (int)P.amask2[i];
  for (int v = tid; v < n; v += KTC) dx[v] = 0.0;
  __syncthreads();
  // ---- forward -------------------------------------------------------------------------------------
  {
    const bool owner = wv >= 1 && wv <= NT, solver = wv == NT + 1;
    const int R = owner ? wv - 1 : 0;
    const int rowl = lane >> 2, jq = lane & 3;   // bulk rows: lane = (row of the tile, four columns jq + 4 m: one 32-byte piece of the V row)
    // ring slot of a stage: one 32-byte piece -- of the chain block on wave 0 (V_k[slot of pivot j of stage k+1][q + 4 m]), of
    // the wave's rows on waves 1..NT, of B_k^-1 on wave NT + 1 --, and on wave 0 the stage's right-hand side and pivot slots
    d4_t fv[SWD];
    double frhs[SWD];
    int fps[SWD];
    unsigned fam[SWD];
    auto load = [&](int s, d4_t &v, unsigned &am, double &rk, int &psj) __attribute__((always_inline)) {
      const int kk = min(s, NS - 1);
      const double *pk = panel + (size_t)kk * pstride;
      am = owner ? ((unsigned)nxp[NS * 4 + kk * 8 + (R >> 1)] >> ((R & 1) * 16)) & 0xffffu : 0u;
      const unsigned nr = ((unsigned)nxp[kk * 4 + (j & 3)] >> (8 * (j >> 2))) & 255u;   // slot of pivot j of stage kk + 1
      const double *src = wv == 0 ? pk + (nr != 255u ? PIV + (int)nr * PIV + 4 * q : 0)
                        : solver  ? minv + (size_t)kk * (PIV * PIV) + j * PIV + 4 * q
                                  : pk + (((am >> rowl) & 1u) ? PIV + (16 * R + rowl) * PIV + 4 * jq : 0);
      v = *(const d4_t *)src;
      if (wv == 0) {
        rk = rhs[min(kk * PIV + j, P.n_unknowns - 1)];
        psj = P.piv_slot[kk * PIV + j];
      }
    };
#pragma unroll
    for (int d = 0; d < SWD; ++d) load(d, fv[d], fam[d], frhs[d], fps[d]);
    double corr = 0.0;
    for (int k0 = 0; k0 <= NS; k0 += SWD) {
#pragma unroll
      for (int d = 0; d < SWD; ++d) {
        const int k = k0 + d, dp = (d + SWD - 1) % SWD;   // stage of the chain; ring slot of stage k - 1
        if (wv == 0 && k < NS) {
          const bool real = k * PIV + j < P.n_unknowns;     // (dummy pivots of a short last stage)
          const double p = (real ? frhs[d] : 0.0) + UF[fps[d]] - corr;
          if (lane < PIV) { pf[(k & 1) * PIV + j] = p; UF[fps[d]] = 0.0; }   // (the slots are retired)
          const unsigned nr = ((unsigned)nxp[k * 4 + (j & 3)] >> (8 * (j >> 2))) & 255u;
          double c[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) c[m] = nr != 255u ? fv[d][m] : 0.0;
          corr = rowsum4(dot4_by_row(p, c));   // what stage k adds to the rows of the pivots of stage k + 1
        }
        if (owner && k >= 1 && k <= NS) {
          // u[row] -= V_{k-1}[row, :] p_{k-1} on the rows that are not pivots of stage k
          const double *pp = pf + ((k - 1) & 1) * PIV + jq;
          const bool on = (fam[dp] >> rowl) & 1u;
          double t = fv[dp][0] * pp[0];
          t = fma(fv[dp][1], pp[4], t);
          t = fma(fv[dp][2], pp[8], t);
          t = fma(fv[dp][3], pp[12], t);
          t = quadsum(on ? t : 0.0);
          if (on && jq == 0) UF[16 * R + rowl] -= t;
        }
        if (solver && k >= 1 && k <= NS) {
          // (lane (j, q) holds B^-1[j][4 q .. 4 q + 3])
          const double *pp = pf + ((k - 1) & 1) * PIV + 4 * q;
          double acc = fv[dp][0] * pp[0];
          acc = fma(fv[dp][1], pp[1], acc);
          acc = fma(fv[dp][2], pp[2], acc);
          acc = fma(fv[dp][3], pp[3], acc);
          acc = rowsum4(acc);
          if (lane < PIV) panel[(size_t)(k - 1) * pstride + j] = acc;
        }
        lds_barrier();
        // ring slot dp (stage k - 1) is free now
        if (k >= 1) load(k - 1 + SWD, fv[dp], fam[dp], frhs[dp], fps[dp]);
      }
    }
  }
  __syncthreads();   // the w entries written above are read back below (same workgroup: visible after the barrier)
  const SweepDs sd = {W.stream + (size_t)b * P.stream_len, W.ds + (size_t)b * P.n_cons, W.g + (size_t)b * P.n_cons, W.s + (size_t)b * P.n_cons, W.trace ? W.trace + ((size_t)b * (P.max_iter + 1) + 64) * 4 : nullptr};
  sweep_backward_early<F, true>(P, panel, dx, W.sol + (size_t)b * P.n_stages * PIV, xs, red, nxp, wv, lane, (double *)(nxp + ((NS * 12 + 3) & ~3)), sd);
}

// =================================================================================================
// k_residual: r = b - K x for the solution the last k_kkt2 / k_chord launch left in W.sol, WITHOUT the factorisation: K
// from the problem's stream (pivot diagonals, equality Jacobian values, inequality blocks G with their barrier weights),
// b by k_step's formula.  r goes to W.rhs in elimination order -- the right-hand side k_chord takes --, so that one more
// solve with the stored factorisation gives the correction of one step of iterative refinement (k_refine_add).
//   K = [ diag + Ji' S Ji   Je' ]      variables:   r_v = b_v - diag_v x_v - sum_r G[r][v] (sig_r (Ji x)_r) - (Je' y)_v
//       [ Je              diag  ]      multipliers: r_e = b_e - (Je x)_e - diag_e y_e
// out[b] = max |r| / max |b|;  keep != 0: x is remembered as the solution the correction will be added to.
__global__ __launch_bounds__(512) void k_residual(DevPlan P, DevWork W, int B, double *out, int keep) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= B) return;
  __shared__ double red[2][512];
  const int n = P.n_sol, m = P.n_cons, NU = P.n_unknowns, NP = P.n_stages * PIV;
  const double *Gs = W.stream + (size_t)b * P.stream_len, *g = W.g + (size_t)b * m, *sig = W.sig + (size_t)b * m, *wr = W.w + (size_t)b * m;
  const double *sol = W.sol + (size_t)b * NP, *dx = W.dx + (size_t)b * n;
  double *ur = W.ur + (size_t)b * m, *res = W.rhs + (size_t)b * NU;
  for (int i = tid; i < P.n_iq_rows; i += blockDim.x) {
    const IqRow R = P.iq_rows[i];
    double t = 0.0;
    for (int a = 0; a < R.n; ++a) t = fma(Gs[R.goff + a], dx[P.block_cols[R.col_off + a]], t);
    ur[R.row] = sig[R.row] * t;
  }
  __threadfence_block();
  __syncthreads();
  double mr = 0.0, mb = 0.0;
  for (int p = tid; p < NU; p += blockDim.x) {
    const int t0 = P.rhs_ptr[p], t1 = P.rhs_ptr[p + 1];
    const bool mult = t1 - t0 == 1 && P.rhs_gpos[t0] < 0;
    double rhs = 0.0, kx = Gs[P.diag_pos[p]] * sol[p];
    if (mult) rhs = -g[P.rhs_row[t0]];
    else
      for (int t = t0; t < t1; ++t) {
        const double gv = Gs[P.rhs_gpos[t]];
        rhs = fma(-gv, wr[P.rhs_row[t]], rhs);
        kx = fma(gv, ur[P.rhs_row[t]], kx);
      }
    for (int e = P.kx_ptr[p]; e < P.kx_ptr[p + 1]; ++e) kx = fma(Gs[P.kx_pos[e]], sol[P.kx_col[e]], kx);
    const double r = rhs - kx;
    res[p] = r;
    mr = fmax(mr, fabs(r));
    mb = fmax(mb, fabs(rhs));
  }
  red[0][tid] = mr; red[1][tid] = mb;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (tid < s) { red[0][tid] = fmax(red[0][tid], red[0][tid + s]); red[1][tid] = fmax(red[1][tid], red[1][tid + s]); }
    __syncthreads();
  }
  if (tid == 0 && out) out[b] = red[0][0] / fmax(red[1][0], 1e-300);
  if (keep) {
    double *sol0 = W.sol0 + (size_t)b * NP, *dx0 = W.dx0 + (size_t)b * n;
    for (int i = tid; i < NP; i += blockDim.x) sol0[i] = sol[i];
    for (int i = tid; i < n; i += blockDim.x) dx0[i] = dx[i];
  }
}
// x = x0 + correction (variables and the solution by unknown position)
__global__ __launch_bounds__(512) void k_refine_add(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= B) return;
  const int n = P.n_sol, NP = P.n_stages * PIV;
  double *sol = W.sol + (size_t)b * NP, *dx = W.dx + (size_t)b * n;
  const double *sol0 = W.sol0 + (size_t)b * NP, *dx0 = W.dx0 + (size_t)b * n;
  for (int i = tid; i < NP; i += blockDim.x) sol[i] += sol0[i];
  for (int i = tid; i < n; i += blockDim.x) dx[i] += dx0[i];
}

}  // namespace qtos
