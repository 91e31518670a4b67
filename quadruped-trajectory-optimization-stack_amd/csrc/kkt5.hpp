// kkt5.hpp -- k_kkt5: the factor + solve kernel with TWO 16-pivot stages -- a pair (a, b) -- behind one set of barriers
// (round 5; Symbolic::pair_mode).  Twelve waves of up to 168 registers per problem.
//
// k_kkt2's launch is  stages x (a latency skeleton of two barriers, each behind a chain of dependent LDS round trips);  its
// Schur operands travel through two LDS panels.  Here a pair of stages is ONE step of three barriers, and the operands of the
// Schur update never touch the LDS:
//
//   * row ownership: the wave that forms row tile R of the factor panels keeps the Schur tiles (R, 0 .. R) in its
//     accumulator registers; operand A of its updates is what it has just computed, operand B are the rows of the pair's
//     pivot columns as they stand in the LDS, blanked on load (a blanked lane reads a row of zeros);
//   * the pair step in "W form" (P1, P2: the pivot columns of a and b as they stand BEFORE the pair; A11, A21, A22: their rows
//     at the pair's own pivots; L21 = A21 A11^-1; S22 = A22 - L21 A21'):
//         V1 = P1 A11^-1      (= V_a, to HBM)       P2' = P2 - V1 A21'       V2 = P2' S22^-1   (= V_b, to HBM)
//         W1 = V1 - V2 L21,   W2 = V2       =>      everything after the pair  -=  W1 P1' + W2 P2'
//     sixteen chained matrix instructions per row tile with no LDS round trip between them, and nothing downstream needs the
//     updated P2' of ANOTHER wave: the next pair's columns and the Schur update use the original P1, P2 that sit in the LDS;
//   * per 16-pivot stage the panels V_k, w_k and the pivot-block inverses in HBM are those of k_kkt2: k_chord, the sweeps
//     and k_residual do not change.
//
//   phase 1   tile waves (one per row tile): V1, P2', V2, W1; the next pair's columns = cells + extracted Schur columns +
//             diagonal - W1 P1[piv]' - W2 P2[piv]'.     rhs wave: the right-hand-side row of all of that.
//             service waves: LDS-DMA of the record of pair j + 2 (one buffer), header of pair j + 2.
//   phase 2   tile waves: U -= W1 P1' + W2 P2' on their tiles.   factor wave: LDL^T(A11), A11^-1, L21, S22 of pair j + 1.
//             service waves: start the assembly of the record of pair j + 2.
//   phase 3   tile waves: the columns of pair j + 2 out of their tiles into the (dead) panels of pair j; then everybody but
//             the factor wave: the rest of the assembly.   factor wave: LDL^T(S22), S22^-1.
//
// Blanking (rows by slot): the rows of the pair's own pivots leave W before anything is applied (their slots retire; a
// pivot of the next pair that takes such a slot enters with the next pair: its "rows" in this pair's panels are the previous
// occupant's and read as zeros); the rows of the next pair's pivots leave both operands of the Schur update (their columns
// were extracted a pair ago and get this pair's update in the panel).
#pragma once
#include "kkt2.hpp"

namespace qtos {

constexpr int KT5 = 768;
constexpr int KKT5_WAVES = KT5 / 64;

template <int F>
struct Kkt5Cfg {
  static constexpr int NT = F / 16;
  static_assert(F % 16 == 0 && NT >= 2 && NT <= 9, "k_kkt5: fronts of 32 .. 144 slots");
  static constexpr int NSVC = KKT5_WAVES - 1 - NT;   // service waves
  static constexpr int FR = (F + 63) & ~63;
  static constexpr int PR = F + 2;                   // rows of a panel: the slots, the right-hand side, a row of zeros
  static constexpr int PSZ = PR * PLD;
  // Row tile of a wave, -1: none.  The factor wave (0) shares its SIMD with waves 4 and 8: a SIMD has one vector ALU and it
  // stands still while an f64 matrix instruction runs, so the tiles go to the other three SIMDs heaviest first (row tile R
  // has R + 1 Schur tiles), serpentine, and the two lightest rows to waves 4 and 8.
#ifndef QTOS_K5_PLACE
#define QTOS_K5_PLACE 0
#endif
  // (PLACE 1: the service waves on the factor wave's SIMD -- no matrix instruction ever blocks their vector instructions --,
  //  all row tiles on the other three)
  static constexpr int slot_wave(int i) {
    constexpr int o0[9] = {1, 2, 3, 7, 6, 5, 4, 8, 9}, o1[9] = {1, 2, 3, 7, 6, 5, 9, 10, 11};
    return QTOS_K5_PLACE == 1 ? o1[i] : o0[i];
  }
  static constexpr int row_of_wave(int w) {
    for (int i = 0; i < NT; ++i) if (slot_wave(i) == w) return NT - 1 - i;
    return -1;
  }
  static constexpr unsigned long long packed_rows() {   // 4 bits per wave: row tile + 1
    unsigned long long v = 0;
    for (int w = 0; w < KKT5_WAVES; ++w) v |= (unsigned long long)(row_of_wave(w) + 1) << (4 * w);
    return v;
  }
  static constexpr int svc_index(int w) {   // rank among the service waves, -1: not one
    if (w == 0 || row_of_wave(w) >= 0) return -1;
    int r = 0;
    for (int v = 1; v < w; ++v) if (row_of_wave(v) < 0) ++r;
    return r;
  }
  static constexpr unsigned long long packed_svc() {
    unsigned long long v = 0;
    for (int w = 0; w < KKT5_WAVES; ++w) v |= (unsigned long long)(svc_index(w) + 1) << (4 * w);
    return v;
  }
};

// LDS layout (doubles)
template <int F>
struct Kkt5Layout {
  using CF = Kkt5Cfg<F>;
  static constexpr int MIV = 0;                         // 2 x 16 x PLD   A11^-1, S22^-1 of the pair being eliminated
  static constexpr int L21 = MIV + 2 * PIV * PLD;       // 16 x PLD       L21 = A21 A11^-1, row major
  static constexpr int LIN = L21 + PIV * PLD;           // 16 x PLD       scratch of the factor wave (L^-1; S22)
  static constexpr int DVN = LIN + PIV * PLD;           // 16             1 / d of the block being factored
  static constexpr int RSC = DVN + PIV;                 // 4 x 16         scratch of the right-hand-side wave
  static constexpr int DGB = RSC + 4 * PIV;             // 3 x 32         pivot diagonals (ring by pair % 3)
  static constexpr int UF = DGB + 3 * 2 * PIV;          // FR             accumulated right-hand-side updates by slot
  static constexpr int XS = UF + CF::FR;                // FR             solution by slot (backward sweep)
  static constexpr int RED = XS + CF::FR;               // 2 x 16 x 16 partial sums of the sweep + 64 dummy slots
  static constexpr int PSB = RED + 2 * 16 * PIV + 64;   // 3 x 32 ints    pivot slots (ring)
  static constexpr int PM = PSB + 3 * 2 * PIV / 2;      // 3 x 8 words    pivot-slot bit masks (ring)
  static constexpr int JM = PM + 3 * 8 / 2;             // FR ushorts     slot -> column offset of its pivot in the pair's two panels
  static constexpr int PAN = (JM + CF::FR / 4 + 1) & ~1;  // 4 panels of PR x PLD: pair q in panels 2 (q & 1), 2 (q & 1) + 1
  static constexpr int VAR = (PAN + 4 * CF::PSZ + 1) & ~1;   // dbuf, then (ints) sbuf, then the cells A
};
__host__ __device__ inline size_t kkt5_dbuf_doubles(int max_drec) { return (((size_t)max_drec * 8 + 1023) & ~(size_t)1023) / 8; }
__host__ __device__ inline size_t kkt5_sbuf_ints(int max_srec) { return (((size_t)max_srec * 4 + 1023) & ~(size_t)1023) / 4; }
inline size_t kkt5_fixed_doubles(int F) {
  const int FR = (F + 63) & ~63;
  size_t o = 2 * PIV * PLD + PIV * PLD + PIV * PLD + PIV + 4 * PIV + 3 * 2 * PIV + 2 * (size_t)FR + 2 * 16 * PIV + 64 + 3 * 2 * PIV / 2 + 3 * 8 / 2;
  o = (o + FR / 4 + 1) & ~(size_t)1;
  return o;
}
// what the backward sweep keeps in LDS in front of the helper waves' tables
inline size_t kkt5_sweep_base_bytes(int F, int NS) { return kkt5_fixed_doubles(F) * sizeof(double) + (size_t)NS * 12 * sizeof(int); }
inline size_t kkt5_lds_bytes(int F, int NS, int max_srec, int max_drec, int n_cells) {
  const size_t PSZ = (size_t)(F + 2) * PLD;
  size_t o = (kkt5_fixed_doubles(F) + 4 * PSZ + 1) & ~(size_t)1;
  o += kkt5_dbuf_doubles(max_drec);
  size_t oi = 2 * o + kkt5_sbuf_ints(max_srec);
  oi += 2 * (((size_t)n_cells + 1) & ~(size_t)1);
  return std::max(oi * sizeof(int), kkt5_sweep_base_bytes(F, NS));
}

// assembly of a pair record (Symbolic::pair_mode: 32 pivot slots / diagonals in front, target words (cell << 13) | first contribution).
// The waves that assemble do nothing else at that time: a thread's chain of dependent LDS round trips is what a round costs, so
// four equality entries / W contributions of a target are in flight at once; the order of a target's sum does not change.
__device__ __forceinline__ void assemble_pair_eq(double *A, const int *sbuf, const double *dbuf, int t0, int nth) {
  const int n = sbuf[0] + sbuf[1];   // equality entries, then the multipliers' right-hand sides: distinct cells
  const int *eidx = sbuf + SHDR + 2 * PIV;
  const double *eval = dbuf + 2 * PIV;
  // (every such cell has ONE contribution in the whole plan and was left zero by the gather that retired it: a store)
  for (int i = t0; i < n; i += 4 * nth) {
    int idx[4];
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int ii = min(i + u * nth, n - 1); idx[u] = eidx[ii]; v[u] = eval[ii]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * nth < n) A[idx[u]] = v[u];
  }
}
template <int W>
__device__ __forceinline__ void assemble_pair_tgt(double *A, const int *sbuf, const double *dbuf, int t_begin, int t_end, int t0, int nth) {
  const int n_tgt = min(sbuf[5], t_end);
  const int *tg = sbuf + sbuf[4];
  const int *cl = tg + sbuf[5] + 1;
  for (int t = t_begin + t0; t < n_tgt; t += nth) {
    const int tv = tg[t], c0 = tv & 8191, c1 = tg[t + 1] & 8191;
    const double a_old = A[tv >> 13];
    double acc = 0;
    for (int j = c0; __builtin_amdgcn_ballot_w64(j < c1) != 0ull; j += W) {
      int code[W];
      double term[W];
#pragma unroll
      for (int w = 0; w < W; ++w) code[w] = cl[min(j + w, c1 - 1)];
#pragma unroll
      for (int w = 0; w < W; ++w) term[w] = gather_term(dbuf, code[w]);
#pragma unroll
      for (int w = 0; w < W; ++w) acc = j + w < c1 ? acc + term[w] : acc;
    }
    A[tv >> 13] = a_old + acc;
  }
}

#ifndef QTOS_K5_SB
#define QTOS_K5_SB 0
#endif
#if QTOS_K5_SB
#define K5_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define K5_SCHED_BARRIER() do {} while (0)
#endif
#ifndef QTOS_K5_ILPS
#define QTOS_K5_ILPS 1   // contributions of a target in flight: service waves
#endif
#ifndef QTOS_K5_ILPT
#define QTOS_K5_ILPT 1   // ... tile waves (their Schur tiles take half the registers)
#endif
#ifndef QTOS_K5_SKIP
#define QTOS_K5_SKIP 1   // skip the matrix instructions of row tiles / Schur tiles whose 16-slot group holds no unknown during the pair
#endif
#ifndef QTOS_K5_ASMT
#define QTOS_K5_ASMT 8   // row tiles 0 .. ASMT - 1 take part in the assembly of phase 3 (the heavy rows have their extraction)
#endif
#ifndef QTOS_K5_ASM2
#define QTOS_K5_ASM2 1   // rounds of targets the service waves assemble during phase 2 (the tile waves join in phase 3)
#endif

template <int F>
__global__ __launch_bounds__(KT5) void k_kkt5(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x;
  if (b >= B || W.done[b] || W.chord[b] == 1) return;   // (a problem flagged for a chord step is k_chord's)
  extern __shared__ double lds[];
  using CF = Kkt5Cfg<F>;
  using LY = Kkt5Layout<F>;
  constexpr int NT = CF::NT, NSVC = CF::NSVC, FR = CF::FR, PSZ = CF::PSZ, ZROW = F + 1;
  const int tid = threadIdx.x, NS = P.n_stages, NP = NS >> 1, n = P.n_sol;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int R = __builtin_amdgcn_readfirstlane((int)((CF::packed_rows() >> (4 * wv)) & 15) - 1);     // row tile of this wave or -1
  const int sidx = __builtin_amdgcn_readfirstlane((int)((CF::packed_svc() >> (4 * wv)) & 15) - 1);   // rank among the service waves or -1
  const bool is_tile = R >= 0, is_fac = wv == 0, is_svc = sidx >= 0;
  double *MIVa = lds + LY::MIV, *MIVb = MIVa + PIV * PLD, *L21 = lds + LY::L21, *LIN = lds + LY::LIN, *DVN = lds + LY::DVN;
  double *RSC = lds + LY::RSC, *dgb = lds + LY::DGB, *UF = lds + LY::UF, *xs = lds + LY::XS, *red = lds + LY::RED;
  int *psb = (int *)(lds + LY::PSB);
  unsigned *pm = (unsigned *)(lds + LY::PM);
  unsigned short *jm = (unsigned short *)(lds + LY::JM);
  double *PAN = lds + LY::PAN;
  const int dstride = (int)kkt5_dbuf_doubles(P.max_drec), sstride = (int)kkt5_sbuf_ints(P.max_srec);
  double *const dbuf = lds + LY::VAR;
  int *const sbuf = (int *)(dbuf + dstride);
  double *A = (double *)(sbuf + sstride);   // cells of the assembled entries
  const double *stream = W.stream + (size_t)b * P.stream_len;
  double *panel = W.panel + (size_t)b * P.panel_stride;
  double *minv_g = W.minv + (size_t)b * NS * (PIV * PIV);
  double *dx = W.dx + (size_t)b * n;
  constexpr int pstride = (F + 1) * PIV;   // per stage: w (16), V (F x 16)
  auto pan = [&](int pair, int t) __attribute__((always_inline)) { return PAN + (2 * (pair & 1) + t) * PSZ; };

  for (int i = tid; i < P.n_cells; i += KT5) A[i] = 0.0;
  for (int i = tid; i < 4 * PSZ; i += KT5) PAN[i] = 0.0;
  for (int i = tid; i < LY::PAN; i += KT5) lds[i] = 0.0;   // (everything in front of the panels)
  for (int v = tid; v < n; v += KT5) dx[v] = 0.0;
  __syncthreads();

  // header of a pair into ring slot pair % 3: pivot slots, diagonals, slot mask (one wave; from global memory)
  auto publish_header = [&](int q, bool with_jm) __attribute__((always_inline)) {
    const int rs = q % 3;
    if (lane < 8) pm[rs * 8 + lane] = 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < 2 * PIV) {
      const int slot = P.piv_slot[q * 2 * PIV + lane];
      const double dg = stream[P.drec_off[q] + lane];
      psb[rs * 2 * PIV + lane] = slot;
      dgb[rs * 2 * PIV + lane] = dg;
      atomicOr(&pm[rs * 8 + (slot >> 5)], 1u << (slot & 31));
      // slot -> where the pivot's column sits in the pair's two panels; within a 16-slot group the rows lk, lk + 4, lk + 8,
      // lk + 12 are adjacent: one 64-bit read gives a lane the four rows of a tile it holds
      if (with_jm) jm[(slot & ~15) + (slot & 3) * 4 + ((slot >> 2) & 3)] = (unsigned short)((lane >> 4) * PSZ + (lane & 15));
    }
  };
  typedef unsigned short us4_t __attribute__((ext_vector_type(4)));
#ifdef QTOS_STAMPS
  // diagnostic build: per wave, cycles spent in each part of a step (fire-and-forget atomics of lane 0 into the trace rows):
  // 0 phase 1 work | 1 barrier 1 | 2 phase 2 work | 3 barrier 2 | 4 phase 3 role work | 5 phase 3 assembly | 6 barrier 3 | 7 sweep
  unsigned long long *st5 = (unsigned long long *)(W.trace + ((size_t)b * (P.max_iter + 1) + 16) * 4);
  unsigned long long ts_ = 0;
  if (tid < 192) st5[tid] = 0ull;
  __syncthreads();
#define K5S(i) do { if (lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); atomicAdd(st5 + wv * 12 + (i), t_ - ts_); ts_ = t_; } } while (0)
  if (lane == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory");
#else
#define K5S(i) do {} while (0)
#endif
  // per-step pointers (pair q lives in panels 2 (q & 1), 2 (q & 1) + 1; rings by pair % 3)
#define K5_STEP_POINTERS \
    const bool live = j >= 0, has_next = j + 1 < NP, has_next2 = j + 2 < NP; \
    const double *PA0 = pan(j, 0), *PA1 = pan(j, 1); \
    double *PN0 = pan(j + 1, 0); \
    const int *psj = psb + (((j + 3) % 3) * 2 * PIV), *psn = psb + (((j + 4) % 3) * 2 * PIV); \
    const unsigned *pmj = pm + ((j + 3) % 3) * 8, *pmn = pm + ((j + 4) % 3) * 8, *pmn2 = pm + ((j + 5) % 3) * 8; \
    (void)live; (void)has_next; (void)has_next2; (void)PA0; (void)PA1; (void)PN0; (void)psj; (void)psn; (void)pmj; (void)pmn; (void)pmn2;
  constexpr int NASMT = QTOS_K5_ASMT < NT ? QTOS_K5_ASMT : NT;
  const int n_tgt2 = NSVC * 64 * QTOS_K5_ASM2;   // targets of a record the service waves take during phase 2

  // Role-specialised loops with the same three barriers per step: the register budget of a wave is its own role's (the Schur
  // tiles are live across the whole loop of the tile waves only).
  if (is_tile) {
    // ================================================ tile waves ================================================
    d4_t U[NT];   // Schur tiles (R, C), C = 0 .. R
#pragma unroll
    for (int c = 0; c < NT; ++c) U[c] = d4_t{0.0, 0.0, 0.0, 0.0};
    d4_t nW1 = {0.0, 0.0, 0.0, 0.0}, nW2 = {0.0, 0.0, 0.0, 0.0};   // -W1, -W2 of the pair (rows 16 R + li, columns lk + 4 g)
    const us4_t *ctab4 = (const us4_t *)P.ctab;
    // table words of the steps ahead, fetched a pair early (from global memory they are a microsecond away): cells of the
    // columns of pair j + 1 (ctab), row masks of the panels of pair j (amask)
    us4_t ct_cur[2] = {us4_t{0, 0, 0, 0}, us4_t{0, 0, 0, 0}}, ct_nxt[2];
    unsigned am_cur[2] = {0u, 0u}, am_nxt[2];
    unsigned pg_cur = 0xffffu, pg_nxt;   // occupied 16-slot groups of pair j (Symbolic::pair_groups)
    const int lane_outer = lane;
    for (int j = -2; j < NP; ++j) {
      K5_STEP_POINTERS
      // (the lane's coordinates re-derived per step from an opaque copy: the address arithmetic below is recomputed, not
      //  kept in registers across the loop -- at 168 registers every invariant the compiler hoists is a spill, and a spill
      //  reload waits on the in-order memory counter for the table words in flight)
      int lane = lane_outer;
      asm volatile("" : "+v"(lane));
      const int li = lane & 15, lk = lane >> 4, row = 16 * R + li;
      const int tile_lane = (li * PLD + lk) * 8;   // (li, lk): row li, column lk of a 16 x 16 tile of a panel, in bytes
      // (the table words fetched during the last step are taken NOW: the memory counter is in order, and a wait for them
      //  placed behind the factor-panel stores below would wait for those stores)
      asm volatile("" : "+v"(ct_cur[0]), "+v"(ct_cur[1]), "+v"(am_cur[0]), "+v"(am_cur[1]), "+v"(pg_cur));
      const unsigned pg = QTOS_K5_SKIP ? (unsigned)__builtin_amdgcn_readfirstlane((int)pg_cur) : 0xffffu;
      const bool rowlive = (pg >> R) & 1u;   // (an empty group: its rows of the pair's columns are zeros, V and W of them too)
      // ---- phase 1 ----
      if (j >= -1) {
        d4_t w1 = {0.0, 0.0, 0.0, 0.0}, w2 = {0.0, 0.0, 0.0, 0.0};
        if (live && rowlive) {
          const int a = 2 * j;
          const unsigned ama = am_cur[0], amb = am_cur[1];
          const int sb_li = psj[PIV + li];                        // slot of pivot li of stage b
          const unsigned wj = pmj[R >> 1];
          const bool retiring = (wj >> ((R & 1) * 16 + li)) & 1u;   // this lane's row belongs to a pivot of the pair
          // operands in the order they are needed (the compiler would hoist all 24 loads to the top: 48 registers)
          double pa[4], ma[4], a21[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            pa[s] = PA0[row * PLD + lk + 4 * s];
            ma[s] = MIVa[li * PLD + lk + 4 * s];
            a21[s] = -PA0[sb_li * PLD + lk + 4 * s];
          }
          double zero = 0.0;
          asm volatile("" : "+v"(zero));
          d4_t v1 = {zero, zero, zero, zero};
#pragma unroll
          for (int s = 0; s < 4; ++s) v1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma[s], pa[s], v1, 0, 0, 0);       // V1 = P1 A11^-1
          double mb[4];
          d4_t p2;
#pragma unroll
          for (int s = 0; s < 4; ++s) { p2[s] = PA1[row * PLD + lk + 4 * s]; mb[s] = MIVb[li * PLD + lk + 4 * s]; }
          K5_SCHED_BARRIER();
#pragma unroll
          for (int s = 0; s < 4; ++s) p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a21[s], v1[s], p2, 0, 0, 0);      // P2' = P2 - V1 A21'
          // the pair's own pivot rows leave the panel (b's: here; a's rows of P1 and P2 are zeros in the LDS already)
#pragma unroll
          for (int g = 0; g < 4; ++g) p2[g] = retiring ? 0.0 : p2[g];
          double lt[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) lt[s] = -L21[(lk + 4 * s) * PLD + li];
          K5_SCHED_BARRIER();
          d4_t v2 = {zero, zero, zero, zero};
#pragma unroll
          for (int s = 0; s < 4; ++s) v2 = __builtin_amdgcn_mfma_f64_16x16x4f64(mb[s], p2[s], v2, 0, 0, 0);       // V2 = P2' S22^-1
          {
            const unsigned am16a = (ama >> ((R & 1) * 16)) & 0xffffu, am16b = (amb >> ((R & 1) * 16)) & 0xffffu;
            if ((am16a >> li) & 1u) *(d4_t *)(panel + (size_t)a * pstride + PIV + row * PIV + 4 * lk) = v1;
            if ((am16b >> li) & 1u) *(d4_t *)(panel + (size_t)(a + 1) * pstride + PIV + row * PIV + 4 * lk) = v2;
          }
          w1 = v1;
#pragma unroll
          for (int s = 0; s < 4; ++s) w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[s], v2[s], w1, 0, 0, 0);        // W1 = V1 - V2 L21
#pragma unroll
          for (int g = 0; g < 4; ++g) { w1[g] = retiring ? 0.0 : w1[g]; w2[g] = retiring ? 0.0 : v2[g]; }
          // right-hand sides of this tile's slots: the right-hand side is one more column of the matrix, its entries at the pair's
          // pivots are row F of the panels:  UF[row] -= W1[row] . P1[F] + W2[row] . P2[F]   (the pair's own slots retire)
          {
            double d = 0.0;
#pragma unroll
            for (int s = 0; s < 4; ++s) d = fma(w1[s], PA0[F * PLD + lk + 4 * s], d);
#pragma unroll
            for (int s = 0; s < 4; ++s) d = fma(w2[s], PA1[F * PLD + lk + 4 * s], d);
            d = rowsum4(d);
            if (lane < PIV) UF[row] = retiring ? 0.0 : UF[row] - d;
          }
        }
        K5S(8);
        if (has_next) {
          const unsigned wn = pmn[R >> 1];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            double *PN = PN0 + t * PSZ;
            const us4_t ct = ct_cur[t];
            const int st = psn[t * PIV + li];                       // slot of pivot li of that stage
            const double dgn = dgb[((j + 4) % 3) * 2 * PIV + t * PIV + li];
            d4_t acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int r = 16 * R + lk + 4 * g;
              acc[g] = PN[r * PLD + li] + A[ct[g]] + (r == st ? dgn : 0.0);
            }
            if (live && rowlive) {
              // a pivot that takes the slot of one of this pair's pivots enters with its own pair: zeros
              const bool fresh = (pmj[st >> 5] >> (st & 31)) & 1u;
              const int srow = fresh ? ZROW : st;
              double n1[4], n2[4];
#pragma unroll
              for (int s = 0; s < 4; ++s) { n1[s] = -PA0[srow * PLD + lk + 4 * s]; n2[s] = -PA1[srow * PLD + lk + 4 * s]; }
#pragma unroll
              for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w1[s], n1[s], acc, 0, 0, 0);
#pragma unroll
              for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[s], n2[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              PN[(16 * R + lk + 4 * g) * PLD + li] = acc[g];
              A[ct[g]] = 0.0;   // retired (the zero cell stays zero)
            }
            K5S(9 + t);
          }
          // operand A of the Schur update: the rows of the next pair's pivots leave it too
          const bool nextpiv = (wn >> ((R & 1) * 16 + li)) & 1u;
#pragma unroll
          for (int g = 0; g < 4; ++g) { nW1[g] = nextpiv ? 0.0 : -w1[g]; nW2[g] = nextpiv ? 0.0 : -w2[g]; }
        }
      }
      K5S(0);
      lds_barrier();
      K5S(1);
      // ---- phase 2:  U(R, C) -= W1 P1[C]' + W2 P2[C]',  operand B blanked on load (rows of this pair's and the next pair's pivots)
      if (live && has_next && rowlive) {
        double b1[2][4], b2[2][4];
        auto tile_loads = [&](int C, double (&x1)[4], double (&x2)[4]) __attribute__((always_inline)) {
          const unsigned g16 = ((pmj[C >> 1] | pmn[C >> 1]) >> ((C & 1) * 16)) & 0xffffu;
          const bool blank = (g16 >> li) & 1u;
          const int base = blank ? ZROW * PLD * 8 + lk * 8 : tile_lane + C * (16 * PLD * 8);
          const char *q1 = (const char *)PA0 + base, *q2 = (const char *)PA1 + base;
#pragma unroll
          for (int s = 0; s < 4; ++s) { x1[s] = *(const double *)(q1 + 32 * s); x2[s] = *(const double *)(q2 + 32 * s); }
        };
        tile_loads(0, b1[0], b2[0]);
#pragma unroll
        for (int C = 0; C < NT; ++C) {
          if (C <= R) {
            if (C + 1 < NT && C + 1 <= R) tile_loads(C + 1, b1[(C + 1) & 1], b2[(C + 1) & 1]);
            if ((pg >> C) & 1u) {
#pragma unroll
              for (int s = 0; s < 4; ++s) U[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(nW1[s], b1[C & 1][s], U[C], 0, 0, 0);
#pragma unroll
              for (int s = 0; s < 4; ++s) U[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(nW2[s], b2[C & 1][s], U[C], 0, 0, 0);
            }
          }
        }
      }
      K5S(2);
      lds_barrier();
      K5S(3);
      {
        const int q2 = min(j + 2, NP - 1), q1 = min(max(j + 1, 0), NP - 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          ct_nxt[t] = ctab4[((size_t)(2 * q2 + t) * NT + R) * 64 + lane];
          am_nxt[t] = P.amask[(2 * q1 + t) * 8 + (R >> 1)];
        }
        pg_nxt = P.pair_groups[q1];
      }
      // ---- phase 3: the columns (rows) of the pivots of pair j + 2 out of the tiles into the panels of pair j, which are dead;
      //      zeroed in place
      if (live && has_next2) {
        double *X = (double *)PA0;
        double *dummy = red + 2 * 16 * PIV + lane;
        // bit 4g of ge4 / gt4: row lk + 4g of a diagonal tile lies on or below / strictly below column li
        unsigned ge4 = 0u, gt4 = 0u;
#pragma unroll
        for (int g = 0; g < 4; ++g) { ge4 |= (lk + 4 * g >= li ? 1u : 0u) << (4 * g); gt4 |= (lk + 4 * g > li ? 1u : 0u) << (4 * g); }
        const unsigned rw2 = (pmn2[R >> 1] >> ((R & 1) * 16)) & 0xffffu;
        const us4_t jr4 = *(const us4_t *)(jm + 16 * R + 4 * lk);   // column offsets of the pivots in rows lk + 4 g of row tile R
#pragma unroll
        for (int C = 0; C < NT; ++C) {
          if (C <= R) {
            const unsigned cw2 = (pmn2[C >> 1] >> ((C & 1) * 16)) & 0xffffu;
            if ((cw2 | rw2) != 0u) {
              const int jc = jm[16 * C + (li & 3) * 4 + (li >> 2)];
              const unsigned cm = ((cw2 >> li) & 1u) ? (R > C ? 0x1111u : ge4) : 0u;
              const unsigned rmk = (rw2 >> lk) & (R > C ? 0x1111u : gt4);
              double *xr = X + (16 * R + lk) * PLD + jc, *xc = X + (16 * C + li) * PLD;
              if (cw2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) *(((cm >> (4 * g)) & 1u) ? xr + g * 4 * PLD : dummy) = U[C][g];
              }
              if (rw2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) *(((rmk >> (4 * g)) & 1u) ? xc + jr4[g] : dummy) = U[C][g];
              }
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int z = ~__builtin_amdgcn_sbfe((int)(cm | rmk), 4 * g, 1);
                U[C][g] = __hiloint2double(__double2hiint(U[C][g]) & z, __double2loint(U[C][g]) & z);
              }
            }
          }
        }
      }
      K5S(4);
      // the rest of the record's targets: every wave but the factor wave (service waves first, then the tile waves by row)
      if (has_next2 && R < NASMT) assemble_pair_tgt<QTOS_K5_ILPT>(A, sbuf, dbuf, n_tgt2, 1 << 30, (NSVC + R) * 64 + lane, (NSVC + NASMT) * 64);
      K5S(5);
      lds_barrier();
      K5S(6);
#pragma unroll
      for (int t = 0; t < 2; ++t) { ct_cur[t] = ct_nxt[t]; am_cur[t] = am_nxt[t]; }
      pg_cur = pg_nxt;
    }
  } else if (is_fac) {
    // ================================================ factor wave ===============================================
    for (int j = -2; j < NP; ++j) {
      K5_STEP_POINTERS
      K5S(0);
      lds_barrier();
      K5S(1);
      if (j >= -1 && has_next) {
        __builtin_amdgcn_s_setprio(3);
        const int c = 2 * (j + 1);
        double *Pc = PN0, *Pd = PN0 + PSZ;
        const int myps = psn[li], mypd = psn[PIV + li];
        // ---- LDL^T of A11 (split layout: lane (li, lk) holds row li, columns 4 g + lk; the entries above the diagonal are
        //      read from the mirrored position: the block that is factored is exactly symmetric) ----
        double a[4], wi[4], myinv;
        int pc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) pc[g] = psn[4 * g + lk];
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = 4 * g + lk > li ? Pc[pc[g] * PLD + li] : Pc[myps * PLD + 4 * g + lk];
        // A21 = rows of c's columns at d's pivots (operand B of the two products below): lane (li, lk): A21[li][lk + 4 s]
        double a21[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) a21[s] = Pc[mypd * PLD + lk + 4 * s];
        ldlt16s(a, wi, myinv, li, lk);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          LIN[li * PLD + 4 * g + lk] = wi[g];
          Pc[myps * PLD + 4 * g + lk] = 0.0;   // c's pivot rows leave both panels of the pair
          Pd[myps * PLD + 4 * g + lk] = 0.0;
        }
        if (lk == (li & 3)) DVN[li] = myinv;
        double zero = 0.0;
        asm volatile("" : "+v"(zero));
        d4_t mi = {zero, zero, zero, zero};
        {
          double lt[4], ld[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) { lt[s] = LIN[(lk + 4 * s) * PLD + li]; ld[s] = DVN[lk + 4 * s]; }
#pragma unroll
          for (int s = 0; s < 4; ++s) mi = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[s], lt[s] * ld[s], mi, 0, 0, 0);   // A11^-1 = L^-T D^-1 L^-1
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          MIVa[(lk + 4 * g) * PLD + li] = mi[g];
          minv_g[(size_t)c * (PIV * PIV) + (lk + 4 * g) * PIV + li] = mi[g];
        }
        // L21 = A21 A11^-1 (A11^-1 is symmetric: its accumulator layout is an A operand)
        d4_t l21 = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 4; ++s) l21 = __builtin_amdgcn_mfma_f64_16x16x4f64(mi[s], a21[s], l21, 0, 0, 0);   // l21[g] = L21[li][lk + 4 g]
#pragma unroll
        for (int g = 0; g < 4; ++g) L21[li * PLD + lk + 4 * g] = l21[g];
        // S22 = A22 - L21 A21'  (accumulator layout: lane (li, lk), g: row lk + 4 g, column li)
        d4_t s22;
        int pdg[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) pdg[g] = psn[PIV + lk + 4 * g];
#pragma unroll
        for (int g = 0; g < 4; ++g) s22[g] = li <= lk + 4 * g ? Pd[pdg[g] * PLD + li] : Pd[mypd * PLD + lk + 4 * g];
#pragma unroll
        for (int s = 0; s < 4; ++s) s22 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l21[s], a21[s], s22, 0, 0, 0);
        // through the scratch into the split layout, the lower triangle mirrored: exactly symmetric
#pragma unroll
        for (int g = 0; g < 4; ++g) LIN[(lk + 4 * g) * PLD + li] = s22[g];
#pragma unroll
        for (int g = 0; g < 4; ++g) Pd[mypd * PLD + lk + 4 * g] = 0.0;   // d's pivot rows leave its panel (their rows in c's panel are A21: kept)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = 4 * g + lk > li ? LIN[(4 * g + lk) * PLD + li] : LIN[li * PLD + 4 * g + lk];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        K5S(2);
        lds_barrier();   // ---- barrier 2 in the middle of the chain ----
        K5S(3);
        ldlt16s(a, wi, myinv, li, lk);
#pragma unroll
        for (int g = 0; g < 4; ++g) LIN[li * PLD + 4 * g + lk] = wi[g];
        if (lk == (li & 3)) DVN[li] = myinv;
        d4_t mi2 = {zero, zero, zero, zero};
        {
          double lt[4], ld[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) { lt[s] = LIN[(lk + 4 * s) * PLD + li]; ld[s] = DVN[lk + 4 * s]; }
#pragma unroll
          for (int s = 0; s < 4; ++s) mi2 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[s], lt[s] * ld[s], mi2, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          MIVb[(lk + 4 * g) * PLD + li] = mi2[g];
          minv_g[(size_t)(c + 1) * (PIV * PIV) + (lk + 4 * g) * PIV + li] = mi2[g];
        }
        __builtin_amdgcn_s_setprio(0);
      } else {
        K5S(2);
        lds_barrier();
        K5S(3);
      }
      K5S(4);
      lds_barrier();
      K5S(6);
    }
  } else {
    // =============================================== service waves ==============================================
    int rt_cur = 0, rt_nxt;   // right-hand-side cells of pair j + 1 (rtab), fetched a pair early
    double rhs_keep = 0.0;
    for (int j = -2; j < NP; ++j) {
      K5_STEP_POINTERS
      rt_nxt = P.rtab[2 * min(j + 2, NP - 1) * PIV + (lane & 31)];
      // LDS-DMA of the record of pair j + 2 (its predecessor was assembled in the previous step), 1 KB per wave instruction
      if (has_next2) {
        const int q = j + 2;
        int d0, d1, s0, s1;
        sload2(P.drec_off + q, d0, d1);
        sload2(P.srec_off + q, s0, s1);
        const int nbd = (d1 - d0) * 8, nbs = (s1 - s0) * 4;
        const char *gd = (const char *)(stream + d0), *gs = (const char *)(P.srec + s0);
        typedef __attribute__((address_space(3))) char lds_char;
        lds_char *ld = (lds_char *)dbuf, *ls = (lds_char *)sbuf;
        for (int c = sidx; c * 1024 < nbd; c += NSVC)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gd + min(c * 1024 + lane * 16, nbd - 16)), (__attribute__((address_space(3))) void *)(ld + c * 1024), 16, 0, 0);
        for (int c = sidx; c * 1024 < nbs; c += NSVC)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gs + min(c * 1024 + lane * 16, nbs - 16)), (__attribute__((address_space(3))) void *)(ls + c * 1024), 16, 0, 0);
      }
      if (sidx == 0 && j >= -1) {
        // ---- the right-hand-side row (row F of the panels) --------------------------------------------------------
        if (live) {
          const int a = 2 * j;
          const int sb_li = psj[PIV + li];
          double part = 0.0;
#pragma unroll
          for (int s = 0; s < 4; ++s) part = fma(MIVa[li * PLD + lk + 4 * s], PA0[F * PLD + lk + 4 * s], part);
          const double w1 = rowsum4(part);                 // w_a[li] on every lane
          if (lane < PIV) { panel[(size_t)a * pstride + lane] = w1; RSC[lane] = w1; }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          part = 0.0;
#pragma unroll
          for (int s = 0; s < 4; ++s) part = fma(PA0[sb_li * PLD + lk + 4 * s], RSC[lk + 4 * s], part);
          const double p2 = PA1[F * PLD + li] - rowsum4(part);   // right-hand side of b's pivots after a
          if (lane < PIV) RSC[PIV + lane] = p2;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          part = 0.0;
#pragma unroll
          for (int s = 0; s < 4; ++s) part = fma(MIVb[li * PLD + lk + 4 * s], RSC[PIV + lk + 4 * s], part);
          const double w2 = rowsum4(part);                 // w_b[li]
          if (lane < PIV) panel[(size_t)(a + 1) * pstride + lane] = w2;
        }
        // the assembled right-hand sides of the next pair's pivots leave their cells now (the cells are re-issued to the record
        // that is assembled in phases 2 and 3); they join the updated right-hand sides in phase 3
        if (has_next && lane < 2 * PIV) { rhs_keep = A[rt_cur]; A[rt_cur] = 0.0; }
      }
      if (sidx == NSVC - 1 && has_next2) publish_header(j + 2, true);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the record has landed
      K5S(0);
      lds_barrier();
      K5S(1);
      K5S(8);
      if (has_next2) {
        assemble_pair_eq(A, sbuf, dbuf, sidx * 64 + lane, NSVC * 64);
        K5S(9);
        assemble_pair_tgt<QTOS_K5_ILPS>(A, sbuf, dbuf, 0, n_tgt2, sidx * 64 + lane, NSVC * 64);
        K5S(10);
      }
      K5S(2);
      lds_barrier();
      K5S(3);
      if (sidx == 0 && j >= -1 && has_next && lane < 2 * PIV) {
        // right-hand-side row of the next pair's panels
        const int st = psn[lane];
        PN0[(lane >> 4) * PSZ + F * PLD + (lane & 15)] = rhs_keep + UF[st];
        UF[st] = 0.0;
      }
      if (has_next2) assemble_pair_tgt<QTOS_K5_ILPS>(A, sbuf, dbuf, n_tgt2, 1 << 30, sidx * 64 + lane, (NSVC + NASMT) * 64);
      K5S(5);
      lds_barrier();
      K5S(6);
      rt_cur = rt_nxt;
    }
  }
  // ---- backward substitution (sweep_backward: one barrier per stage, one-stage look-ahead) ---------------------------
  __syncthreads();   // drains the factor-panel stores: they are read back below
  {
    int *nxp = (int *)PAN;   // (the panels are dead)
    for (int i = tid; i < NS * 4; i += KT5) nxp[i] = P.nxt_pack[i];
    for (int i = tid; i < NS * 8; i += KT5) nxp[NS * 4 + i] = (int)P.amask2[i];
    for (int i = tid; i < FR; i += KT5) xs[i] = 0.0;
    __syncthreads();
    const SweepDs sd = {W.stream + (size_t)b * P.stream_len, W.ds + (size_t)b * P.n_cons, W.g + (size_t)b * P.n_cons, W.s + (size_t)b * P.n_cons, nullptr};
    sweep_backward<F, KT5, 9>(P, panel, dx, W.sol + (size_t)b * P.n_stages * PIV, xs, red, nxp, wv, lane, lds + ((LY::PAN + NS * 6 + 1) & ~1), sd);
  }
  K5S(7);
#undef K5S
#undef K5_STEP_POINTERS
}

}  // namespace qtos
