// kkt3.hpp -- k_kkt3: the factor + solve kernel of round 4 for fronts of up to 128 slots (larger fronts and records with
// continuation parts stay with k_kkt2).  Same chain of fronts, cells, panels, sweeps and k_chord as kkt2.hpp; what changed:
//
//   * The inequality blocks J' S J are condensed ON THE MATRIX CORE by one wave.  k_kkt2 summed them entry by entry into the
//     cells through a gather table -- one thread per target, three barrier weights, six Jacobian values and a decoded
//     contribution word per term: a third of the kernel's vector instructions, spread over fifteen waves.  Here wave 15, which
//     has no job in phase AB, takes a record's blocks one after the other in that phase: G' S G of a block of
//     up to 32 columns is three 16 x 16 tiles (one for up to 16 columns), one f64 matrix instruction per four rows each; a
//     lane then holds four entries of the lower triangle and adds each to its cell with ds_add_f64 -- the cells come from a
//     table in the record (Symbolic::emit_iq_section), entries above the diagonal go to a trash cell.  One wave, LDS
//     operations in program order: the summation order of a cell is fixed.  Right-hand sides -G' w by a row sum across the
//     wave's four row groups.  (Round 2 tried the products with read-modify-write cells: three to four LDS round trips per
//     block in sequence; the atomic add has none.  Experiment A1 of this round condensed into the Schur tiles instead: 48
//     products per stage, slower -- profiles/r04_experiments.)
//   * No special prologue: the stage loop starts two stages early (k = -2, -1 build the panels of stages 0 and 1 through the
//     same path as every other stage).
#pragma once
#include "kkt2.hpp"

namespace qtos {

inline size_t kkt3_lds_bytes(int F, int NS, int max_srec, int max_drec, int n_cells) { return kkt2_lds_bytes(F, NS, max_srec, max_drec, n_cells); }

// The inequality blocks of one record, by ONE wave (section layout: Symbolic::emit_iq_section).  abase = LDS byte address of
// the cells.
// The inequality blocks and static entries of one record (section layout: Symbolic::emit_iq_section), shared by THREE waves:
// wave TY takes the 16 x 16 tiles of type TY of every block -- (0,0): class-0 columns x class-0 columns, (1,0): class 1 x
// class 0, (2 = (1,1)): class 1 x class 1 -- and the static entries of that type.  A pair of variables has the same type in
// every block (Symbolic::color_columns), so the three waves' cells never meet and each wave's adds (LDS atomics: no value
// comes back, nothing waits for them) are executed in its own program order: the summation order of a cell is fixed.
// abase = LDS byte address of the cells.  Two-stage software pipeline over the blocks: tables of block b + 2 and Jacobian
// values of block b + 1 are in flight while block b is multiplied.
struct IqS1 { int colt, rct, ta, tb; };
struct IqS2 { IqS1 s; double gA, gB, sg, ww, hA, hB, sh, wh; };
template <int TY>
__device__ __forceinline__ void condense_type(unsigned abase, const int *sbuf, const double *dbuf, int lane, int part, unsigned long long *stp = nullptr) {
#ifdef QTOS_STAMPS
  unsigned long long tq_ = 0;
#define IQ_ST0() do { if (lane == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq_) :: "memory"); } while (0)
#define IQ_ST(i) do { if (lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stp[i] += t_ - tq_; tq_ = t_; } } while (0)
#else
#define IQ_ST0() do {} while (0)
#define IQ_ST(i) do {} while (0)
#endif
  IQ_ST0();
  const int li = lane & 15, lk = lane >> 4;
  const int *q = sbuf + sbuf[2];
  const int hv = q[8 + lane];        // the headers of up to 16 blocks, one int per lane
  // part 0 (phase AB): the static entries and the first QTOS_IQ_SPLIT-th of the blocks; part 1 (phase C, another wave: the
  // barrier between the phases orders the two): the rest
#ifndef QTOS_IQ_SPLIT
#define QTOS_IQ_SPLIT 2
#endif
  const int nb_all = __builtin_amdgcn_readfirstlane(q[0]), nb_ab = (nb_all + QTOS_IQ_SPLIT - 1) / QTOS_IQ_SPLIT;
  const int b_first = part == 0 ? 0 : nb_ab, nb = part == 0 ? nb_ab : nb_all;
  const int nst = part == 0 ? __builtin_amdgcn_readfirstlane(q[1 + TY]) : 0, sto = __builtin_amdgcn_readfirstlane(q[4 + TY]);
  for (int i = lane; i < nst; i += 64) {
    const int v = q[sto + i];
    const unsigned a = abase + 8u * (unsigned)(v >> 12);
    const double x = dbuf[v & 4095];
    asm volatile("ds_add_f64 %0, %1" :: "v"(a), "v"(x));
  }
  IQ_ST(9);
  auto present = [&](int b) { return b < nb && ((__builtin_amdgcn_readlane(hv, 4 * b + 1) >> (16 + TY)) & 1); };
  auto stage1 = [&](int b, IqS1 &o) __attribute__((always_inline)) {
    o.colt = 0xffff; o.rct = 0; o.ta = 0; o.tb = 0;
    if (!present(b)) return;
    const int pres = (__builtin_amdgcn_readlane(hv, 4 * b + 1) >> 16) & 7;
    const int *d = q + __builtin_amdgcn_readlane(hv, 4 * b + 2);
    typedef int i2_t __attribute__((ext_vector_type(2)));
    const i2_t t = ((const i2_t *)(d + 32 + 128 * __builtin_popcount(pres & ((1 << TY) - 1))))[lane];
    o.colt = d[li]; o.rct = d[16 + li]; o.ta = t[0]; o.tb = t[1];
  };
  auto stage2 = [&](int b, const IqS1 &s1, IqS2 &o) __attribute__((always_inline)) {
    o.s = s1;
    o.gA = o.gB = o.sg = o.ww = o.hA = o.hB = o.sh = o.wh = 0.0;
    if (!present(b)) return;
    const int mn = __builtin_amdgcn_readlane(hv, 4 * b + 1), m = mn & 255, n = (mn >> 8) & 255;
    const double *G = dbuf + __builtin_amdgcn_readlane(hv, 4 * b);
    const int a0 = s1.colt & 255, a1 = (s1.colt >> 8) & 255;
    const int cA = TY == 0 ? a0 : a1, cB = TY == 2 ? a1 : a0;
    const bool rv = lk < m, rv2 = 4 + lk < m;
    if (rv && cA != 255) o.gA = G[lk * n + cA];
    if (rv2 && cA != 255) o.hA = G[(4 + lk) * n + cA];       // rows 4 .. 7 (friction pyramids: five rows)
    if (TY == 1) {
      if (rv && cB != 255) o.gB = G[lk * n + cB];
      if (rv2 && cB != 255) o.hB = G[(4 + lk) * n + cB];
    }
    if (rv) { o.sg = G[m * n + lk]; o.ww = G[m * n + m + lk]; }
    if (rv2) { o.sh = G[m * n + 4 + lk]; o.wh = G[m * n + m + 4 + lk]; }
  };
  auto multiply_add = [&](int b, const IqS2 &o) __attribute__((always_inline)) {
    if (!present(b)) return;
    const int m = __builtin_amdgcn_readlane(hv, 4 * b + 1) & 255;
    const double gB = TY == 1 ? o.gB : o.gA, hB = TY == 1 ? o.hB : o.hA;
    double zero = 0.0;
    asm volatile("" : "+v"(zero));
    d4_t D = {zero, zero, zero, zero};
    D = __builtin_amdgcn_mfma_f64_16x16x4f64(o.sg * o.gA, gB, D, 0, 0, 0);
    if (m > 4) D = __builtin_amdgcn_mfma_f64_16x16x4f64(o.sh * o.hA, hB, D, 0, 0, 0);
    const unsigned c0 = abase + 8u * ((unsigned)o.s.ta & 0xffffu), c1 = abase + 8u * ((unsigned)o.s.ta >> 16);
    const unsigned c2 = abase + 8u * ((unsigned)o.s.tb & 0xffffu), c3 = abase + 8u * ((unsigned)o.s.tb >> 16);
    asm volatile("ds_add_f64 %0, %1" :: "v"(c0), "v"(D[0]));
    asm volatile("ds_add_f64 %0, %1" :: "v"(c1), "v"(D[1]));
    asm volatile("ds_add_f64 %0, %1" :: "v"(c2), "v"(D[2]));
    asm volatile("ds_add_f64 %0, %1" :: "v"(c3), "v"(D[3]));
    if (TY != 1) {
      // right-hand sides of the class's columns: -sum_r G[r][c] w_r, column c on lane li of every row group
      double r = o.gA * o.ww;
      if (m > 4) r = fma(o.hA, o.wh, r);
      r = -rowsum4(r);
      if (lk == 0) {
        const unsigned ar = abase + 8u * (TY == 0 ? (unsigned)o.s.rct & 0xffffu : (unsigned)o.s.rct >> 16);
        asm volatile("ds_add_f64 %0, %1" :: "v"(ar), "v"(r));
      }
    }
  };
  if (nb > b_first) {
    IqS1 s1a, s1b;
    IqS2 s2;
    stage1(b_first, s1a);
    stage1(b_first + 1, s1b);
    stage2(b_first, s1a, s2);
    for (int b = b_first; b < nb; ++b) {
      IqS1 s1c;
      IqS2 s2n;
      stage1(b + 2, s1c);
      stage2(b + 1, s1b, s2n);
      multiply_add(b, s2);
      s2 = s2n; s1b = s1c;
    }
  }
  IQ_ST(10);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the cells are read and written by ordinary accesses from here on
  IQ_ST(11);
}

// The rest of a record: equality Jacobian entries and multiplier right-hand sides are distinct cells of their own (any thread).
__device__ __forceinline__ void assemble_eq(double *A, const int *sbuf, const double *dbuf, int t0, int nth) {
  const int n_ent = sbuf[0], n_rhs = sbuf[1];
  const int *eidx = sbuf + SHDR + PIV;
  const double *eval = dbuf + PIV;
  for (int i = t0; i < n_ent; i += nth) A[eidx[i]] += eval[i];
  const int *rsl = eidx + n_ent;
  const double *rval = eval + n_ent;
  for (int i = t0; i < n_rhs; i += nth) A[rsl[i]] += rval[i];
}
// targets [t_begin, t_end) of a record's gather table (kernels.hpp assemble_stage: one thread per target, fixed summation order)
__device__ __forceinline__ void assemble_targets(double *A, const int *sbuf, const double *dbuf, int t_begin, int t_end, int t0, int nth) {
  const int n_tgt = sbuf[5];
  const int *tg = sbuf + sbuf[4];                     // n_tgt + 1 ints: (cell << 12) | first contribution
  const int *cl = tg + n_tgt + 1;                     // one self-contained int per contribution
  for (int t = t_begin + t0; t < min(t_end, n_tgt); t += nth) {
    const int tv = tg[t], c0 = tv & 4095, c1 = tg[t + 1] & 4095;
    const double a_old = A[tv >> 12];
    double acc = 0;
    for (int j = c0; j < c1; ++j) acc += gather_term(dbuf, cl[j]);
    A[tv >> 12] = a_old + acc;
  }
}
#ifndef QTOS_AB_ROUNDS
#define QTOS_AB_ROUNDS 64
#endif
// MODE 0: the inequality blocks condensed by matrix instructions (records of Symbolic::iq_mfma); MODE 1: the records and the
// gather-table assembly of k_kkt2 (assemble_stage), its first QTOS_AB_ROUNDS targets per thread moved into phase AB onto the
// waves that have no job there (with the equality entries), the rest on every wave but the factor wave at the end of phase C.
template <int F, int MODE>
__global__ __launch_bounds__(KT2) void k_kkt3(DevPlan P, DevWork W, int B) {
  static_assert(F <= 128 && F % 16 == 0, "k_kkt3: fronts of up to 128 slots (waves 9 .. 15 must be free in phase AB)");
  const int b = blockIdx.x;
  if (b >= B || W.done[b] || W.chord[b] == 1) return;   // (a problem flagged for a chord step is k_chord's)
  extern __shared__ double lds[];
  using CF = Kkt2Cfg<F>;
  using LY = Kkt2Layout<F>;
#ifndef QTOS_NU3
#define QTOS_NU3 0
#endif
  // (MODE 1: the number of update waves may be overridden for experiments -- waves 4, 8, 12 join as update indices 12 .. 14)
  constexpr int NT = CF::NT, NU = (MODE == 1 && QTOS_NU3 > 0 && F == 128) ? QTOS_NU3 : CF::NU, MAXT2 = (CF::NTILE + NU - 1) / NU, FR = CF::FR, PSZ = LY::PSZ;
  const int tid = threadIdx.x, NS = P.n_stages, n = P.n_sol;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
  double *Lib = lds + LY::LIB, *dvb = lds + LY::DVB, *dgb = lds + LY::DGB, *UF = lds + LY::UF, *xs = lds + LY::XS;
  double *red = lds + LY::RED, *PB = lds + LY::PB;
  int *psb = (int *)(lds + LY::PSB), *hib = (int *)(lds + LY::HIB), *jm = (int *)(lds + LY::JM);
  unsigned *pm = (unsigned *)(lds + LY::PM);
  unsigned char *jmb = (unsigned char *)jm;   // slot -> pivot index of stages k+2 / k+3 (layout: kkt2.hpp)
  double *Minv = lds + LY::MIV;
  // record buffers: [dbuf 0][dbuf 1][sbuf 0][sbuf 1], record s lives in buffer s & 1, filled by LDS-DMA
  const int dstride = (int)kkt2_dbuf_doubles(F, P.max_drec), sstride = (int)kkt2_sbuf_ints(F, P.max_srec);
  double *const dbuf0 = lds + LY::VAR;
  int *const sbuf0 = (int *)(dbuf0 + 2 * dstride);
  int *hiall = sbuf0 + 2 * sstride;
  double *A = (double *)(hiall + ((NS + 4) & ~3));   // cells of the assembled entries
  const double *stream = W.stream + (size_t)b * P.stream_len;
  double *panel = W.panel + (size_t)b * P.panel_stride;
  double *dx = W.dx + (size_t)b * n;
  const int pstride = (F + 1) * PIV;   // per stage: w (16), V (F x 16)
  auto r3 = [](int s) { return (s + 3) % 3; };   // ring slots of the stages -2 ..

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
  for (int i = tid; i < 3 * PSZ + F * PLD; i += KT2) PB[i] = 0.0;   // (the three panels and the blanked copy behind them)
  for (int i = tid; i < PIV * PLD; i += KT2) Minv[i] = 0.0;
  for (int i = tid; i < FR; i += KT2) { UF[i] = 0.0; xs[i] = 0.0; }
  if (tid < 16) pm[tid] = 0u;
  for (int v = tid; v < n; v += KT2) dx[v] = 0.0;
  // record 0 and its header; everything else of the start-up is the stage loop itself, from k = -2
  {
    const int s0 = P.srec_off[0], s1 = P.srec_off[1], d0 = P.drec_off[0], d1 = P.drec_off[1];
    for (int i = tid; i < s1 - s0; i += KT2) sbuf0[i] = P.srec[s0 + i];
    for (int i = tid; i < d1 - d0; i += KT2) dbuf0[i] = stream[d0 + i];
  }
  __syncthreads();
  if (tid < PIV) {
    const int slot = sbuf0[SHDR + tid];
    psb[tid] = slot;
    jmb[(slot & ~15) + (slot & 3) * 4 + ((slot >> 2) & 3)] = (unsigned char)tid;
    dgb[tid] = dbuf0[tid];
    atomicOr(&pm[slot >> 5], 1u << (slot & 31));
  }
  if (tid == 0) { hib[0] = sbuf0[3]; hiall[0] = (sbuf0[3] + 15) & ~15; }
  __syncthreads();

  // wave 0: LDL^T + L^-1 + (L D L^T)^-1 of the pivot block of the panel Pn, then the pivot rows leave the panel
  double *minv_g = W.minv + (size_t)b * NS * (PIV * PIV);   // inverse of every pivot block, kept for chord steps
  auto factor_block = [&](double *Pn, const int myps, const int *ps, double *Lin, double *dvn, int ks) __attribute__((always_inline)) {
    // split layout (ldlt16s): lane (li, lk) holds row li of the block, columns c = 4 g + lk; entries above the diagonal are
    // read from the mirrored position: the block that is factored is exactly symmetric
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

#ifdef QTOS_STAMPS
  // diagnostic build: per wave, cycles spent in each part of a stage (accumulated in LDS by lane 0)
  __shared__ unsigned long long st2[16][12];
  unsigned long long ts_ = 0;
  if (tid < 192) st2[tid / 12][tid % 12] = 0;
  __syncthreads();
  KS2_START();
#endif
  int prow_next = 0;   // pivot slot li of stage k+1
  typedef unsigned short us4_t __attribute__((ext_vector_type(4)));
  const us4_t *ctab4 = (const us4_t *)P.ctab;
  us4_t ct_cur = {0, 0, 0, 0};
  int rc_cur = 0;      // cell of the assembled rhs of pivot li of stage k+1
  const int tid_outer = tid, lane_outer = lane;
  for (int k = -2; k < NS; ++k) {
    int tid = tid_outer, lane = lane_outer;
    asm volatile("" : "+v"(tid), "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    const int pb = (k & 1) ? 2 : 0;
    double *Pk = PB + pb * PSZ, *Yk = PB + PSZ, *Xn = PB + (2 - pb) * PSZ;
    const bool has_next = k + 1 >= 0 && k + 1 < NS;
    KS2(7);
    // the records of stage k+2 (landed during the previous stage; record 0: start-up above)
    double *dbuf = dbuf0 + (k & 1) * dstride;
    int *sbuf = sbuf0 + (k & 1) * sstride;
    const us4_t ct_nxt = ctab4[((size_t)min(k + 2, NS - 1) * NT + min(wv, NT - 1)) * 64 + lane];
    const int rc_nxt = P.rtab[min(k + 2, NS - 1) * PIV + li];
    // ---- AB(k) ----------------------------------------------------------------------------------------
    const Mask256 m1 = load_mask8(pm + ((k + 1) & 1) * 8, lane);   // pivot slots of stage k+1
    if (wv < NT) {
     if (k >= -1) {
      const int R = wv;
      const unsigned am_word = k >= 0 ? P.amask[k * 8 + (R >> 1)] : 0u;
      const int prow = has_next ? prow_next : 0;
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
      const double dgn = dgb[r3(k + 1) * PIV + li];
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
      {
        // operands of the Schur update: -V (accumulator layout -> row-major) and the raw rows, next pivots' rows blanked
        const bool myrowpiv = has_next && ((grp16(m1, R) >> li) & 1u);
        double *Pm = lds + LY::PMB;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Yk[(16 * R + li) * PLD + lk + 4 * g] = myrowpiv ? 0.0 : -vt[g];
          Pm[(16 * R + li) * PLD + lk + 4 * g] = myrowpiv ? 0.0 : pr[g];
        }
      }
      if (has_next) {
#pragma unroll
        for (int g = 0; g < 4; ++g) Xn[(16 * R + lk + 4 * g) * PLD + li] = acc[g];
      }
      const unsigned am16 = (am_word >> ((R & 1) * 16)) & 0xffffu;
      if ((am16 >> li) & 1u) {   // (no panel of a stage before the first: am_word = 0)
        double *pv = panel + (size_t)k * pstride + PIV;
        *(d4_t *)(pv + (16 * R + li) * PIV + 4 * lk) = vt;
      }
     }
    } else if (wv == NT) {
     if (k >= -1) {
      // right-hand-side row: w = (L D L^T)^-1 p_F; rhs -= P w; right-hand side of the next pivots
      double part = 0.0;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) part = fma(Minv[li * PLD + lk + 4 * s4], Pk[F * PLD + lk + 4 * s4], part);
      part = rowsum4(part);                    // w[li] on every lane
      if (lane < PIV && k >= 0) panel[(size_t)k * pstride + lane] = part;
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
    }
    else if (k + 2 < NS) {
      // ---- the waves without a job in this phase assemble record k+2 into the cells.  None of the cells they touch is read
      //      or retired by the tile waves here: those belong to the columns of stage k+1, and a retired cell is handed out
      //      again two stages later (Symbolic::compact_cells).  Waves 13 .. 15: the inequality blocks and static entries, one
      //      tile type each (condense_type); the others: equality entries and multiplier right-hand sides.
      if constexpr (MODE == 1) {
        assemble_eq(A, sbuf, dbuf, (wv - NT - 1) * 64 + lane, (15 - NT) * 64);
        assemble_targets(A, sbuf, dbuf, 0, QTOS_AB_ROUNDS * (15 - NT) * 64, (wv - NT - 1) * 64 + lane, (15 - NT) * 64);
      } else {
      typedef __attribute__((address_space(3))) double lds_double;
      const unsigned abase = (unsigned)(size_t)(lds_double *)A;
#if !(defined(QTOS_IQ_ABL) && (QTOS_IQ_ABL & 4))
#ifdef QTOS_STAMPS
      unsigned long long *stp = &st2[wv][0];
#else
      unsigned long long *stp = nullptr;
#endif
      if (wv == 13) condense_type<0>(abase, sbuf, dbuf, lane, 0, stp);
      else if (wv == 14) condense_type<1>(abase, sbuf, dbuf, lane, 0, stp);
      else if (wv == 15) condense_type<2>(abase, sbuf, dbuf, lane, 0, stp);
      else
#endif
        assemble_eq(A, sbuf, dbuf, (wv - NT - 1) * 64 + lane, (12 - NT) * 64);
      }
    }
    KS2(0);
    lds_barrier();
    KS2(1);
    // ---- C(k) ---------------------------------------------------------------------------------------
    KS2(8);
    if (wv == 0) {
      __builtin_amdgcn_s_setprio(3);
      if (has_next) factor_block(Xn, prow_next, psb + r3(k + 1) * PIV, Lib + ((k + 1) & 1) * PIV * PLD, dvb + ((k + 1) & 1) * PIV, k + 1);
      __builtin_amdgcn_s_setprio(0);
    } else if (is_upd) {
      const Mask256 m2 = load_mask8(pm + (k & 1) * 8, lane);   // pivot slots of stage k+2
      const bool extract = k + 2 < NS;
      const unsigned char *jm2 = jmb + (k & 1) * FR;
      double *Xnn = Pk;   // the panel of stage k is dead: it receives the columns of stage k+2
      const double *Bop = lds + LY::PMB;
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
      // the products (the extraction behind them starts with no LDS round trip of its own)
      int jc8[MAXT2], jr32[MAXT2];
#pragma unroll
      for (int t = 0; t < MAXT2; ++t) {
        const int rc = rcs[t] < 0 ? 0 : rcs[t];
        jc8[t] = jm2[16 * (rc & 255) + (li & 3) * 4 + (li >> 2)];
        jr32[t] = *(const int *)(jm2 + 16 * (rc >> 8) + 4 * lk);
      }
      if (k >= 0) {
        tile_loads(rcs[0], wa[0], pbv[0]);
#pragma unroll
        for (int t = 0; t < MAXT2; ++t) {
          if (t + 1 < MAXT2) tile_loads(rcs[t + 1], wa[(t + 1) & 1], pbv[(t + 1) & 1]);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            U[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[t & 1][s4], pbv[t & 1][s4], U[t], 0, 0, 0);
        }
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
    if constexpr (MODE == 1) {
      // the targets phase AB left: every wave but the factor wave (the waves without Schur tiles first: low item indices)
      const int apos = is_upd ? (15 - NU) + uw : uw - NU;
      if (wv >= 1 && k + 2 < NS) assemble_targets(A, sbuf, dbuf, QTOS_AB_ROUNDS * (15 - NT) * 64, 1 << 30, apos * 64 + lane, 15 * 64);
    }
    // LDS-DMA of the records of stage k+3 into the other buffer by waves 8 and 12 (1 KB per instruction, chunk c of a record
    // by wave c mod 2; wave 12 takes the chunks with the header it publishes below)
    if ((wv == 8 || wv == 12) && k + 3 < NS) {
      const int s = k + 3, wi = wv == 12 ? 0 : 1;
      int d0, d1, s0, s1;
      sload2(P.drec_off + s, d0, d1);
      sload2(P.srec_off + s, s0, s1);
      const int nbd = (d1 - d0) * 8, nbs = (s1 - s0) * 4;
      const char *gd = (const char *)(stream + d0), *gs = (const char *)(P.srec + s0);
      typedef __attribute__((address_space(3))) char lds_char;
      lds_char *ld = (lds_char *)(dbuf0 + (s & 1) * dstride), *ls = (lds_char *)(sbuf0 + (s & 1) * sstride);
      for (int c = wi; c * 1024 < nbd; c += 2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gd + min(c * 1024 + lane * 16, nbd - 16)), (__attribute__((address_space(3))) void *)(ld + c * 1024), 16, 0, 0);
      for (int c = wi; c * 1024 < nbs; c += 2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gs + min(c * 1024 + lane * 16, nbs - 16)), (__attribute__((address_space(3))) void *)(ls + c * 1024), 16, 0, 0);
    }
    // the second part of the inequality blocks of record k+2, one tile type per wave (condense_type), while the records travel
    if (MODE == 0 && !(wv & 3) && wv != 0 && k + 2 < NS) {
      typedef __attribute__((address_space(3))) double lds_double;
      const unsigned abase = (unsigned)(size_t)(lds_double *)A;
#ifdef QTOS_STAMPS
      unsigned long long *stp = &st2[wv][0];
#else
      unsigned long long *stp = nullptr;
#endif
#if !(defined(QTOS_IQ_ABL) && (QTOS_IQ_ABL & 4))
      if (wv == 4) condense_type<0>(abase, sbuf, dbuf, lane, 1, stp);
      else if (wv == 8) condense_type<1>(abase, sbuf, dbuf, lane, 1, stp);
      else condense_type<2>(abase, sbuf, dbuf, lane, 1, stp);
#endif
    }
    if (wv == 8 || wv == 12) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wv == 12) {
      // header of stage k+3, published to the LDS rings (their slots have no reader left in this phase: stage k's pivot
      // slots / diagonals, stage k+1's slot map and mask) from the first chunks of the record, which this wave has just waited for
      const int hs = k + 3;
      if (hs < NS) {
        if (lane < 8) pm[(hs & 1) * 8 + lane] = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int *hb = sbuf0 + (hs & 1) * sstride;
        const double *hdb = dbuf0 + (hs & 1) * dstride;
        if (lane == 0) { const int hv = hb[3]; hib[r3(hs)] = hv; hiall[hs] = (hv + 15) & ~15; }
        if (lane < PIV) {
          const int hv = hb[SHDR + lane];
          dgb[r3(hs) * PIV + lane] = hdb[lane];
          psb[r3(hs) * PIV + lane] = hv;
          jmb[(hs & 1) * FR + (hv & ~15) + (hv & 3) * 4 + ((hv >> 2) & 3)] = (unsigned char)lane;
          atomicOr(&pm[(hs & 1) * 8 + (hv >> 5)], 1u << (hv & 31));
        }
      }
    }
    KS2(3);
    lds_barrier();
    KS2(4);
    if (k + 2 < NS) prow_next = psb[r3(k + 2) * PIV + li];
    ct_cur = ct_nxt;
    rc_cur = rc_nxt;
  }
  // ---- backward substitution (sweep_backward: one barrier per stage, one-stage look-ahead) -----------------------
  __syncthreads();  // drains the factor-panel stores: they are read back below
  KS2(7);
  {
    int *nxp = (int *)PB;   // (the panels are dead)
    for (int i = tid; i < NS * 4; i += KT2) nxp[i] = P.nxt_pack[i];
    for (int i = tid; i < NS * 8; i += KT2) nxp[NS * 4 + i] = (int)P.amask2[i];
    __syncthreads();
    const SweepDs sd = {W.stream + (size_t)b * P.stream_len, W.ds + (size_t)b * P.n_cons, W.g + (size_t)b * P.n_cons, W.s + (size_t)b * P.n_cons};
    sweep_backward<F>(P, panel, dx, W.sol + (size_t)b * P.n_stages * PIV, xs, red, nxp, wv, lane, lds + ((LY::PB + NS * 6 + 1) & ~1), sd);
  }
#ifdef QTOS_STAMPS
  KS2(6);
  __syncthreads();
  if (tid < 192 && W.trace) W.trace[((size_t)b * (P.max_iter + 1) + 16) * 4 + tid] = (double)st2[tid / 12][tid % 12];
#endif
}

}  // namespace qtos
