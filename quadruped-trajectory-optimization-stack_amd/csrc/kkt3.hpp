// kkt3.hpp -- k_kkt3: the factor + solve kernel for fronts of up to 128 slots without continuation records (the default up to
// 112 slots; larger fronts and records with continuation parts stay with k_kkt2).  Same chain of fronts, cells, records,
// panels, sweeps and k_chord as kkt2.hpp -- and the same arithmetic: bit-identical plans.  What differs:
//
//   * The assembly of record k+2 starts in phase AB, on the waves that have no job there (their cells are not the ones the
//     tile waves read in that phase: those belong to stage k+1's columns, and a retired cell is re-issued two stages later):
//     equality entries and the first QTOS_AB_ROUNDS targets per thread of the gather table; the rest on every wave but the
//     factor wave at the end of phase C.  -5 % per launch on 112 slots, -6 % on 96; +4 % on 128, where seven idle waves are
//     too few (profiles/r04_experiments).
//   * No special prologue: the stage loop starts two stages early (k = -2, -1 build the panels of stages 0 and 1 through the
//     same path as every other stage).
//
// The template keeps its second parameter (the kernel's name in every committed profile is k_kkt3<F, 1>): MODE 0 of rounds
// 4 - 5 -- the inequality blocks G' S G condensed on the matrix core, correct and slower -- lives in
// scratch/experiments/kkt3_mode0.hpp.
#pragma once
#include "kkt2.hpp"

namespace qtos {

inline size_t kkt3_lds_bytes(int F, int NS, int max_srec, int max_drec, int n_cells) { return kkt2_lds_bytes(F, NS, max_srec, max_drec, n_cells); }

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
// The records and the gather-table assembly of k_kkt2 (assemble_stage), its first QTOS_AB_ROUNDS targets per thread moved into
// phase AB onto the waves that have no job there (with the equality entries), the rest on every wave but the factor wave at the
// end of phase C.
template <int F, int MODE>
__global__ __launch_bounds__(KT2) void k_kkt3(DevPlan P, DevWork W, int B) {
  static_assert(F <= 128 && F % 16 == 0, "k_kkt3: fronts of up to 128 slots (waves 9 .. 15 must be free in phase AB)");
  static_assert(MODE == 1, "k_kkt3: MODE 0 left the library (scratch/experiments/kkt3_mode0.hpp)");
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
  // targets of the gather table assembled in phase AB by the waves that idle there.  Fronts of up to 96 slots: none -- with the
  // records of round 5 (reduced swings) and six tile waves the idle waves' assembly lengthens phase AB by more than it takes off
  // phase C: -0.6 % (trot) .. -1.5 % (walk, knots200, reference_compat, exp_5, mixed) per launch without it (round 6,
  // profiles/r06_experiments/kkt96_tuning.log); the equality entries stay in phase AB.  112 and 128 slots: as measured in round 4.
  constexpr int AB_R = F <= 96 ? 0 : QTOS_AB_ROUNDS;
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
      //      again two stages later (Symbolic::compact_cells).
      assemble_eq(A, sbuf, dbuf, (wv - NT - 1) * 64 + lane, (15 - NT) * 64);
      assemble_targets(A, sbuf, dbuf, 0, AB_R * (15 - NT) * 64, (wv - NT - 1) * 64 + lane, (15 - NT) * 64);
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
    {
      // the targets phase AB left: every wave but the factor wave (the waves without Schur tiles first: low item indices)
      const int apos = is_upd ? (15 - NU) + uw : uw - NU;
      if (wv >= 1 && k + 2 < NS) assemble_targets(A, sbuf, dbuf, AB_R * (15 - NT) * 64, 1 << 30, apos * 64 + lane, 15 * 64);
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
