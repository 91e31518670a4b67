// kkt4.hpp -- k_kkt4: the stage of k_kkt2 re-scheduled so that the panel chain and the Schur updates run SIDE BY SIDE
// (fronts of up to 128 slots, records without continuation parts; same records, cells, panels in HBM, sweeps and k_chord).
//
// k_kkt2's stage is a strict alternation: AB(k) builds the panel of stage k+1 from columns that phase C(k-1) extracted from
// the Schur tiles AFTER update k-1, and C(k) needs the V of AB(k) for update k -- update -> extraction -> panel -> update is
// one dependent cycle, its two halves run on disjoint sets of waves, and half the workgroup waits in either phase (stamps:
// profiles/r04_experiments).  The factor wave is not on that cycle, so signalling it ahead with flags gains nothing.  What
// breaks the cycle is extracting a stage's columns one update EARLIER and applying the missing update to the panel itself:
//
//     P_{k+1} = [ cells + U_{<= k-2}[:, piv_{k+1}] + diag ]  -  V_{k-1} P_{k-1}[piv_{k+1}]'  -  V_k P_k[piv_{k+1}]'
//
// (k_kkt2: U_{<= k-1} and the last term only).  Then update k-1 and the extraction of stage k+2's columns have the whole
// of stage k to themselves, on waves of their own:
//
//   waves 0 .. NT-1    tile waves: V_k, P_{k+1} (12 chained matrix instructions instead of 8), -V_k as operand A of update k
//                      (transposed, rows of the pivots of stages k+1 and k+2 blanked: both sets have left the Schur tiles when
//                      that update is applied); wave 0 then factors the pivot block of stage k+1, the others take shares of
//                      the assembly of record k+2
//   wave NT            right-hand-side row (as k_kkt2), then the header of stage k+3 (pivot slots from global memory: the
//                      records are single-buffered and travel late)
//   waves NT+1 .. 15   update waves, 36 tiles on 7 waves at 128 slots: LDS-DMA of record k+2 (one buffer: its predecessor was
//                      assembled in the previous stage), update k-1 -- operand B is the panel P_{k-1} itself, blanked on load --,
//                      extraction of the columns of stage k+2, their share of the assembly
//
// Two barriers per stage as before, but the chain between them is tile phase + factorisation; the update waves cross both
// with work in hand.  Role-specialised loops (tile / factor waves and update waves run different code with the same barrier
// count): the register budget of a wave is its own role's.  Four panel buffers (P_{k-1} .. P_{k+2}) and two operand buffers
// instead of three and two -- the LDS comes from single-buffering the records.
#pragma once
#include "kkt3.hpp"

namespace qtos {

template <int F>
struct Kkt4Layout {
  static constexpr int NT = F / 16, FR = (F + 63) & ~63, PSZ = (F + 1) * PLD, YSZ = F * PIV;
  static constexpr int LIB = 0;                          // 16 x PLD      L^-1 of the block being factored
  static constexpr int DVB = LIB + PIV * PLD;            // 16            1 / d
  static constexpr int DGB = DVB + PIV;                  // 4 x 16        pivot diagonals (ring by stage & 3)
  static constexpr int UF = DGB + 4 * PIV;               // FR            accumulated rhs updates
  static constexpr int DUM = UF + FR;                    // 64            where the extraction parks the lanes that have nothing to store
  static constexpr int PSB = DUM + 64;                   // 4 x 16 ints   pivot slots (ring)
  static constexpr int JM = PSB + 4 * PIV / 2;           // 4 x FR bytes  slot -> pivot index (ring)
  static constexpr int PM = JM + 4 * FR / 8;             // 4 x 8 ints    pivot-slot bit masks (ring)
  static constexpr int MIV = PM + 4 * 8 / 2;             // 2 x 16 x PLD  (L D L^T)^-1 of the pivot blocks of stages k, k+1
  static constexpr int PAN = MIV + 2 * PIV * PLD;        // 4 panels of (F+1) x PLD, stage s in panel s & 3
  static constexpr int YB = PAN + 4 * PSZ;               // 2 x 16 x F    -V_s transposed ([column][slot]), stage s in s & 1
  static constexpr int VAR = (YB + 2 * YSZ + 1) & ~1;    // dbuf, then (ints) sbuf, then the cells A
  // backward sweep (the panels are dead): solution by slot, partial sums, sweep tables
  static constexpr int XS = PAN, RED = XS + FR, NXP = RED + 2 * 16 * PIV + 64;
};
inline size_t kkt4_sweep_base_bytes(int F, int NS) {
  const int FR = (F + 63) & ~63;
  return (PIV * PLD + PIV + 4 * PIV + FR + 64 + 4 * PIV / 2 + 4 * FR / 8 + 16 + 2 * PIV * PLD + FR + 2 * 16 * PIV + 64) * sizeof(double) + (size_t)NS * 12 * sizeof(int);
}
inline size_t kkt4_lds_bytes(int F, int NS, int max_srec, int max_drec, int n_cells) {
  const int FR = (F + 63) & ~63, PSZ = (F + 1) * PLD;
  size_t o = PIV * PLD + PIV + 4 * PIV + FR + 64 + 4 * PIV / 2 + 4 * FR / 8 + 16 + 2 * PIV * PLD + 4 * (size_t)PSZ + 2 * (size_t)F * PIV;
  o = (o + 1) & ~(size_t)1;
  o += kkt2_dbuf_doubles(F, max_drec);
  size_t oi = 2 * o + kkt2_sbuf_ints(F, max_srec);
  oi += 2 * (((size_t)n_cells + 1) & ~(size_t)1);
  const size_t sweep = (PIV * PLD + PIV + 4 * PIV + FR + 64 + 4 * PIV / 2 + 4 * FR / 8 + 16 + 2 * PIV * PLD + FR + 2 * 16 * PIV + 64) * sizeof(double) + (size_t)NS * 12 * sizeof(int);
  return std::max(oi * sizeof(int), sweep);
}

template <int F>
__global__ __launch_bounds__(KT2) void k_kkt4(DevPlan P, DevWork W, int B) {
  static_assert(F <= 128 && F % 16 == 0, "k_kkt4: fronts of up to 128 slots");
  const int b = blockIdx.x;
  if (b >= B || W.done[b] || W.chord[b] == 1) return;   // (a problem flagged for a chord step is k_chord's)
  extern __shared__ double lds[];
  using LY = Kkt4Layout<F>;
  constexpr int NT = LY::NT, FR = LY::FR, PSZ = LY::PSZ, YSZ = LY::YSZ;
  // Schur tiles: MAXT per update wave (tiles iu + NUW i), the rest on the tile waves 1 .. NT-1 and the right-hand-side wave,
  // MAXTT each (tiles NUW MAXT + wv - 1 + NT i): those apply their update and extract in phase F, where they would idle and
  // the matrix pipe is free of the panel products
  constexpr int NTILE = NT * (NT + 1) / 2, NUW = 15 - NT, MAXT = (NTILE + 14) / 15;
  constexpr int MAXTT = NTILE > NUW * MAXT ? (NTILE - NUW * MAXT + NT - 1) / NT : 0;
  const int tid = threadIdx.x, NS = P.n_stages, n = P.n_sol;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
  double *Lib = lds + LY::LIB, *dvb = lds + LY::DVB, *dgb = lds + LY::DGB, *UF = lds + LY::UF, *dum = lds + LY::DUM;
  double *MIV = lds + LY::MIV, *PAN = lds + LY::PAN, *YB = lds + LY::YB;
  int *psb = (int *)(lds + LY::PSB);
  unsigned char *jmb = (unsigned char *)(lds + LY::JM);
  unsigned *pm = (unsigned *)(lds + LY::PM);
  const int dstride = (int)kkt2_dbuf_doubles(F, P.max_drec), sstride = (int)kkt2_sbuf_ints(F, P.max_srec);
  double *const dbuf = lds + LY::VAR;
  int *const sbuf = (int *)(dbuf + dstride);
  double *A = (double *)(sbuf + sstride);   // cells of the assembled entries
  const double *stream = W.stream + (size_t)b * P.stream_len;
  double *panel = W.panel + (size_t)b * P.panel_stride;
  double *dx = W.dx + (size_t)b * n;
  const int pstride = (F + 1) * PIV;   // per stage: w (16), V (F x 16)

  for (int i = tid; i < P.n_cells; i += KT2) A[i] = 0.0;
  for (int i = tid; i < 4 * PSZ + 2 * YSZ; i += KT2) PAN[i] = 0.0;   // (the panels and the operand buffers behind them)
  for (int i = tid; i < 2 * PIV * PLD; i += KT2) MIV[i] = 0.0;
  for (int i = tid; i < FR; i += KT2) UF[i] = 0.0;
  if (tid < 32) pm[tid] = 0u;
  for (int v = tid; v < n; v += KT2) dx[v] = 0.0;
  __syncthreads();
  // header of a stage into ring slot s & 3 (one wave; pivot slots and diagonals straight from global memory)
  auto publish_header = [&](int s) __attribute__((always_inline)) {
    const int rs = s & 3;
    if (lane < 8) pm[rs * 8 + lane] = 0u;
    if (lane < PIV) {
      const int slot = P.piv_slot[s * PIV + lane];
      const double dg = stream[P.drec_off[s] + lane];
      psb[rs * PIV + lane] = slot;
      jmb[rs * FR + (slot & ~15) + (slot & 3) * 4 + ((slot >> 2) & 3)] = (unsigned char)lane;
      dgb[rs * PIV + lane] = dg;
      atomicOr(&pm[rs * 8 + (slot >> 5)], 1u << (slot & 31));
    }
  };
  if (wv == 0) publish_header(0);
  __syncthreads();

#ifdef QTOS_STAMPS
  __shared__ unsigned long long st2[16][12];
  unsigned long long ts_ = 0;
  if (tid < 192) st2[tid / 12][tid % 12] = 0;
  __syncthreads();
  KS2_START();
#endif
  int tile_lane = (li * PLD + lk) * 8;        // operand B: row li, column lk of a 16 x 16 tile of a panel, in bytes
  int ytile_lane = (lk * F + li) * 8;         // operand A: the transposed -V: [column lk][slot li]
  asm volatile("" : "+v"(tile_lane), "+v"(ytile_lane));
  unsigned ge4_keep = 0u, gt4_keep = 0u;
#pragma unroll
  for (int g = 0; g < 4; ++g) { ge4_keep |= (lk + 4 * g >= li ? 1u : 0u) << (4 * g); gt4_keep |= (lk + 4 * g > li ? 1u : 0u) << (4 * g); }
  // update k-1 on a wave's tiles: U -= V_{k-1} P_{k-1}', operand B = the panel itself with the rows of the pivots of stages k
  // and k+1 blanked on load (the rows of stage k-1's own pivots are zero since its factorisation)
  auto update_tiles = [&](auto &U, const auto &rcs, int k) __attribute__((always_inline)) {
    constexpr int NTL = sizeof(U) / sizeof(U[0]);
    const double *Yp = YB + ((k - 1) & 1) * YSZ, *Pp = PAN + ((k - 1) & 3) * PSZ;
    const Mask256 ma = load_mask8(pm + (k & 3) * 8, lane);
    Mask256 mb = load_mask8(pm + ((k + 1) & 3) * 8, lane);
    if (k + 1 >= NS) mb.v = 0;
    double wa[2][4], pbv[2][4];
    auto tile_loads = [&](int rc, double (&w)[4], double (&pq)[4]) __attribute__((always_inline)) {
      const int R = rc < 0 ? 0 : rc >> 8, C = rc < 0 ? 0 : rc & 255;
      const int oR = __builtin_amdgcn_readfirstlane(R * (16 * 8)), oC = __builtin_amdgcn_readfirstlane(C * (16 * PLD * 8));
      const char *wrow = (const char *)Yp + (ytile_lane + oR), *prow2 = (const char *)Pp + (tile_lane + oC);
      const bool blank = ((grp16(ma, C) | grp16(mb, C)) >> li) & 1u;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        w[s4] = *(const double *)(wrow + 4 * F * 8 * s4);
        const double pv = *(const double *)(prow2 + 32 * s4);
        pq[s4] = blank ? 0.0 : pv;
      }
    };
    tile_loads(rcs[0], wa[0], pbv[0]);
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      if (t + 1 < NTL) tile_loads(rcs[t + 1], wa[(t + 1) & 1], pbv[(t + 1) & 1]);
      if (rcs[t] >= 0) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          U[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[t & 1][s4], pbv[t & 1][s4], U[t], 0, 0, 0);
      }
    }
  };
  // the columns of stage k+2 (Schur updates up to k-1) out of a wave's tiles into that stage's panel, zeroed in place
  auto extract_tiles = [&](auto &U, const auto &rcs, int k) __attribute__((always_inline)) {
    constexpr int NTL = sizeof(U) / sizeof(U[0]);
    const Mask256 m2 = load_mask8(pm + ((k + 2) & 3) * 8, lane);
    const unsigned char *jm2 = jmb + ((k + 2) & 3) * FR;
    double *Xnn = PAN + ((k + 2) & 3) * PSZ;
    double *dummy = dum + lane;
    const unsigned ge4 = ge4_keep, gt4 = gt4_keep;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int rc = rcs[t];
      if (rc < 0) continue;
      const int R = rc >> 8, C = rc & 255;
      const unsigned cw2 = grp16(m2, C), rw2 = grp16(m2, R);
      if ((cw2 | rw2) == 0u) continue;
      const int jc = jm2[16 * C + (li & 3) * 4 + (li >> 2)];
      const int jr32 = *(const int *)(jm2 + 16 * R + 4 * lk);
      const unsigned cm = ((cw2 >> li) & 1u) ? (R > C ? 0x1111u : ge4) : 0u;
      const unsigned rmk = (rw2 >> lk) & (R > C ? 0x1111u : gt4);
      double *xr = Xnn + (16 * R + lk) * PLD + jc, *xc = Xnn + (16 * C + li) * PLD;
      if (cw2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *(((cm >> (4 * g)) & 1u) ? xr + g * 4 * PLD : dummy) = U[t][g];
      }
      if (rw2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *(((rmk >> (4 * g)) & 1u) ? xc + ((jr32 >> (8 * g)) & 255) : dummy) = U[t][g];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int z = ~__builtin_amdgcn_sbfe((int)(cm | rmk), 4 * g, 1);
        U[t][g] = __hiloint2double(__double2hiint(U[t][g]) & z, __double2loint(U[t][g]) & z);
      }
    }
  };
  auto tile_of = [](int t) { int R = 0; while (((R + 1) * (R + 2)) >> 1 <= t) ++R; return (R << 8) | (t - ((R * (R + 1)) >> 1)); };
  const int tid_outer = tid, lane_outer = lane;
  if (wv > NT) {
    // ================================ update waves ===========================================================
    const int iu = wv - NT - 1;
    d4_t U[MAXT];
    int tRC[MAXT];   // (R << 8) | C, or -1
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      U[i] = d4_t{0.0, 0.0, 0.0, 0.0};
      const int t = iu + NUW * i;
      tRC[i] = (t < NTILE && i < MAXT) ? tile_of(t) : -1;
    }
    for (int k = -2; k < NS; ++k) {
      int lane = lane_outer;
      asm volatile("" : "+v"(lane));
      const int li = lane & 15, lk = lane >> 4;
      KS2(7);
      // ---- LDS-DMA of record k+2 into the (single) record buffers: its predecessor was assembled in the previous stage
      if (k + 2 < NS) {
        const int s = k + 2;
        int d0, d1, s0, s1;
        sload2(P.drec_off + s, d0, d1);
        sload2(P.srec_off + s, s0, s1);
        const int nbd = (d1 - d0) * 8, nbs = (s1 - s0) * 4, ncd = (nbd + 1023) >> 10;
        const char *gd = (const char *)(stream + d0), *gs = (const char *)(P.srec + s0);
        typedef __attribute__((address_space(3))) char lds_char;
        lds_char *ld = (lds_char *)dbuf, *ls = (lds_char *)sbuf;
        for (int c = iu; c * 1024 < nbd; c += NUW)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gd + min(c * 1024 + lane * 16, nbd - 16)), (__attribute__((address_space(3))) void *)(ld + c * 1024), 16, 0, 0);
        for (int c = (iu + NUW - ncd % NUW) % NUW; c * 1024 < nbs; c += NUW)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gs + min(c * 1024 + lane * 16, nbs - 16)), (__attribute__((address_space(3))) void *)(ls + c * 1024), 16, 0, 0);
      }
      KS2(8);
      int rcs[MAXT];
#pragma unroll
      for (int t = 0; t < MAXT; ++t) { rcs[t] = __builtin_amdgcn_readfirstlane(tRC[t]); asm volatile("" : "+s"(rcs[t])); }
#ifndef QTOS_K4_UPRIO
#define QTOS_K4_UPRIO 0
#endif
      if (QTOS_K4_UPRIO) __builtin_amdgcn_s_setprio(QTOS_K4_UPRIO);
      if (k >= 1) update_tiles(U, rcs, k);
      if (QTOS_K4_UPRIO) __builtin_amdgcn_s_setprio(0);
      KS2(5);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the record has landed (the other waves read it behind the barrier)
      lds_barrier();
      KS2(1);
      if (k + 2 < NS) extract_tiles(U, rcs, k);
      KS2(2);
      // ---- share of the assembly of record k+2 (update waves: the low item indices, i.e. the targets with the fewest terms)
#ifndef QTOS_K4_USHARE
#define QTOS_K4_USHARE 1
#endif
      // (QTOS_K4_USHARE = 0: the update waves take no part in the assembly, the NT tile-side waves share it)
      if (QTOS_K4_USHARE && k + 2 < NS) assemble_stage(A, F, sbuf, dbuf, iu * 64 + lane, 15 * 64);
      KS2(3);
      lds_barrier();
      KS2(4);
    }
  } else {
    // ================================ tile waves, right-hand-side wave, factor wave ============================
    double *minv_g = W.minv + (size_t)b * NS * (PIV * PIV);   // inverse of every pivot block, kept for chord steps
    auto factor_block = [&](double *Pn, const int myps, const int *ps, double *Minv, int ks) __attribute__((always_inline)) {
      // split layout (ldlt16s, kernels.hpp): lane (li, lk) holds row li of the block, columns c = 4 g + lk; entries above the
      // diagonal are read from the mirrored position: the block that is factored is exactly symmetric
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
        Lib[li * PLD + 4 * g + lk] = wi[g];
        prow_p[4 * g] = 0.0;   // the pivot rows leave the panel
      }
      if (lk == (li & 3)) dvb[li] = myinv;
      double zero = 0.0;
      asm volatile("" : "+v"(zero));
      d4_t mi = {zero, zero, zero, zero};
      double lt[4], ld[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) { lt[s4] = Lib[(lk + 4 * s4) * PLD + li]; ld[s4] = dvb[lk + 4 * s4]; }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) mi = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[s4], lt[s4] * ld[s4], mi, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        Minv[(lk + 4 * g) * PLD + li] = mi[g];
        minv_g[(size_t)ks * (PIV * PIV) + (lk + 4 * g) * PIV + li] = mi[g];
      }
    };
    int prow_next = 0;   // pivot slot li of stage k+1
    typedef unsigned short us4_t __attribute__((ext_vector_type(4)));
    const us4_t *ctab4 = (const us4_t *)P.ctab;
    us4_t ct_cur = {0, 0, 0, 0};
    int rc_cur = 0;      // cell of the assembled rhs of pivot li of stage k+1
    double vp[4] = {0.0, 0.0, 0.0, 0.0};   // V_{k-1} of this wave's tile rows (accumulator layout)
    constexpr int NTT = MAXTT > 0 ? MAXTT : 1;
    d4_t UT[NTT];
    int tRCT[NTT];
#pragma unroll
    for (int i = 0; i < NTT; ++i) {
      UT[i] = d4_t{0.0, 0.0, 0.0, 0.0};
      const int t = NUW * MAXT + (wv - 1) + NT * i;
      tRCT[i] = (wv >= 1 && i < MAXTT && t < NTILE) ? tile_of(t) : -1;
    }
    for (int k = -2; k < NS; ++k) {
      int tid = tid_outer, lane = lane_outer;
      asm volatile("" : "+v"(tid), "+v"(lane));
      const int li = lane & 15, lk = lane >> 4;
      double *Pk = PAN + (k & 3) * PSZ, *Pn = PAN + ((k + 1) & 3) * PSZ, *Pp = PAN + ((k - 1) & 3) * PSZ;
      const double *Mk = MIV + (k & 1) * PIV * PLD;
      const bool has_next = k + 1 >= 0 && k + 1 < NS;
      KS2(7);
      const us4_t ct_nxt = ctab4[((size_t)min(k + 2, NS - 1) * NT + min(wv, NT - 1)) * 64 + lane];
      const int rc_nxt = P.rtab[min(k + 2, NS - 1) * PIV + li];
      // ---- phase A(k) ----
      if (wv < NT) {
        if (k >= -1) {
          const int R = wv;
          const Mask256 m1 = load_mask8(pm + ((k + 1) & 3) * 8, lane);   // pivot slots of stage k+1
          Mask256 m2 = load_mask8(pm + ((k + 2) & 3) * 8, lane);         // ... of stage k+2
          if (k + 2 >= NS) m2.v = 0;
          const unsigned am_word = k >= 0 ? P.amask[k * 8 + (R >> 1)] : 0u;
          const int prow = has_next ? prow_next : 0;
          double pr[4], lm[4];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            pr[s4] = Pk[(16 * R + li) * PLD + lk + 4 * s4];
            lm[s4] = Mk[li * PLD + lk + 4 * s4];
          }
          double zero = 0.0;
          asm volatile("" : "+v"(zero));
          // V = P (L D L^T)^-1 in accumulator layout: vt[g] = V[16R+li][lk+4g] -- V itself as the A operand of the next products
          d4_t vt = {zero, zero, zero, zero};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) vt = __builtin_amdgcn_mfma_f64_16x16x4f64(lm[s4], pr[s4], vt, 0, 0, 0);
          // columns of stage k+1: assembled entries, extracted Schur updates (up to k-2), pivot diagonal, minus the two updates
          // the tiles had not seen: V_{k-1} P_{k-1}[piv]' and V_k P_k[piv]' (the raw rows of the next pivots are the B operands)
          // (a pivot of stage k+1 that entered the front only then sits in a slot that belonged to a pivot of stage k: in P_k that
          //  row is zero since the factorisation, in P_{k-1} it is still the old occupant's -- not this unknown's: blank it)
          const bool x2_other = k >= 0 && ((pm[(k & 3) * 8 + (prow >> 5)] >> (prow & 31)) & 1u);
          double npp[4], np2[4], xv[4], av[4];
          int aidx[4];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const int r = 16 * R + lk + 4 * s4;
            npp[s4] = Pk[prow * PLD + lk + 4 * s4];
            np2[s4] = x2_other ? 0.0 : Pp[prow * PLD + lk + 4 * s4];
            xv[s4] = Pn[r * PLD + li];
            aidx[s4] = ct_cur[s4];
            av[s4] = A[aidx[s4]];
          }
          const double dgn = dgb[((k + 1) & 3) * PIV + li];
          d4_t acc;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int r = 16 * R + lk + 4 * g;
            acc[g] = xv[g] + av[g] + (r == prow ? dgn : 0.0);
            npp[g] = -npp[g];
            np2[g] = -np2[g];
          }
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[s4], np2[s4], acc, 0, 0, 0);   // acc -= V_{k-1} P_{k-1}[piv]'
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vt[s4], npp[s4], acc, 0, 0, 0);   // acc -= V_k P_k[piv]'
#pragma unroll
          for (int g = 0; g < 4; ++g) A[aidx[g]] = 0.0;   // retired (the zero cell stays zero)
          {
            // operand A of update k: -V, transposed; the rows of the pivots of stages k+1 and k+2 have left the tiles by then
            const bool gone = ((grp16(m1, R) | grp16(m2, R)) >> li) & 1u;
            double *Yk = YB + (k & 1) * YSZ;
#pragma unroll
            for (int g = 0; g < 4; ++g) Yk[(lk + 4 * g) * F + 16 * R + li] = (gone || !has_next) ? 0.0 : -vt[g];
          }
          if (has_next) {
#pragma unroll
            for (int g = 0; g < 4; ++g) Pn[(16 * R + lk + 4 * g) * PLD + li] = acc[g];
          }
          const unsigned am16 = (am_word >> ((R & 1) * 16)) & 0xffffu;
          if ((am16 >> li) & 1u) {
            double *pv = panel + (size_t)k * pstride + PIV;
            *(d4_t *)(pv + (16 * R + li) * PIV + 4 * lk) = vt;
          }
          // V_k for the next stage's second term -- without the rows of stage k+1's pivots: their slots change hands when that
          // stage is over, and what sits there in P_{k+2} is another unknown's row
          const bool leaves = (grp16(m1, R) >> li) & 1u;
#pragma unroll
          for (int g = 0; g < 4; ++g) vp[g] = leaves ? 0.0 : vt[g];
        }
      } else if (k >= -1) {
        // right-hand-side row: w = (L D L^T)^-1 p_F; rhs -= P w; right-hand side of the next pivots
        double part = 0.0;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) part = fma(Mk[li * PLD + lk + 4 * s4], Pk[F * PLD + lk + 4 * s4], part);
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
            Pn[F * PLD + lane] = A[rc_cur] + UF[c];
            A[rc_cur] = 0.0;
            UF[c] = 0.0;
          }
        }
      }
      KS2(0);
      lds_barrier();
      KS2(1);
      // ---- phase F(k) ----
      if (wv == 0) {
        __builtin_amdgcn_s_setprio(3);
        if (has_next) factor_block(Pn, prow_next, psb + ((k + 1) & 3) * PIV, MIV + ((k + 1) & 1) * PIV * PLD, k + 1);
        __builtin_amdgcn_s_setprio(0);
      } else {
        if (wv == NT && k + 3 < NS) publish_header(k + 3);
        if constexpr (MAXTT > 0) {
          int rct[NTT];
#pragma unroll
          for (int t = 0; t < NTT; ++t) { rct[t] = __builtin_amdgcn_readfirstlane(tRCT[t]); asm volatile("" : "+s"(rct[t])); }
          if (k >= 1) update_tiles(UT, rct, k);
          KS2(5);
          if (k + 2 < NS) extract_tiles(UT, rct, k);
          KS2(3);
        }
        // shares of the assembly of record k+2 behind the update waves' (item indices NUW ..)
        if (k + 2 < NS) {
          if (QTOS_K4_USHARE) assemble_stage(A, F, sbuf, dbuf, (NUW + wv - 1) * 64 + lane, 15 * 64);
          else assemble_stage(A, F, sbuf, dbuf, (wv - 1) * 64 + lane, NT * 64);
        }
      }
      KS2(2);
      lds_barrier();
      KS2(4);
      if (k + 2 < NS) prow_next = psb[((k + 2) & 3) * PIV + li];
      ct_cur = ct_nxt;
      rc_cur = rc_nxt;
    }
  }
  // ---- backward substitution (sweep_backward: one barrier per stage, one-stage look-ahead) -----------------------
  __syncthreads();  // drains the factor-panel stores: they are read back below
  KS2(7);
  {
    double *xs = lds + LY::XS, *red = lds + LY::RED;
    int *nxp = (int *)(lds + LY::NXP);
    for (int i = tid; i < FR; i += KT2) xs[i] = 0.0;
    for (int i = tid; i < NS * 4; i += KT2) nxp[i] = P.nxt_pack[i];
    for (int i = tid; i < NS * 8; i += KT2) nxp[NS * 4 + i] = (int)P.amask2[i];
    __syncthreads();
    const SweepDs sd = {W.stream + (size_t)b * P.stream_len, W.ds + (size_t)b * P.n_cons, W.g + (size_t)b * P.n_cons, W.s + (size_t)b * P.n_cons};
    sweep_backward<F>(P, panel, dx, W.sol + (size_t)b * P.n_stages * PIV, xs, red, nxp, wv, lane, lds + ((LY::NXP + NS * 6 + 1) & ~1), sd);
  }
#ifdef QTOS_STAMPS
  KS2(6);
  __syncthreads();
  if (tid < 192 && W.trace) W.trace[((size_t)b * (P.max_iter + 1) + 16) * 4 + tid] = (double)st2[tid / 12][tid % 12];
#endif
}

}  // namespace qtos
