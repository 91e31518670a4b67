// one SIMD: rounds of 12 MFMAs per wave (3 accumulators x 4 dependent), completion forced at the end of each round
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
// GAPKIND: 0 none, 1 int VALU (dependent), 2 LDS reads, 3 scalar ALU, 4 f32 VALU, 5 s_sleep
template <int GAPKIND, bool BARRIER>
__global__ void k2(double *io, unsigned long long *cyc, unsigned mask) {
  __shared__ double lds[4096];
  const int tid = threadIdx.x, wv = tid >> 6;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = 1.0;
  double b = io[tid + 1], c = io[tid + 2];
  d4_t U[3];
  for (int i = 0; i < 3; ++i) U[i] = d4_t{b, c, b, c};
  int x = tid; float f = (float)b; double ls = 0;
  int sx = __builtin_amdgcn_readfirstlane(x);
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if ((mask >> wv) & 1u) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(b)::"memory");
    for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) U[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, U[t], 0, 0, 0);
      if (GAPKIND == 1) {
#pragma unroll
        for (int i = 0; i < 100; ++i) { x = (x ^ i) + (x >> 3); }
      } else if (GAPKIND == 2) {
#pragma unroll
        for (int i = 0; i < 24; ++i) ls += lds[(tid + i * 64 + rep) & 4095];
      } else if (GAPKIND == 3) {
#pragma unroll
        for (int i = 0; i < 100; ++i) { sx = sx * 3 + 1; asm volatile("" : "+s"(sx)); }
      } else if (GAPKIND == 4) {
#pragma unroll
        for (int i = 0; i < 100; ++i) f = fmaf(f, 1.0001f, 0.5f);
      } else if (GAPKIND == 5) {
        asm volatile("s_sleep 8");
      }
      // completion of this round's MFMAs
      asm volatile("s_nop 0" : "+v"(U[0]), "+v"(U[1]), "+v"(U[2]));
      double sink = U[0][0] + U[1][0] + U[2][0];
      asm volatile("" :: "v"(sink));
      if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(b)::"memory");
  } else if (BARRIER) {
    for (int rep = 0; rep < 16; ++rep) __builtin_amdgcn_s_barrier();
  }
  cyc[wv] = t1 - t0;
  double s = U[0][0] + U[1][1] + U[2][2] + x + f + ls + sx;
  if (s == 123.456) io[tid] = s;
}
// DP chain on a younger / older wave than the MFMA stream
__global__ void k1(double *io, unsigned long long *cyc, int dpw, int mw) {
  const int tid = threadIdx.x, wv = tid >> 6;
  double a = io[tid], b = io[tid + 1], c = io[tid + 2];
  d4_t acc = {b, c, b, c};
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if (wv == dpw) {
    asm volatile("s_sleep 4\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a)::"memory");
#pragma unroll
    for (int i = 0; i < 512; ++i) a = fma(a, b, c);
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a)::"memory");
  } else if (wv == mw) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(b)::"memory");
#pragma unroll 8
    for (int i = 0; i < 256; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);
    a += acc[0];
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a)::"memory");
  }
  cyc[wv] = t1 - t0;
  if (a == 123.456) io[tid] = a;
}
template <int G, bool BAR>
void run(const char *name, double *io, unsigned long long *cyc) {
  unsigned long long c[16];
  for (unsigned m : {0x0001u, 0x0011u, 0x0111u, 0x1111u, 0xffffu}) {
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k2<G, BAR>), dim3(1), dim3(1024), 0, 0, io, cyc, m);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
    printf("%-28s %s waves %04x, cycles per round:", name, BAR ? "barrier/round" : "free-running ", m);
    for (int w = 0; w < 16; ++w) if ((m >> w) & 1u) printf(" %5llu", c[w] / 16);
    printf("\n");
  }
}
int main() {
  double *io; unsigned long long *cyc;
  (void)hipMalloc(&io, 8 * 4096); (void)hipMalloc(&cyc, 128);
  (void)hipMemset(io, 0, 8 * 4096);
  unsigned long long c[16];
  for (int dpw : {0, 4, 12})
    for (int mw : {0, 4, 8, 1}) {
      if (dpw == mw) continue;
      for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k1, dim3(1), dim3(1024), 0, 0, io, cyc, dpw, mw);
      (void)hipDeviceSynchronize(); (void)hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
      printf("512 dependent f64 FMA on wave %2d: %6llu cycles; 256 MFMA on wave %d: %6llu\n", dpw, c[dpw], mw, c[mw]);
    }
  run<0, false>("12 MFMA + completion", io, cyc);
  run<0, true>("12 MFMA + completion", io, cyc);
  run<1, true>("12 MFMA, 100 int VALU", io, cyc);
  run<2, true>("12 MFMA, 24 LDS reads", io, cyc);
  run<3, true>("12 MFMA, 100 SALU", io, cyc);
  run<4, true>("12 MFMA, 100 f32 FMA", io, cyc);
  run<5, true>("12 MFMA, s_sleep 8", io, cyc);
  run<1, false>("12 MFMA, 100 int VALU", io, cyc);
  return 0;
}
