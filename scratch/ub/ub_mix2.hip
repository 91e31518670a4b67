// one SIMD (waves 0, 4, 8, 12 of a 1024-thread block): a dependent f64 VALU chain on wave 0 against 0..3 waves of f64 MFMA
// streams; and bursts of 12 MFMAs per wave (3 accumulators x 4 dependent) separated by other work, as k_kkt2's update loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void k1(double *io, unsigned long long *cyc, unsigned mfma_mask, int prio) {
  const int tid = threadIdx.x, wv = tid >> 6;
  double a = io[tid], b = io[tid + 1], c = io[tid + 2];
  d4_t acc = {b, c, b, c};
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if (wv == 0) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    asm volatile("s_sleep 4\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a)::"memory");
#pragma unroll
    for (int i = 0; i < 512; ++i) a = fma(a, b, c);
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a)::"memory");
  } else if ((mfma_mask >> wv) & 1u) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(b)::"memory");
#pragma unroll 8
    for (int i = 0; i < 128; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);
    a += acc[0];
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a)::"memory");
  }
  cyc[wv] = t1 - t0;
  if (a == 123.456) io[tid] = a;
}
// bursts: every wave in mask: [gap of int VALU work] then 3 tiles x 4 MFMAs, REP times; per-wave total time
template <int GAP, bool MUL>
__global__ void k2(double *io, unsigned long long *cyc, unsigned mask) {
  const int tid = threadIdx.x, wv = tid >> 6;
  double b = io[tid + 1], c = io[tid + 2], dv = io[tid + 3];
  d4_t U[3];
  for (int i = 0; i < 3; ++i) U[i] = d4_t{b, c, b, c};
  int x = tid;
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if ((mask >> wv) & 1u) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(b)::"memory");
    for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          double w = b;
          if (MUL) { asm volatile("" : "+v"(w)); w = w * dv; }
          U[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, c, U[t], 0, 0, 0);
        }
#pragma unroll
      for (int i = 0; i < GAP; ++i) { x = x * 3 + 1; }
      asm volatile("" : "+v"(x));
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(b)::"memory");
  } else {
    for (int rep = 0; rep < 16; ++rep) __builtin_amdgcn_s_barrier();
  }
  cyc[wv] = t1 - t0;
  double s = U[0][0] + U[1][1] + U[2][2] + x;
  if (s == 123.456) io[tid] = s;
}
int main() {
  double *io; unsigned long long *cyc;
  (void)hipMalloc(&io, 8 * 4096); (void)hipMalloc(&cyc, 128);
  (void)hipMemset(io, 0, 8 * 4096);
  unsigned long long c[16];
  for (int prio : {0, 1})
    for (unsigned m : {0x0000u, 0x0010u, 0x0110u, 0x1110u, 0x0002u, 0x0eeeu}) {
      for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k1, dim3(1), dim3(1024), 0, 0, io, cyc, m, prio);
      (void)hipDeviceSynchronize(); (void)hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
      printf("wave 0: 512 dependent f64 FMA (setprio %d) with MFMA streams (128 each) on waves %04x: wave 0 %6llu;  MFMA waves:", prio * 3, m, c[0]);
      for (int w = 1; w < 16; ++w) if ((m >> w) & 1u) printf(" %llu", c[w]);
      printf("\n");
    }
  auto show = [&](const char *name, unsigned m) {
    (void)hipDeviceSynchronize(); (void)hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
    printf("%s, waves %04x (cycles per burst round of 12 MFMAs/wave):", name, m);
    for (int w = 0; w < 16; ++w) if ((m >> w) & 1u) printf(" %5llu", c[w] / 16);
    printf("\n");
  };
  for (unsigned m : {0x0001u, 0x0011u, 0x0111u, 0x1111u, 0xeeeeu, 0xffffu}) {
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k2<0, false>), dim3(1), dim3(1024), 0, 0, io, cyc, m); show("bursts, no gap, no mul", m);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k2<0, true>), dim3(1), dim3(1024), 0, 0, io, cyc, m); show("bursts, no gap, f64 mul per MFMA", m);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k2<200, false>), dim3(1), dim3(1024), 0, 0, io, cyc, m); show("bursts, 200 int ops gap, no mul", m);
  }
  return 0;
}
