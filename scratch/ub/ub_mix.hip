// does a wave's f64 MFMA stream slow OTHER kinds of instructions of a second wave on the same SIMD? (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void k(double *io, unsigned long long *cyc, int mode, int with_mfma, int victim) {
  __shared__ double lds[4096];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = 1.0;
  __syncthreads();
  double a = io[tid], b = io[tid + 1], c = io[tid + 2];
  int x = tid, y = tid * 3 + 1;
  d4_t acc = {a, b, c, a};
  unsigned long long t0 = 0, t1 = 0;
  if (wv == 0 && with_mfma) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a), "+v"(x)::"memory");
#pragma unroll 16
    for (int i = 0; i < 1024; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);
    a += acc[0];
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a), "+v"(x)::"memory");
  }
  if (wv == victim) {
    asm volatile("s_sleep 20\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a), "+v"(x)::"memory");
    if (mode == 0) {          // dependent f64 FMA
#pragma unroll
      for (int i = 0; i < 512; ++i) a = fma(a, b, c);
    } else if (mode == 1) {   // int VALU dependent
#pragma unroll
      for (int i = 0; i < 512; ++i) x = x * 3 + y;
    } else if (mode == 2) {   // LDS reads (independent)
      double s = 0;
#pragma unroll
      for (int i = 0; i < 512; ++i) s += lds[(tid + i * 64) & 4095];
      a += s;
    } else if (mode == 3) {   // LDS stores
#pragma unroll
      for (int i = 0; i < 512; ++i) lds[(lane + i * 64) & 4095] = a;
    } else if (mode == 4) {   // v_and / cndmask mix (32-bit VALU, independent pairs)
      int z = y;
#pragma unroll
      for (int i = 0; i < 256; ++i) { x = (x & y) ^ i; z = (z | x) + i; }
      x += z;
    } else if (mode == 5) {   // f32 FMA dependent
      float f = (float)a, g = (float)b;
#pragma unroll
      for (int i = 0; i < 512; ++i) f = fmaf(f, g, g);
      a += f;
    } else if (mode == 7) {   // MFMA
      d4_t ac2 = {a, a, a, a};
#pragma unroll
      for (int i = 0; i < 64; ++i) ac2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, ac2, 0, 0, 0);
      a += ac2[0];
    } else if (mode == 8) {   // 4 independent f64 FMA chains
      double a1 = a, a2 = b, a3 = c, a4 = a + b;
#pragma unroll
      for (int i = 0; i < 128; ++i) { a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); a4 = fma(a4, b, c); }
      a = a1 + a2 + a3 + a4;
    } else if (mode == 6) {   // scalar ALU
      int s = __builtin_amdgcn_readfirstlane(x);
#pragma unroll
      for (int i = 0; i < 512; ++i) { s = s * 3 + 1; asm volatile("" : "+s"(s)); }
      x += s;
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a), "+v"(x)::"memory");
  }
  cyc[wv] = t1 - t0;
  io[tid] = a + x;
}
__global__ void kid(int *out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (int)id;
}
int main() {
  int *ids; (void)hipMalloc(&ids, 64 * 4);
  int same = 4, other = 1;
  for (int nt : {1024, 512}) {
    hipLaunchKernelGGL(kid, dim3(1), dim3(nt), 0, 0, ids); (void)hipDeviceSynchronize();
    int h[16]; (void)hipMemcpy(h, ids, 64, hipMemcpyDeviceToHost);
    printf("block of %d threads: SIMD of wave w:", nt);
    for (int w = 0; w < nt / 64; ++w) printf(" %d", (h[w] >> 4) & 3);
    printf("   (wave slot:");
    for (int w = 0; w < nt / 64; ++w) printf(" %d", h[w] & 15);
    printf(")\n");
    if (nt == 512) { same = -1; for (int w = 1; w < 8; ++w) { if (((h[w] >> 4) & 3) == ((h[0] >> 4) & 3) && same < 0) same = w; } for (int w = 1; w < 8; ++w) if (((h[w] >> 4) & 3) != ((h[0] >> 4) & 3)) { other = w; break; } }
  }
  printf("partner of wave 0 on its SIMD: wave %d; on another SIMD: wave %d\n", same, other);
  double *io; unsigned long long *cyc;
  (void)hipMalloc(&io, 8 * 2048); (void)hipMalloc(&cyc, 64);
  (void)hipMemset(io, 0, 8 * 2048);
  hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, io, cyc, 0, 1, 4); (void)hipDeviceSynchronize();   // warm the instruction cache
  const char *names[] = {"dep f64 FMA x512", "dep int mad x512", "LDS read b64 x512", "LDS store b64 x512", "int and/or x512", "dep f32 FMA x512", "salu x512", "MFMA x64", "4 indep f64 FMA x128"};
  for (int mode = 0; mode < 9; ++mode) {
    printf("%-20s", names[mode]);
    for (int victim : {same, other})
      for (int with : {0, 1}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, io, cyc, mode, with, victim);
        (void)hipDeviceSynchronize();
        unsigned long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
        printf("  %s wave %d (%s SIMD): %6llu", with ? "with MFMA on wave 0," : "alone,", victim, victim == same ? "same" : "other", c[victim]); if (with) printf(" (wave 0: %llu)", c[0]);
      }
    printf("\n");
  }
  return 0;
}
