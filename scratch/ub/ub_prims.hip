// micro-benchmarks of the primitives k_kkt is built from (diagnostic only): cycles per operation on one CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define T0() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a), "+v"(b), "+v"(c)::"memory")
#define T1() asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a), "+v"(b), "+v"(c)::"memory")
constexpr int N = 256;
__global__ void k(double *io, unsigned long long *cyc, int nthreads_active) {
  __shared__ double lds[4096];
  __shared__ int ilds[1024];
  const int tid = threadIdx.x;
  unsigned long long t0, t1;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = 1.0 + 1e-9 * i;
  for (int i = tid; i < 1024; i += blockDim.x) ilds[i] = (i * 17 + 5) & 1023;
  __syncthreads();
  double a = io[tid], b = io[tid + 1], c = io[tid + 2], d = io[tid+3];
  // 0: dependent f64 FMA chain
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) a = fma(a, b, c);
  T1(); if (tid == 0) cyc[0] = (t1 - t0);
  // 1: 4 independent f64 FMA chains (throughput)
  double a1 = a + 1, a2 = a + 2, a3 = a + 3;
  T0();
#pragma unroll
  for (int i = 0; i < N / 4; ++i) { a = fma(a, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); }
  a += a1 + a2 + a3;
  T1(); if (tid == 0) cyc[1] = (t1 - t0);
  // 2: dependent DPP f64 fmac chain
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_fmac_f64 %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
  T1(); if (tid == 0) cyc[2] = (t1 - t0);
  // 3: independent DPP fmac (4 accumulators)
  T0();
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    asm volatile("v_fmac_f64 %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(d), "v"(b));
    asm volatile("v_fmac_f64 %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(d), "v"(b));
    asm volatile("v_fmac_f64 %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a2) : "v"(d), "v"(b));
    asm volatile("v_fmac_f64 %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(d), "v"(b));
  }
  T1(); if (tid == 0) cyc[3] = (t1 - t0);
  a += a1 + a2 + a3;
  // 4: LDS pointer chase (ds_read_b32 dependent)
  int p = tid & 1023;
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) p = ilds[p];
  T1(); if (tid == 0) cyc[4] = (t1 - t0);
  a += p;
  // 5: ds_bpermute dependent chain (via __shfl_xor on double = 2 bpermutes)
  T0();
#pragma unroll
  for (int i = 0; i < N / 4; ++i) a += __shfl_xor(a, 16);
  T1(); if (tid == 0) cyc[5] = (t1 - t0) * 4;
  // 6: LDS-only barrier, all waves
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  T1(); if (tid == 0) cyc[6] = (t1 - t0);
  // 7: rcp f64 dependent
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) a = __builtin_amdgcn_rcp(a);
  T1(); if (tid == 0) cyc[7] = (t1 - t0);
  // 8: dependent MFMA f64 16x16x4
  d4_t acc = {a, b, c, d};
  T0(); acc[0] = a;
#pragma unroll
  for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);
  a += acc[0];
  T1(); if (tid == 0) cyc[8] = (t1 - t0);
  // 9: 2 independent MFMA chains
  d4_t acc2 = {d, c, b, a};
  T0(); acc2[0] = a; acc[0] = a;
#pragma unroll
  for (int i = 0; i < N / 2; ++i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, b, acc2, 0, 0, 0); }
  a += acc[1] + acc2[1];
  T1(); if (tid == 0) cyc[9] = (t1 - t0);
  a += acc[0] + acc[1] + acc[2] + acc[3] + acc2[0] + acc2[1] + acc2[2] + acc2[3];
  // 10: ds_read_b64 load -> FMA -> store round trip (dependent through LDS)
  T0();
#pragma unroll
  for (int i = 0; i < N / 4; ++i) { double v = lds[(tid * 17 + i) & 4095]; a = fma(a, v, c); lds[(tid * 17 + i + 1) & 4095] = a; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  T1(); if (tid == 0) cyc[10] = (t1 - t0) * 4;
  // 11: readlane -> VALU use chain
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) { int lo = __builtin_amdgcn_readlane(__double2loint(a), 5); a = a + (double)lo; }
  T1(); if (tid == 0) cyc[11] = (t1 - t0);
  // 12: 32 independent ds_read_b64 then sum (latency of a batch)
  T0();
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += lds[((tid >> 4) + 4 * i + r) & 4095];
    a += s;
  }
  T1(); if (tid == 0) cyc[12] = (t1 - t0) * (N / 8);
  io[tid] = a;
}
int main() {
  double *io; unsigned long long *cyc;
  hipMalloc(&io, 8 * 2048); hipMalloc(&cyc, 8 * 32);
  std::vector<double> h(2048, 1.0000001);
  const char *names[] = {"dep f64 FMA", "4x indep f64 FMA", "dep DPP fmac f64", "4x indep DPP fmac f64", "LDS pointer chase (b32)", "shfl_xor f64 (2 bpermute) dep",
                         "lds barrier", "dep v_rcp_f64", "dep MFMA f64 16x16x4", "2x indep MFMA f64", "LDS load-fma-store round trip", "readlane->cvt->add chain", "32 indep LDS loads + sum (per batch/8)"};
  for (int nt : {64, 128, 256, 512}) {
    (void)hipMemcpy(io, h.data(), 8 * 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, io, cyc, nt);
    (void)hipDeviceSynchronize();
    unsigned long long c[32]; (void)hipMemcpy(c, cyc, 8 * 32, hipMemcpyDeviceToHost);
    printf("threads %d (waves %d):\n", nt, nt / 64);
    for (int i = 0; i < 13; ++i) printf("  %-42s %8.1f cycles/op\n", names[i], (double)c[i] / N);
  }
  return 0;
}
