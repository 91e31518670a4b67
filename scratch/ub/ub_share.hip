// do two waves on one SIMD share the VALU / LDS / MFMA issue rate? (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void k(double *io, unsigned long long *cyc, int mode, unsigned active_mask) {
  __shared__ double lds[4096];
  const int tid = threadIdx.x, wv = tid >> 6;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = 1.0;
  __syncthreads();
  double a = io[tid], b = io[tid + 1], c = io[tid + 2];
  int x = tid, y = tid * 3 + 1;
  d4_t acc = {a, b, c, a};
  unsigned long long t0 = 0, t1 = 0;
  if ((active_mask >> wv) & 1u) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a), "+v"(x)::"memory");
    if (mode == 0) {          // dependent f64 FMA
#pragma unroll
      for (int i = 0; i < 512; ++i) a = fma(a, b, c);
    } else if (mode == 1) {   // int VALU
#pragma unroll
      for (int i = 0; i < 512; ++i) x = x * 3 + y;
    } else if (mode == 2) {   // LDS reads (independent)
      double s = 0;
#pragma unroll
      for (int i = 0; i < 512; ++i) s += lds[(tid + i * 64) & 4095];
      a += s;
    } else {                  // MFMA
#pragma unroll
      for (int i = 0; i < 128; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);
      a += acc[0];
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a), "+v"(x)::"memory");
  }
  cyc[wv] = t1 - t0;
  io[tid] = a + x;
}
int main() {
  double *io; unsigned long long *cyc;
  (void)hipMalloc(&io, 8 * 2048); (void)hipMalloc(&cyc, 64);
  (void)hipMemset(io, 0, 8 * 2048);
  const char *names[] = {"dep f64 FMA x512", "int mad x512", "LDS read b64 x512", "MFMA f64 x128"};
  for (int mode = 0; mode < 4; ++mode)
    for (unsigned mask : {0x01u, 0x11u, 0x03u, 0xffu}) {
      hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, io, cyc, mode, mask);
      (void)hipDeviceSynchronize();
      unsigned long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
      printf("%-20s waves %02x:", names[mode], mask);
      for (int w = 0; w < 8; ++w) printf(" %6llu", c[w]);
      printf("\n");
    }
  return 0;
}
