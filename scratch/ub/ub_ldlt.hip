// micro-benchmark: cycles of the in-register 16x16 LDL^T variants on one wave (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ldlt_variants.hpp"
using namespace qtos;

template <int VAR>
__global__ void k(const double *Bin, double *out, unsigned long long *cyc, int reps, int nwaves) {
  __shared__ double Bs[PIV * PLD], Lm[PIV * PLD], Li[PIV * PLD], dv[PIV];
  const int tid = threadIdx.x;
  for (int e = tid; e < PIV * PIV; e += blockDim.x) Bs[(e >> 4) * PLD + (e & 15)] = Bin[e];
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  if (tid < 64 * nwaves) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r) {
      if constexpr (VAR >= 1) {
        const int li = tid & 15, lane = tid & 63, lk = lane >> 4;
        double a[4], w[4], myinv;
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = Bs[li * PLD + 4 * g + lk];
        if constexpr (VAR == 1) ldlt16s(a, w, myinv, li, lk); else ldlt16p(a, w, myinv, li, lk);
#pragma unroll
        for (int g = 0; g < 4; ++g) { Lm[li * PLD + 4 * g + lk] = a[g]; Li[li * PLD + 4 * g + lk] = w[g]; }
        if (lk == (li & 3)) dv[li] = myinv;
      } else {
        const int li = tid & 15, lane = tid & 63;
        double a[PIV], v[PIV], myinv;
#pragma unroll
        for (int j = 0; j < PIV; ++j) a[j] = Bs[(li >= j ? li : j) * PLD + (li >= j ? j : li)];
        ldlt16(a, v, myinv, li);
        if (lane < PIV) {
#pragma unroll
          for (int j = 0; j < PIV; ++j) Lm[li * PLD + j] = a[j];
          dv[li] = myinv;
        } else if (lane < 2 * PIV) {
#pragma unroll
          for (int j = 0; j < PIV; ++j) Li[li * PLD + j] = j == li ? 1.0 : v[j];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  }
  __syncthreads();
  if (tid == 0) cyc[0] = (t1 - t0) / reps;
  for (int e = tid; e < PIV * PIV; e += blockDim.x) { out[e] = Lm[(e >> 4) * PLD + (e & 15)]; out[256 + e] = Li[(e >> 4) * PLD + (e & 15)]; }
  if (tid < 16) out[512 + tid] = dv[tid];
}

int main() {
  std::vector<double> B(256);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) B[i * 16 + j] = (i == j ? 4.0 + i : 0.0) + 0.1 * ((i * 7 + j * 7) % 5 - 2) * (i != j);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < i; ++j) B[j * 16 + i] = B[i * 16 + j];
  double *dB, *dO; unsigned long long *dC;
  hipMalloc(&dB, 256 * 8); hipMalloc(&dO, 1024 * 8); hipMalloc(&dC, 8);
  hipMemcpy(dB, B.data(), 256 * 8, hipMemcpyHostToDevice);
  for (int var = 0; var < 3; ++var)
  for (int nw = 1; nw <= 2; ++nw) {
    hipMemset(dO, 0, 1024 * 8);
    if (var == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(128), 0, 0, dB, dO, dC, 200, nw);
    if (var == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(128), 0, 0, dB, dO, dC, 200, nw);
    if (var == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(128), 0, 0, dB, dO, dC, 200, nw);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    std::vector<double> O(1024); hipMemcpy(O.data(), dO, 1024 * 8, hipMemcpyDeviceToHost);
    // check: Linv B Linv^T == diag(1 / dinv)
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int p = 0; p < 16; ++p) for (int q = 0; q < 16; ++q) s += O[256 + i * 16 + p] * B[p * 16 + q] * O[256 + j * 16 + q];
      const double want = i == j ? 1.0 / O[512 + i] : 0.0;
      e1 = fmax(e1, fabs(s - want));
      if (j > i) e2 = fmax(e2, fabs(O[256 + i * 16 + j]));
    }
    printf("variant %d waves %d: %llu cycles per ldlt16 (err LDLt %.2e, Linv %.2e)\n", var, nw, c, e1, e2);
  }
  return 0;
}
