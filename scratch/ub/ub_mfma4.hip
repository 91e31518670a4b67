// f64 MFMA throughput of ONE SIMD with 1..4 waves issuing (waves w, w+4, w+8, w+12 of a 1024-thread block share a SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int NIND>
__global__ void k(double *io, unsigned long long *cyc, unsigned mask) {
  const int tid = threadIdx.x, wv = tid >> 6;
  double b = io[tid + 1], c = io[tid + 2];
  d4_t acc[NIND];
  for (int i = 0; i < NIND; ++i) acc[i] = d4_t{b, c, b, c};
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if ((mask >> wv) & 1u) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(b)::"memory");
#pragma unroll 4
    for (int i = 0; i < 256 / NIND; ++i)
#pragma unroll
      for (int j = 0; j < NIND; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc[j], 0, 0, 0);
    for (int j = 0; j < NIND; ++j) b += acc[j][0];
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(b)::"memory");
  }
  if (blockIdx.x == gridDim.x - 1) cyc[wv] = t1 - t0;
  if (b == 123.456) io[tid] = b;
}
template <int NIND>
void run(double *io, unsigned long long *cyc, int grid) {
  for (unsigned mask : {0x0001u, 0x0011u, 0x0111u, 0x1111u, 0x000fu, 0x00ffu, 0xffffu}) {
    hipLaunchKernelGGL(k<NIND>, dim3(grid), dim3(1024), 0, 0, io, cyc, mask);
    hipLaunchKernelGGL(k<NIND>, dim3(grid), dim3(1024), 0, 0, io, cyc, mask);
    (void)hipDeviceSynchronize();
    unsigned long long c[16]; (void)hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
    printf("grid %d, %d independent accumulators, 256 MFMA per wave, waves %04x:", grid, NIND, mask);
    for (int w = 0; w < 16; ++w) if ((mask >> w) & 1u) printf(" %6llu", c[w]);
    printf("\n");
  }
}
int main() {
  double *io; unsigned long long *cyc;
  (void)hipMalloc(&io, 8 * 2048); (void)hipMalloc(&cyc, 128);
  (void)hipMemset(io, 0, 8 * 2048);
  for (int grid : {1, 256}) { run<1>(io, cyc, grid); run<2>(io, cyc, grid); }
  return 0;
}
