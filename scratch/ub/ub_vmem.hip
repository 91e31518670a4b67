// issue cost of vector-memory instructions for one wave / eight waves (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2_t __attribute__((ext_vector_type(2)));
__global__ void k(const double *src, double *dst, unsigned long long *cyc) {
  const int tid = threadIdx.x;
  unsigned long long t0, t1, t2;
  const double *p = src + (size_t)blockIdx.x * 65536 + tid;
  double v[32];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = p[i * 512];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += v[i];
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2), "+v"(s)::"memory");
  if (tid == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t0; }
  // dwordx4
  const d2_t *p4 = (const d2_t *)(src + (size_t)blockIdx.x * 65536 + 32768) + tid;
  d2_t w[16];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = p4[i * 512];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
#pragma unroll
  for (int i = 0; i < 16; ++i) s += w[i][0] + w[i][1];
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2), "+v"(s)::"memory");
  if (tid == 0 && blockIdx.x == 0) { cyc[2] = t1 - t0; cyc[3] = t2 - t0; }
  // stores
  double *q = dst + (size_t)blockIdx.x * 65536 + tid;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
  for (int i = 0; i < 32; ++i) q[i * 512] = s + i;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
  if (tid == 0 && blockIdx.x == 0) { cyc[4] = t1 - t0; cyc[5] = t2 - t0; }
}
int main() {
  double *src, *dst; unsigned long long *cyc;
  (void)hipMalloc(&src, 8ull * 65536 * 256); (void)hipMalloc(&dst, 8ull * 65536 * 256); (void)hipMalloc(&cyc, 64);
  (void)hipMemset(src, 0, 8ull * 65536 * 256);
  for (int nb : {1, 256}) for (int nt : {64, 512}) {
    hipLaunchKernelGGL(k, dim3(nb), dim3(nt), 0, 0, src, dst, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("blocks %3d threads %3d: 32 x load b64: issue %5llu total %5llu | 16 x load b128: issue %5llu total %5llu | 32 x store b64: issue %5llu total %5llu\n", nb, nt, c[0], c[1], c[2], c[3], c[4], c[5]);
  }
  return 0;
}
