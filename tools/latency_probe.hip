// A small kernel reads what the previous kernel wrote (8 doubles per thread), waits for them, adds them up.
// s_memrealtime stamps (100 MHz): entry, loads issued, loads returned (s_waitcnt vmcnt(0)), sum done.
// build: hipcc -O3 --offload-arch=gfx950 tools/latency_probe.hip -o tools/latency_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void writer(double* p, int n, double v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v + i;
}

__global__ void __launch_bounds__(256) reader(const double* p, double* out, long long* st, int stride, int mode) {
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  long long t0 = wall_clock64();
  const double* base = p + (size_t)blockIdx.x * 16 * stride + q * stride + c;
  double v[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = base[4 * i * stride];
    v[4 + i] = base[4 * i * stride + 64];
  }
  long long t1 = wall_clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t2 = wall_clock64();
  double s0 = 0.0, s1 = 0.0;
  if (mode == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s0 += v[i], s1 += v[4 + i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) s0 += v[i];
    asm volatile("" : "+v"(s0));
#pragma unroll
    for (int i = 0; i < 4; ++i) s1 += v[4 + i];
  }
  asm volatile("" : "+v"(s0), "+v"(s1));
  long long t3 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = s0 + s1;
  long long t4 = wall_clock64();
  if (threadIdx.x == 0) {
    long long* s = st + blockIdx.x * 8;
    s[0] = t1 - t0, s[1] = t2 - t1, s[2] = t3 - t2, s[3] = t4 - t3;
  }
}

int main() {
  const int stride = 448, nblk = 16;
  const int n = nblk * 16 * stride + 4096;
  double *p, *out;
  long long* st;
  hipMalloc(&p, n * sizeof(double));
  hipMalloc(&out, nblk * 256 * sizeof(double));
  hipMalloc(&st, 4096);
  for (int rep = 0; rep < 6; ++rep) {
    hipLaunchKernelGGL(writer, dim3((n + 255) / 256), dim3(256), 0, 0, p, n, (double)rep);
    hipLaunchKernelGGL(reader, dim3(nblk), dim3(256), 0, 0, p, out, st, stride, rep & 1);
    hipDeviceSynchronize();
    long long h[16 * 8];
    hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
    printf("launch %d (mode %d) block 0: issue %lld  wait %lld  sum %lld  store-issue %lld | block 5: issue %lld wait %lld sum %lld store %lld  (10 ns ticks)\n",
           rep, rep & 1, h[0], h[1], h[2], h[3], h[40], h[41], h[42], h[43]);
  }
  return 0;
}
