// Dev tool: what does an s_barrier cost the fp64 matrix pipe?  Waves issue NMF independent
// v_mfma_f64_4x4x4_4b_f64 (32 accumulators, as the 128 x 64 GEMM tile's wave) between barriers; the
// kernel is run at 1-4 workgroups per CU with 4- or 8-wave workgroups.  Prints shader cycles per MFMA per
// SIMD and the implied TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_barrier_probe.hip -o tools/mfma_barrier_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

// BAR: 0 none, 1 bare s_barrier, 2 __syncthreads
template <int NMF, int BAR, int THREADS>
__global__ void __launch_bounds__(THREADS) probe(long long* out, double* sink, int rounds) {
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  const double a = threadIdx.x * 1e-3 + 0.25, b = blockIdx.x * 1e-3 + 1.0;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < NMF; ++i) acc[i & 31] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i & 31], 0, 0, 0);
    if (BAR == 1) {
      __asm__ volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __asm__ volatile("" ::: "memory");
    } else if (BAR == 2) {
      __syncthreads();
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
  sink[(size_t)blockIdx.x * THREADS + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = w1 - w0;
  }
}

template <int NMF, int BAR, int THREADS>
static void run(int n_cu, long long* d, double* sink) {
  const int total_mf = 1 << 18;
  const int rounds = total_mf / NMF;
  const int step = 256 / THREADS > 0 ? 256 / THREADS : 1;
  for (int wg = step; wg * THREADS <= 6 * 256; wg += step) {
    hipLaunchKernelGGL((probe<NMF, BAR, THREADS>), dim3(n_cu * wg), dim3(THREADS), 0, 0, d, sink, rounds);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<NMF, BAR, THREADS>), dim3(n_cu * wg), dim3(THREADS), 0, 0, d, sink, rounds);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NMF, BAR, THREADS>), dim3(n_cu * wg), dim3(THREADS), 0, 0, d, sink, rounds);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> o(n_cu * wg * 2);
    hipMemcpy(o.data(), d, o.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < n_cu * wg; ++i) cyc += o[2 * i], wall += o[2 * i + 1];
    cyc /= n_cu * wg, wall /= n_cu * wg;
    const double us = wall / 100.0;
    const int waves_per_simd = wg * THREADS / 256;
    const double mf = (double)NMF * rounds;
    printf("nmf %4d bar %d threads %d wg/cu %d (waves/simd %d): %.0f MHz, %6.2f cyc per MFMA per SIMD, %6.1f TFLOP/s (per-WG clocks); "
           "kernel %.1f us -> %6.1f TFLOP/s (makespan)\n", NMF,
           BAR, THREADS, wg, waves_per_simd, cyc / us, cyc / mf / (waves_per_simd > 0 ? waves_per_simd : 1),
           512.0 * mf * (THREADS / 64) * n_cu * wg / us / 1e6, ms * 1e3, 512.0 * mf * (THREADS / 64) * n_cu * wg / (ms * 1e3) / 1e6);
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  long long* d;
  double* sink;
  hipMalloc(&d, n_cu * 16 * 2 * sizeof(long long));
  hipMalloc(&sink, (size_t)n_cu * 16 * 512 * sizeof(double));
  run<128, 0, 256>(n_cu, d, sink);
  run<64, 1, 256>(n_cu, d, sink);
  run<128, 1, 256>(n_cu, d, sink);
  run<128, 0, 512>(n_cu, d, sink);
  run<128, 1, 512>(n_cu, d, sink);
  run<128, 1, 128>(n_cu, d, sink);
  run<128, 1, 64>(n_cu, d, sink);
  return 0;
}
