// Dev tool: where do the waves of a workgroup land (CU / SIMD), and what MFMA rate does each wave get?
// Every wave issues the same stream of independent v_mfma_f64_4x4x4_4b_f64 and records HW_ID and its
// own elapsed shader cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_place_probe.hip -o tools/mfma_place_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <vector>

template <int NMF, int BAR, int THREADS>
__global__ void __launch_bounds__(THREADS) probe(long long* out, double* sink, int rounds) {
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  const double a = threadIdx.x * 1e-3 + 0.25, b = blockIdx.x * 1e-3 + 1.0;
  const long long c0 = clock64();
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < NMF; ++i) acc[i & 31] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i & 31], 0, 0, 0);
    if (BAR == 1) {
      __asm__ volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __asm__ volatile("" ::: "memory");
    }
  }
  const long long c1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
  sink[(size_t)blockIdx.x * THREADS + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, offset 0, size 32
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
    const size_t w = (size_t)blockIdx.x * (THREADS / 64) + threadIdx.x / 64;
    out[4 * w] = c1 - c0;
    out[4 * w + 1] = hw;
    out[4 * w + 2] = xcc;
    out[4 * w + 3] = c0;
  }
}

template <int NMF, int BAR, int THREADS>
static void run(int n_cu, int wg, long long* d, double* sink, bool dump) {
  const int rounds = (1 << 16) / NMF;
  const int nw = n_cu * wg * (THREADS / 64);
  hipLaunchKernelGGL((probe<NMF, BAR, THREADS>), dim3(n_cu * wg), dim3(THREADS), 0, 0, d, sink, rounds);
  hipDeviceSynchronize();
  std::vector<long long> o((size_t)nw * 4);
  hipMemcpy(o.data(), d, o.size() * sizeof(long long), hipMemcpyDeviceToHost);
  // per (xcc, se, sh, cu): number of waves per SIMD; histogram of SIMD loads and mean per-wave period by SIMD load
  std::map<long long, std::vector<int>> cu_waves;
  for (int w = 0; w < nw; ++w) {
    const unsigned hw = (unsigned)o[4 * w + 1];
    const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const long long key = ((o[4 * w + 2] & 15) << 16) | (se << 8) | (sh << 4) | cu;
    auto& v = cu_waves[key];
    if (v.empty()) v.assign(4, 0);
    v[simd]++;
  }
  std::map<int, int> hist_cu;        // waves per CU -> count
  std::map<int, int> hist_simd;      // waves per SIMD -> count
  for (auto& kv : cu_waves) {
    hist_cu[kv.second[0] + kv.second[1] + kv.second[2] + kv.second[3]]++;
    for (int s = 0; s < 4; ++s) hist_simd[kv.second[s]]++;
  }
  std::map<int, std::pair<double, int>> period;   // SIMD load -> (sum of periods, waves)
  for (int w = 0; w < nw; ++w) {
    const unsigned hw = (unsigned)o[4 * w + 1];
    const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const long long key = ((o[4 * w + 2] & 15) << 16) | (se << 8) | (sh << 4) | cu;
    const int load = cu_waves[key][simd];
    period[load].first += (double)o[4 * w] / ((double)NMF * rounds);
    period[load].second++;
  }
  printf("nmf %d bar %d threads %d wg/cu %d: distinct CUs %zu;", NMF, BAR, THREADS, wg, cu_waves.size());
  printf(" waves/CU:");
  for (auto& kv : hist_cu) printf(" %d x%d", kv.first, kv.second);
  printf("; waves/SIMD:");
  for (auto& kv : hist_simd) printf(" %d x%d", kv.first, kv.second);
  printf("; cycles per own MFMA by SIMD load:");
  for (auto& kv : period) printf(" [%d] %.1f", kv.first, kv.second.first / kv.second.second);
  printf("\n");
  if (dump) {
    for (int b = 0; b < 4; ++b) {
      printf("  block %d:", b);
      for (int k = 0; k < THREADS / 64; ++k) {
        const size_t w = (size_t)b * (THREADS / 64) + k;
        const unsigned hw = (unsigned)o[4 * w + 1];
        printf(" (xcc %lld se %u sh %u cu %u simd %u wave %u: %.1f)", o[4 * w + 2] & 15, (hw >> 13) & 7, (hw >> 12) & 1,
               (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, (double)o[4 * w] / ((double)NMF * rounds));
      }
      printf("\n");
    }
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  long long* d;
  double* sink;
  hipMalloc(&d, (size_t)n_cu * 8 * 8 * 4 * sizeof(long long));
  hipMalloc(&sink, (size_t)n_cu * 8 * 512 * sizeof(double));
  run<128, 0, 256>(n_cu, 1, d, sink, true);
  run<128, 0, 256>(n_cu, 2, d, sink, true);
  run<128, 0, 256>(n_cu, 3, d, sink, false);
  run<128, 0, 256>(n_cu, 4, d, sink, false);
  run<128, 1, 256>(n_cu, 2, d, sink, false);
  run<128, 0, 512>(n_cu, 1, d, sink, true);
  run<128, 1, 512>(n_cu, 1, d, sink, true);
  run<128, 0, 512>(n_cu, 2, d, sink, false);
  run<128, 1, 512>(n_cu, 2, d, sink, false);
  run<128, 0, 128>(n_cu, 2, d, sink, true);
  run<128, 0, 128>(n_cu, 4, d, sink, false);
  run<128, 0, 64>(n_cu, 4, d, sink, true);
  run<128, 0, 64>(n_cu, 8, d, sink, false);
  return 0;
}
