// Probe: lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (operands and result, 4 blocks of 4x4x4).
// For every source lane s: one-hot A (B = ones) and one-hot B (A = ones) show which result lanes it feeds;
// one-hot A at s and one-hot B at t shows which (s, t) share block and k.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(int mode, int s, int t, double* out) {
  const int l = threadIdx.x;
  double a = 1.0, b = 1.0;
  if (mode == 0) a = (l == s) ? 1.0 : 0.0;
  if (mode == 1) b = (l == s) ? 1.0 : 0.0;
  if (mode == 2) { a = (l == s) ? 1.0 : 0.0; b = (l == t) ? 1.0 : 0.0; }
  double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[l] = c;
}

int main() {
  double* d; hipMalloc(&d, 64 * sizeof(double));
  std::vector<double> h(64);
  auto run = [&](int mode, int s, int t) {
    probe<<<1, 64>>>(mode, s, t, d); hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
  };
  printf("A lane s -> result lanes it reaches (B = 1):\n");
  for (int s = 0; s < 64; ++s) { run(0, s, 0); printf("A%02d:", s); for (int l = 0; l < 64; ++l) if (h[l] != 0) printf(" %d", l); printf("\n"); }
  printf("B lane s -> result lanes it reaches (A = 1):\n");
  for (int s = 0; s < 64; ++s) { run(1, s, 0); printf("B%02d:", s); for (int l = 0; l < 64; ++l) if (h[l] != 0) printf(" %d", l); printf("\n"); }
  printf("A lane s pairs with B lanes t (same block and k), result lane:\n");
  for (int s = 0; s < 64; s += 1) { printf("A%02d:", s); for (int t = 0; t < 64; ++t) { run(2, s, t); for (int l = 0; l < 64; ++l) if (h[l] != 0) printf(" B%d->D%d", t, l); } printf("\n"); if (s == 20) break; }
  return 0;
}
