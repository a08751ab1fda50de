// Dev tool: accuracy of the lean log / sincos of vb_rng.h against libm on random arguments, and their throughput
// next to OCML's log / sincospi.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iviabel_amd/csrc tools/rng_math_check.hip -o tools/rng_math_check.bin
#include "vb_rng.h"

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

using namespace vb;

__global__ void eval(const double* u, double* lg, double* sn, double* cs, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  lg[i] = vb_log_unit(u[i]);
  vb_sincos_turn(u[i], &sn[i], &cs[i]);
}

template <int MODE>
__global__ void speed(double* out, int iters) {
  double u = 0.123 + 1e-4 * threadIdx.x, acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    double s, c, l;
    if (MODE == 0) {
      l = vb_log_unit(u);
      vb_sincos_turn(u, &s, &c);
    } else {
      l = log(u);
      sincospi(2.0 * u, &s, &c);
    }
    acc += l + s * c;
    u = u * 0.999 + 1e-4;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const int n = 1 << 22;
  std::vector<double> h(n);
  std::mt19937_64 g(7);
  for (int i = 0; i < n; ++i) {
    const unsigned long long x = g();
    h[i] = ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    if (i % 8 == 1) h[i] = ldexp(h[i], -(int)(x % 60));        // small arguments
    if (i % 8 == 2) h[i] = 1.0 - ldexp(h[i], -(int)(x % 50));  // arguments next to 1
    if (h[i] <= 0.0 || h[i] >= 1.0) h[i] = 0.5;
  }
  double *u, *lg, *sn, *cs;
  hipMalloc(&u, n * 8); hipMalloc(&lg, n * 8); hipMalloc(&sn, n * 8); hipMalloc(&cs, n * 8);
  hipMemcpy(u, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(eval, dim3(n / 256), dim3(256), 0, 0, u, lg, sn, cs, n);
  std::vector<double> l(n), s(n), c(n);
  hipMemcpy(l.data(), lg, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(s.data(), sn, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), cs, n * 8, hipMemcpyDeviceToHost);
  double el = 0, es = 0, ec = 0;
  for (int i = 0; i < n; ++i) {
    const long double lr = logl((long double)h[i]);
    el = fmax(el, (double)fabsl(((long double)l[i] - lr) / lr));
    const long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)h[i];
    es = fmax(es, (double)fabsl((long double)s[i] - sinl(a)));
    ec = fmax(ec, (double)fabsl((long double)c[i] - cosl(a)));
  }
  printf("max relative error of log: %.3e; max absolute error of sin: %.3e, cos: %.3e  (%d arguments)\n", el, es, ec, n);
  double* out;
  hipMalloc(&out, 1024 * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(speed<0>, dim3(1024), dim3(256), 0, 0, out, 2000);
      else hipLaunchKernelGGL(speed<1>, dim3(1024), dim3(256), 0, 0, out, 2000);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%s: %.1f us for 1024 x 256 threads x 2000 (log + sincos) pairs\n", mode == 0 ? "lean" : "OCML", ms * 1e3);
  }
  return 0;
}
