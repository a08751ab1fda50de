// Round 6: what bounds a one-pass row kernel over two N x D matrices (the C3 call's model_prior_maha_rows_kernel: 67 MB in
// 20.5 us = 3.3 TB/s)?  Variants of the access pattern with the same arithmetic shape (three row sums), timed back to back.
// build: hipcc -O3 --offload-arch=gfx950 tools/rows_probe.hip -o tools/rows_probe.bin ; run: tools/rows_probe.bin [n] [d]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double wsum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// A: a wave owns R rows, lanes stride the columns by 64 with 8-byte loads, STEPS column steps' loads in flight
template <int R, int STEPS>
__global__ void __launch_bounds__(256) rows8(const double* __restrict__ X, const double* __restrict__ E, long ld, long n, int d,
                                             const double* __restrict__ q0, const double* __restrict__ q1,
                                             double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= n) return;
  double a[R], b[R], c[R];
#pragma unroll
  for (int r = 0; r < R; ++r) a[r] = b[r] = c[r] = 0.0;
  for (int c0 = lane; c0 < d; c0 += 64 * STEPS) {
    double z[STEPS][R], e[STEPS][R], m[STEPS], iv[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int col = c0 + 64 * s < d ? c0 + 64 * s : d - 1;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const long row = row0 + r < n ? row0 + r : n - 1;
        z[s][r] = X[row * ld + col], e[s][r] = E[row * ld + col];
      }
      m[s] = q0[col], iv[s] = q1[col];
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (c0 + 64 * s >= d) break;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double dz = z[s][r] - m[s];
        a[r] -= 0.5 * dz * dz * iv[s];
        b[r] -= 0.5 * z[s][r] * z[s][r] * iv[s];
        c[r] = fma(e[s][r], e[s][r], c[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const double sa = wsum(a[r]), sb = wsum(b[r]), sc = wsum(c[r]);
    if (lane == 0 && row0 + r < n) out[row0 + r] = sa, out[n + row0 + r] = sb, out[2 * n + row0 + r] = sc;
  }
}

// B: the same with 16-byte loads (a lane owns the column pair 2 lane, 2 lane + 1 of each 128-column step)
template <int R, int STEPS>
__global__ void __launch_bounds__(256) rows16(const double* __restrict__ X, const double* __restrict__ E, long ld, long n, int d,
                                              const double* __restrict__ q0, const double* __restrict__ q1,
                                              double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= n) return;
  double a[R], b[R], c[R];
#pragma unroll
  for (int r = 0; r < R; ++r) a[r] = b[r] = c[r] = 0.0;
  for (int c0 = 2 * lane; c0 < d; c0 += 128 * STEPS) {
    d2 z[STEPS][R], e[STEPS][R], m[STEPS], iv[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int col = c0 + 128 * s < d ? c0 + 128 * s : d - 2;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const long row = row0 + r < n ? row0 + r : n - 1;
        z[s][r] = *reinterpret_cast<const d2*>(X + row * ld + col), e[s][r] = *reinterpret_cast<const d2*>(E + row * ld + col);
      }
      m[s] = *reinterpret_cast<const d2*>(q0 + col), iv[s] = *reinterpret_cast<const d2*>(q1 + col);
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (c0 + 128 * s >= d) break;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const d2 dz = z[s][r] - m[s];
        a[r] -= 0.5 * dz.x * dz.x * iv[s].x + 0.5 * dz.y * dz.y * iv[s].y;
        b[r] -= 0.5 * z[s][r].x * z[s][r].x * iv[s].x + 0.5 * z[s][r].y * z[s][r].y * iv[s].y;
        c[r] = fma(e[s][r].x, e[s][r].x, fma(e[s][r].y, e[s][r].y, c[r]));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const double sa = wsum(a[r]), sb = wsum(b[r]), sc = wsum(c[r]);
    if (lane == 0 && row0 + r < n) out[row0 + r] = sa, out[n + row0 + r] = sb, out[2 * n + row0 + r] = sc;
  }
}

// C: persistent-ish: gridDim.x workgroups walk the row groups (a wave takes R rows at a time, 8-byte loads, STEPS steps)
template <int R, int STEPS>
__global__ void __launch_bounds__(256) rows8_walk(const double* __restrict__ X, const double* __restrict__ E, long ld, long n, int d,
                                                  const double* __restrict__ q0, const double* __restrict__ q1,
                                                  double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long waves = (long)gridDim.x * 4;
  for (long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6); g * R < n; g += waves) {
    const long row0 = g * R;
    double a[R], b[R], c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = b[r] = c[r] = 0.0;
    for (int c0 = lane; c0 < d; c0 += 64 * STEPS) {
      double z[STEPS][R], e[STEPS][R], m[STEPS], iv[STEPS];
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int col = c0 + 64 * s < d ? c0 + 64 * s : d - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const long row = row0 + r < n ? row0 + r : n - 1;
          z[s][r] = X[row * ld + col], e[s][r] = E[row * ld + col];
        }
        m[s] = q0[col], iv[s] = q1[col];
      }
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        if (c0 + 64 * s >= d) break;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double dz = z[s][r] - m[s];
          a[r] -= 0.5 * dz * dz * iv[s];
          b[r] -= 0.5 * z[s][r] * z[s][r] * iv[s];
          c[r] = fma(e[s][r], e[s][r], c[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double sa = wsum(a[r]), sb = wsum(b[r]), sc = wsum(c[r]);
      if (lane == 0 && row0 + r < n) out[row0 + r] = sa, out[n + row0 + r] = sb, out[2 * n + row0 + r] = sc;
    }
  }
}

// a producer in front of every timed launch, as in the call (the sampling product has just written X)
__global__ void touch(double* X, long count, double v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) X[i] = v + 1e-9 * (double)(i & 1023);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// 20 launches between two events (a single launch between events measures the events); producer: a writer of X in front of every
// launch, as in the call (the sampling product has just written X) -- its own 20 launches are timed alone and subtracted
template <typename F>
static void timeit(const char* name, F launch, double bytes, double* X, long count, bool producer) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  std::vector<float> ts, tw;
  for (int rep = 0; rep < 15; ++rep) {
    float ms, msw = 0.f;
    if (producer) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, 0, X, count, 0.5);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&msw, e0, e1));
    }
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) {
      if (producer) hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, 0, X, count, 0.5);
      launch();
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back((ms - msw) / 20);
  }
  std::sort(ts.begin(), ts.end());
  printf("%-36s %s  median %6.2f us  min %6.2f us  %5.2f TB/s\n", name, producer ? "after a writer" : "back to back  ", 1e3 * ts[ts.size() / 2],
         1e3 * ts[0], bytes / (1e-3 * ts[ts.size() / 2]) / 1e12);
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 16384;
  const int d = argc > 2 ? atoi(argv[2]) : 256;
  const long ld = d, count = n * ld;
  double *X, *E, *q0, *q1, *out;
  CK(hipMalloc(&X, count * 8));
  CK(hipMalloc(&E, count * 8));
  CK(hipMalloc(&q0, d * 8));
  CK(hipMalloc(&q1, d * 8));
  CK(hipMalloc(&out, 3 * n * 8));
  hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, 0, X, count, 0.5);
  hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, 0, E, count, 0.25);
  hipLaunchKernelGGL(touch, dim3(1), dim3(256), 0, 0, q0, (long)d, 0.1);
  hipLaunchKernelGGL(touch, dim3(1), dim3(256), 0, 0, q1, (long)d, 1.0);
  CK(hipDeviceSynchronize());
  const double bytes = 2.0 * count * 8;
#define RUN(NAME, KERNEL, GRID)                                                                                              \
  for (int prod = 0; prod < 2; ++prod)                                                                                       \
    timeit(NAME, [&] { hipLaunchKernelGGL(KERNEL, dim3((unsigned)(GRID)), dim3(256), 0, 0, X, E, ld, n, d, q0, q1, out); }, bytes, X, \
           count, prod == 1);
  RUN("8 B, 4 rows/wave, 1 step", (rows8<4, 1>), (n + 15) / 16)
  RUN("8 B, 4 rows/wave, 2 steps", (rows8<4, 2>), (n + 15) / 16)
  RUN("8 B, 4 rows/wave, 4 steps", (rows8<4, 4>), (n + 15) / 16)
  RUN("8 B, 2 rows/wave, 4 steps", (rows8<2, 4>), (n + 7) / 8)
  RUN("8 B, 1 row/wave, 4 steps", (rows8<1, 4>), (n + 3) / 4)
  RUN("8 B, 8 rows/wave, 1 step", (rows8<8, 1>), (n + 31) / 32)
  RUN("16 B, 4 rows/wave, 1 step", (rows16<4, 1>), (n + 15) / 16)
  RUN("16 B, 4 rows/wave, 2 steps", (rows16<4, 2>), (n + 15) / 16)
  RUN("16 B, 2 rows/wave, 2 steps", (rows16<2, 2>), (n + 7) / 8)
  RUN("16 B, 8 rows/wave, 1 step", (rows16<8, 1>), (n + 31) / 32)
  RUN("16 B, 8 rows/wave, 2 steps", (rows16<8, 2>), (n + 31) / 32)
  RUN("8 B walk 512 wgs, 4 rows, 2 steps", (rows8_walk<4, 2>), 512)
  RUN("8 B walk 256 wgs, 4 rows, 2 steps", (rows8_walk<4, 2>), 256)
  RUN("8 B walk 1024 wgs, 2 rows, 4 steps", (rows8_walk<2, 4>), 1024)
  RUN("8 B walk 512 wgs, 2 rows, 4 steps", (rows8_walk<2, 4>), 512)
  return 0;
}
