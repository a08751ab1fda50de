// Dev tool: times the fp64 MFMA GEMM building block (viabel_amd/csrc/vb_gemm_f64.h) by itself and
// measures the effective shader clock while it runs (clock64 = shader cycles, wall_clock64 = 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iviabel_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench.bin
// Run:   tools/gemm_bench.bin [M N K]
#include "vb_gemm_f64.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

using namespace vb;

struct EpiStore {
  double* C;
  int64_t ldc;
  __device__ void operator()(int, int row, int col, double acc) const { C[(int64_t)row * ldc + col] = acc; }
};

// store variants for the end-of-kernel write-back experiment (GEMM_STORE=nt|wt)
struct EpiStoreNT {
  double* C;
  int64_t ldc;
  __device__ void operator()(int, int row, int col, double acc) const { __builtin_nontemporal_store(acc, &C[(int64_t)row * ldc + col]); }
};
struct EpiStoreWT {
  double* C;
  int64_t ldc;
  __device__ void operator()(int, int row, int col, double acc) const {
    __hip_atomic_store(&C[(int64_t)row * ldc + col], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
// tri_mode 1's contract: B[k][j] == 0 for k > j (the kernel may skip what the contract says is zero)
__global__ void make_upper_kernel(const double* src, double* dst, int64_t ld, int K, int N) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)K * ld) return;
  const int k = (int)(i / ld), j = (int)(i % ld);
  dst[i] = (j < N && k > j) ? 0.0 : src[i];
}

struct EpiSlab {
  double* C;
  int64_t ldc, slab;
  __device__ void operator()(int split, int row, int col, double acc) const {
    C[split * slab + (int64_t)row * ldc + col] = acc;
  }
};

__global__ void clock_probe(long long* out, int iters) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = w1 - w0;
  }
  if (s == 12345.678) out[0] = 0;
}

// Build-up probes: the GEMM's inner loop with its parts switched on one at a time.
//   LEVEL 0: 64 MFMAs per k-step on register operands (4 A x 16 B fragments, as the 128 x 128 tile loop)
//   LEVEL 1: + the 20 ds_read_b64 fragment loads per k-step (no barriers, LDS content constant)
//   LEVEL 2: + one __syncthreads per 4 k-steps
//   LEVEL 3: + the ds_write of a staged slab (register -> LDS) per 4 k-steps
template <int LEVEL, int AF, int NB>
__global__ void __launch_bounds__(256, 2) loop_probe(long long* out, double* sink, int slabs) {
  __shared__ double As[2][16][32 * AF + 16];
  __shared__ double Bs[2][16][8 * NB + 16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = t; i < 2 * 16 * (32 * AF + 16); i += 256) (&As[0][0][0])[i] = 1e-3 * i;
  for (int i = t; i < 2 * 16 * (8 * NB + 16); i += 256) (&Bs[0][0][0])[i] = 1e-3 * i + 1;
  __syncthreads();
  const int fi = lane & 15, fk = lane >> 4, fblk = (lane >> 2) & 3, fj = lane & 3;
  double acc[AF][NB];
  for (int a = 0; a < AF; ++a)
    for (int r = 0; r < NB; ++r) acc[a][r] = 0;
  double af[AF], bf[NB];
  for (int a = 0; a < AF; ++a) af[a] = 1e-3 * (t + a);
  for (int r = 0; r < NB; ++r) bf[r] = 1e-3 * (t - r);
  d2v stage[6];
  for (int i = 0; i < 6; ++i) stage[i] = (d2v){1e-3 * t, 2e-3 * i};
  const long long c0 = clock64(), w0 = wall_clock64();
  int buf = 0;
  for (int s = 0; s < slabs; ++s) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (LEVEL >= 1) {
#pragma unroll
        for (int a = 0; a < AF; ++a) af[a] = As[buf][4 * kk + fk][wm * (16 * AF) + a * 16 + fi];
#pragma unroll
        for (int r = 0; r < NB; ++r) bf[r] = Bs[buf][4 * kk + fk][wn * (4 * NB) + 4 * ((fblk + r) & (NB - 1)) + fj];
      }
#pragma unroll
      for (int a = 0; a < AF; ++a)
#pragma unroll
        for (int r = 0; r < NB; ++r) acc[a][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[a], bf[r], acc[a][r], 0, 0, 0);
    }
    if (LEVEL >= 3) {
#pragma unroll
      for (int i = 0; i < (AF + NB / 4); ++i) {
        const int p = i * 256 + t;
        if (i < NB / 4) *reinterpret_cast<d2v*>(&Bs[buf ^ 1][p / (4 * NB)][(p % (4 * NB)) * 2]) = stage[i];
        else *reinterpret_cast<d2v*>(&As[buf ^ 1][(p - NB / 4 * 256) / (16 * AF)][((p - NB / 4 * 256) % (16 * AF)) * 2]) = stage[i];
      }
    }
    if (LEVEL >= 2) {
      __syncthreads();
      buf ^= 1;
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double sum = 0;
  for (int a = 0; a < AF; ++a)
    for (int r = 0; r < NB; ++r) sum += acc[a][r];
  sink[blockIdx.x * 256 + t] = sum;
  if (t == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = w1 - w0;
  }
}

template <int LEVEL, int AF, int NB>
static void run_loop_probe(hipStream_t st, int n_cu, long long* d, double* sink) {
  for (int wg = 1; wg <= 2; ++wg) {
    const int slabs = 2000;
    hipLaunchKernelGGL((loop_probe<LEVEL, AF, NB>), dim3(n_cu * wg), dim3(256), 0, st, d, sink, slabs);
    hipDeviceSynchronize();
    std::vector<long long> o(n_cu * wg * 2);
    hipMemcpy(o.data(), d, o.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < n_cu * wg; ++i) cyc += o[2 * i], wall += o[2 * i + 1];
    cyc /= n_cu * wg, wall /= n_cu * wg;
    const double us = wall / 100.0, mf = 4.0 * AF * NB * slabs;
    printf("loop_probe level %d tile %dx%d wg/cu=%d: %.0f MHz, %.2f cycles per MFMA per SIMD, %.1f TFLOP/s\n", LEVEL, 32 * AF,
           8 * NB, wg, cyc / us, cyc / mf / wg, 512.0 * mf * 4 * n_cu * wg / us / 1e6);
  }
}

static hipStream_t g_time_stream = nullptr;      // the stream the timed lambdas launch on (for GEMM_GRAPH)
template <class F>
static float time_it(F&& f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  // GEMM_GRAPH=n: the same launches as a captured graph of n kernel nodes, replayed reps / n times -- what is the
  // launch-to-launch cost of dependent kernels inside a graph, against back-to-back stream launches?
  const int gn = getenv("GEMM_GRAPH") ? atoi(getenv("GEMM_GRAPH")) : 0;
  if (gn > 0 && g_time_stream) {
    hipGraph_t graph;
    hipGraphExec_t exec;
    if (hipStreamBeginCapture(g_time_stream, hipStreamCaptureModeGlobal) != hipSuccess) printf("capture failed\n");
    for (int i = 0; i < gn; ++i) f();
    if (hipStreamEndCapture(g_time_stream, &graph) != hipSuccess) printf("end capture failed\n");
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) printf("instantiate failed\n");
    const int launches = reps / gn > 0 ? reps / gn : 1;
    for (int i = 0; i < 3; ++i) hipGraphLaunch(exec, g_time_stream);
    hipDeviceSynchronize();
    hipEventRecord(e0, g_time_stream);
    for (int i = 0; i < launches; ++i) hipGraphLaunch(exec, g_time_stream);
    hipEventRecord(e1, g_time_stream);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(exec);
    hipGraphDestroy(graph);
    return ms / (launches * gn);
  }
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", p.name, n_cu, p.clockRate);
  const size_t big = (size_t)4096 * 4096;
  double *A, *B, *C;
  hipMalloc(&A, big * 8);
  hipMalloc(&B, big * 8);
  hipMalloc(&C, big * 8 * 2);
  std::vector<double> h(big);
  for (size_t i = 0; i < big; ++i) h[i] = (double)((i * 2654435761u) % 1000) * 1e-3 - 0.5;
  hipMemcpy(A, h.data(), big * 8, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), big * 8, hipMemcpyHostToDevice);
  hipStream_t st;
  hipStreamCreate(&st);
  g_time_stream = st;

  // GEMM_REPS: timed launches per measurement in profiling mode (default 20; thousands = seconds of sustained load,
  // for power / clock sampling from the host with rocm-smi)
  const int kReps = getenv("GEMM_REPS") ? atoi(getenv("GEMM_REPS")) : 20;
  if (argc > 4) {   // profiling mode: only the dense GEMM in tile configuration argv[4]
    const int cfg = atoi(argv[4]);
    if (argc > 5 && argv[5][0] == 'n') {   // 'n<bits>[a]': standard normals, mantissa truncated to <bits> bits (0 = keep all 52);
      // suffix 'a': only operand A is truncated.  Operand-dependent power is what sets the sustained clock.
      const int bits = atoi(argv[5] + 1);
      const bool only_a = argv[5][strlen(argv[5]) - 1] == 'a';
      srand48(7);
      std::vector<double> hb(big);
      auto trunc = [&](double x) {
        if (bits <= 0 || bits >= 52) return x;
        unsigned long long u;
        memcpy(&u, &x, 8);
        u &= ~((1ull << (52 - bits)) - 1);
        memcpy(&x, &u, 8);
        return x;
      };
      for (size_t i = 0; i < big; i += 2) {
        const double u1 = drand48() + 1e-300, u2 = drand48(), r = sqrt(-2.0 * log(u1));
        const double a = r * cos(6.283185307179586 * u2), b = r * sin(6.283185307179586 * u2);
        h[i] = trunc(a), h[i + 1] = trunc(b);
        hb[i] = only_a ? b : trunc(b), hb[i + 1] = only_a ? a : trunc(a);
      }
      hipMemcpy(A, h.data(), big * 8, hipMemcpyHostToDevice);
      hipMemcpy(B, hb.data(), big * 8, hipMemcpyHostToDevice);
    } else if (argc > 5 && argv[5][0] != 'r') {   // data pattern: 'z' all zero, 'o' all ones, 'r' keep the pseudo-random fill
      const double v = argv[5][0] == 'z' ? 0.0 : 1.0;
      for (size_t i = 0; i < big; ++i) h[i] = v;
      hipMemcpy(A, h.data(), big * 8, hipMemcpyHostToDevice);
      hipMemcpy(B, h.data(), big * 8, hipMemcpyHostToDevice);
    }
    // argv[6]: 'd' dense (default), 't' triangular k range (Z = E L'), 'g' Gram C = A' B with lower-triangular
    // tiles and argv[7] row slabs (M = N = D, K = rows)
    const char mode = argc > 6 ? argv[6][0] : 'd';
    const int splits = argc > 7 ? atoi(argv[7]) : 1;
    GemmArgs g;
    g.A = A, g.B = B, g.lda = K, g.ldb = N, g.M = M, g.N = N, g.K = K, g.tri_mode = mode == 't' ? 1 : 0;
#ifdef VB_GEMM_CLOCK
    if (getenv("GEMM_BN_MIN")) g.dbg_bn_min = atoi(getenv("GEMM_BN_MIN"));
    if (getenv("GEMM_BN_MAX")) g.dbg_bn_max = atoi(getenv("GEMM_BN_MAX"));
    if (getenv("GEMM_PRIO_SLABS")) g.dbg_prio_slabs = atoi(getenv("GEMM_PRIO_SLABS"));
    if (g.dbg_prio_slabs) printf("PROBE: wave priority 3 for tiles of %s %d k slabs\n", g.dbg_prio_slabs > 0 ? "at most" : "more than", abs(g.dbg_prio_slabs));
    if (g.dbg_bn_min > 0 || g.dbg_bn_max < (1 << 30))
      printf("PROBE: only the column blocks %d .. %d of the triangle do work (the others leave at once)\n", g.dbg_bn_min, g.dbg_bn_max);
#endif
    float ms;
    long nb_mode = 0;
    int slabs_per_wg = K / 16;
    if (mode == 'g' || mode == 'G' || mode == 'D') {
      // 'G': the same Gram product over all tiles (no triangle); 'D': the same shape and row slabs with A given
      // as A[m][k] (what separates "k-major A" from "split-K geometry")
      GemmArgs g3;
      // GEMM_PAD_A / GEMM_PAD_B: extra doubles in the row stride of the k-major operands (power-of-two strides put
      // every k row of a tile on the same L2 channels)
      const int pad_a = getenv("GEMM_PAD_A") ? atoi(getenv("GEMM_PAD_A")) : 0, pad_b = getenv("GEMM_PAD_B") ? atoi(getenv("GEMM_PAD_B")) : 0;
      g3.A = A, g3.B = B, g3.lda = mode == 'D' ? M : N + pad_a, g3.ldb = N + pad_b, g3.M = N, g3.N = N, g3.K = M, g3.tri_mode = mode == 'g' ? 2 : 0;
      if (mode == 'D')
        ms = time_it([&] { gemm_f64_launch<true>(st, g3, splits, n_cu, EpiSlab{C, N, (int64_t)N * N}, cfg); }, kReps);
      else
        ms = time_it([&] { gemm_f64_launch<false>(st, g3, splits, n_cu, EpiSlab{C, N, (int64_t)N * N}, cfg); }, kReps);
      nb_mode = gemm_count_blocks(g3, (cfg == 3 || cfg == 4) ? 64 : 128, cfg == 1 ? 128 : 64);
      slabs_per_wg = ((M + splits - 1) / splits + 15) / 16;
      printf("cfg %d gram A[k][m] D=%d rows=%d splits=%d (%ld tiles x %d): %.1f us  %.2f TFLOP/s (dense convention)\n", cfg, N, M,
             splits, nb_mode, splits, ms * 1e3, 2.0 * M * N * N / ms / 1e9);
    } else {
      const char* sv = getenv("GEMM_STORE");
      {
        const float e = time_it([&] { hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(256), 0, st, (int*)nullptr); }, 200);
        printf("empty 512-workgroup kernel back to back: %.2f us per launch\n", e * 1e3);
      }
      if (sv && sv[0] == 'n') ms = time_it([&] { gemm_f64_launch<true>(st, g, 1, n_cu, EpiStoreNT{C, N}, cfg); }, kReps);
      else if (sv && sv[0] == 'w') ms = time_it([&] { gemm_f64_launch<true>(st, g, 1, n_cu, EpiStoreWT{C, N}, cfg); }, kReps);
      else ms = time_it([&] { gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{C, N}, cfg); }, kReps);
      printf("cfg %d %s  A[m][k]  M=%d N=%d K=%d: %.1f us  %.2f TFLOP/s (dense convention)\n", cfg, mode == 't' ? "tri-k" : "dense",
             M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    }
#ifdef VB_GEMM_CLOCK
    {
      std::vector<long long> o(8 * 4096);
      hipMemcpyFromSymbol(o.data(), HIP_SYMBOL(vb_gemm_dbg), o.size() * sizeof(long long));
      const int nb = (mode == 'g' || mode == 'G' || mode == 'D') ? (int)nb_mode : (M / ((cfg == 3 || cfg == 4) ? 64 : 128)) * (N / ((cfg == 1 || cfg == 6) ? 128 : 64));
      const int nw = 4 * (nb < 1024 ? nb : 1024);
      double pro = 0, loop = 0, epi = 0, wall = 0, first = 1e30, last = 0;
      for (int i = 0; i < nw; ++i) {
        pro += o[8 * i], loop += o[8 * i + 1], epi += o[8 * i + 2], wall += o[8 * i + 3];
        first = o[8 * i + 4] < first ? (double)o[8 * i + 4] : first;
        last = o[8 * i + 4] + o[8 * i + 3] > last ? (double)(o[8 * i + 4] + o[8 * i + 3]) : last;
      }
      pro /= nw, loop /= nw, epi /= nw, wall /= nw;
      const double mhz = (pro + loop + epi) / (wall / 100.0);
      printf("   per wave: prologue %.0f, main loop %.0f (%.1f per slab of the longest k range), epilogue %.0f shader cycles; lifetime %.1f us -> %.0f MHz; "
             "first start to last end %.1f us\n", pro, loop, loop / slabs_per_wg, epi, wall / 100.0, mhz, (last - first) / 100.0);
      {   // lifetimes of the individual workgroups (wave 0), sorted: the shape of the schedule
        std::vector<double> life, start;
        for (int i = 0; i < nw; i += 4) life.push_back(o[8 * i + 3] / 100.0), start.push_back((o[8 * i + 4] - first) / 100.0);
        std::vector<double> sl = life, ss = start;
        std::sort(sl.begin(), sl.end());
        std::sort(ss.begin(), ss.end());
        printf("   workgroup lifetimes (us): min %.1f  25%% %.1f  median %.1f  75%% %.1f  max %.1f;  start times: median %.1f  90%% %.1f  max %.1f\n",
               sl.front(), sl[sl.size() / 4], sl[sl.size() / 2], sl[3 * sl.size() / 4], sl.back(), ss[ss.size() / 2],
               ss[9 * ss.size() / 10], ss.back());
      }
      if (mode == 't') {   // tri_mode 1: the schedule per column block (launch order: heaviest first, then the light half ascending)
        const int bmr = (cfg == 3 || cfg == 4) ? 64 : 128, bnc = (cfg == 1 || cfg == 6) ? 128 : 64;
        const int tm = M / bmr, tn = N / bnc, half = (tn + 1) / 2;
        printf("   column block: k slabs | mean start, mean end (us from the first start) | mean lifetime | loop cycles per slab\n");
        for (int idx = 0; idx < tn; ++idx) {
          const int bn = idx < half ? tn - 1 - idx : idx - half;
          double s0 = 0, e0 = 0, lf = 0, lp = 0, emax = 0;
          int cnt = 0;
          for (int bmi = 0; bmi < tm; ++bmi) {
            const int b = idx * tm + bmi;
            if (b >= 1024) break;
            const long long* w = &o[8 * (4 * b)];
            s0 += (w[4] - first) / 100.0, e0 += (w[4] + w[3] - first) / 100.0, lf += w[3] / 100.0, lp += (double)w[1];
            emax = fmax(emax, (w[4] + w[3] - first) / 100.0);
            ++cnt;
          }
          if (!cnt) continue;
          const int ns = (bn + 1) * bnc / 16;
          printf("   bn %2d: %3d slabs | start %6.1f end %6.1f (last %6.1f) | life %6.1f us | %7.0f cycles per slab\n", bn, ns, s0 / cnt,
                 e0 / cnt, emax, lf / cnt, lp / cnt / ns);
        }
      }
      printf("   in us at that clock: prologue %.1f, loop %.1f, epilogue %.1f; MFMA floor of the loop (2 waves/SIMD x 16 cycles) %.1f us\n",
             pro / mhz, loop / mhz, epi / mhz, 2.0 * (K / 16) * 4 * (cfg == 1 ? 64 : cfg == 2 ? 32 : 16) * 16 / mhz);
    }
#endif
    return 0;
  }
  {  // effective clock under pure MFMA load
    long long* d;
    hipMalloc(&d, n_cu * 8 * 2 * sizeof(long long));
    for (int wg = 1; wg <= 4; ++wg) {
      for (int iters : {20000}) {
        hipLaunchKernelGGL(clock_probe, dim3(n_cu * wg), dim3(256), 0, st, d, iters);
        hipDeviceSynchronize();
        std::vector<long long> o(n_cu * wg * 2);
        hipMemcpy(o.data(), d, o.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int i = 0; i < n_cu * wg; ++i) cyc += o[2 * i], wall += o[2 * i + 1];
        cyc /= n_cu * wg, wall /= n_cu * wg;
        const double us = wall / 100.0;
        printf("clock_probe wg/cu=%d iters=%d: %.0f shader cycles in %.1f us -> %.0f MHz; %.2f cycles per MFMA; %.1f TFLOP/s\n",
               wg, iters, cyc, us, cyc / us, cyc / (16.0 * iters) / wg, 512.0 * 16 * iters * 4 * n_cu * wg / us / 1e6);
      }
    }
  }

  {
    long long* d;
    hipMalloc(&d, n_cu * 8 * 2 * sizeof(long long));
    run_loop_probe<0, 4, 16>(st, n_cu, d, C);
    run_loop_probe<1, 4, 16>(st, n_cu, d, C);
    run_loop_probe<2, 4, 16>(st, n_cu, d, C);
    run_loop_probe<3, 4, 16>(st, n_cu, d, C);
    run_loop_probe<0, 4, 8>(st, n_cu, d, C);
    run_loop_probe<1, 4, 8>(st, n_cu, d, C);
    run_loop_probe<2, 4, 8>(st, n_cu, d, C);
    run_loop_probe<3, 4, 8>(st, n_cu, d, C);
  }
  {   // correctness of the LDS-DMA kernel against the register-staged kernel (same inputs, all three tiles)
    std::vector<double> c0((size_t)M * N), c1((size_t)M * N);
    double* Bu;      // upper-triangular copy of B for the tri_mode 1 checks
    hipMalloc(&Bu, (size_t)K * N * 8);
    hipLaunchKernelGGL(make_upper_kernel, dim3((unsigned)(((size_t)K * N + 255) / 256)), dim3(256), 0, st, (const double*)B, Bu,
                       (int64_t)N, K, N);
    hipDeviceSynchronize();
    for (int cfg = 1; cfg <= 3; ++cfg)
      for (int tri = 0; tri <= 1; ++tri) {
        GemmArgs g;
        g.A = A, g.B = tri ? Bu : B, g.lda = K, g.ldb = N, g.M = M, g.N = N, g.K = K, g.tri_mode = tri;
        hipMemset(C, 0, (size_t)M * N * 8);
        gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{C, N}, cfg, 1);
        hipDeviceSynchronize();
        hipMemcpy(c0.data(), C, c0.size() * 8, hipMemcpyDeviceToHost);
        hipMemset(C, 0, (size_t)M * N * 8);
        gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{C, N}, cfg, 0);
        hipDeviceSynchronize();
        hipMemcpy(c1.data(), C, c1.size() * 8, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        for (size_t i = 0; i < c0.size(); ++i) {
          md = fmax(md, fabs(c0[i] - c1[i]));
          mx = fmax(mx, fabs(c0[i]));
        }
        printf("check cfg %d tri %d A[m][k]: max |dma - reg| = %.3e (max |C| %.3e)\n", cfg, tri, md, mx);
      }
    for (int cfg = 7; cfg <= 8; ++cfg)      // the 64 x 32 tiles (LDS-DMA only) against the register-staged 64 x 64 kernel
      for (int tri = 0; tri <= 1; ++tri) {
        GemmArgs g;
        g.A = A, g.B = tri ? Bu : B, g.lda = K, g.ldb = N, g.M = M, g.N = N, g.K = K, g.tri_mode = tri;
        hipMemset(C, 0, (size_t)M * N * 8);
        gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{C, N}, 3, 1);
        hipDeviceSynchronize();
        hipMemcpy(c0.data(), C, c0.size() * 8, hipMemcpyDeviceToHost);
        hipMemset(C, 0, (size_t)M * N * 8);
        gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{C, N}, cfg, 0);
        hipDeviceSynchronize();
        hipMemcpy(c1.data(), C, c1.size() * 8, hipMemcpyDeviceToHost);
        double md = 0;
        for (size_t i = 0; i < c0.size(); ++i) md = fmax(md, fabs(c0[i] - c1[i]));
        printf("check cfg %d tri %d (64 x 32 tiles): max |dma - reg 64 x 64| = %.3e\n", cfg, tri, md);
      }
    for (int cfg = 4; cfg <= 5; ++cfg) {     // the two-stage kernels against the register-staged kernel
      GemmArgs g;
      g.A = A, g.B = Bu, g.lda = K, g.ldb = N, g.M = M, g.N = N, g.K = K, g.tri_mode = 1;
      hipMemset(C, 0, (size_t)M * N * 8);
      gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{C, N}, cfg == 4 ? 3 : 2, 1);
      hipDeviceSynchronize();
      hipMemcpy(c0.data(), C, c0.size() * 8, hipMemcpyDeviceToHost);
      hipMemset(C, 0, (size_t)M * N * 8);
      gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{C, N}, cfg, 0);
      hipDeviceSynchronize();
      hipMemcpy(c1.data(), C, c1.size() * 8, hipMemcpyDeviceToHost);
      double md = 0;
      for (size_t i = 0; i < c0.size(); ++i) md = fmax(md, fabs(c0[i] - c1[i]));
      printf("check cfg %d tri 1: max |dma - reg| = %.3e\n", cfg, md);
    }
    // k-major A, lower-triangular tiles, split-K
    GemmArgs g3;
    g3.A = A, g3.B = B, g3.lda = N, g3.ldb = N, g3.M = N, g3.N = N, g3.K = M, g3.tri_mode = 2;
    std::vector<double> s0((size_t)2 * N * N), s1((size_t)2 * N * N);
    for (int cfg = 1; cfg <= 5; ++cfg) {
      for (int flag = 1; flag >= 0; --flag) {
        hipMemset(C, 0, (size_t)2 * N * N * 8);
        gemm_f64_launch<false>(st, g3, 2, n_cu, EpiSlab{C, N, (int64_t)N * N}, cfg, flag);
        hipDeviceSynchronize();
        hipMemcpy(flag ? s0.data() : s1.data(), C, s0.size() * 8, hipMemcpyDeviceToHost);
      }
      double md = 0;
      for (int sp = 0; sp < 2; ++sp)
        for (int i = 0; i < N; ++i)
          for (int j = 0; j <= i; ++j) {
            const size_t e = (size_t)sp * N * N + (size_t)i * N + j;
            md = fmax(md, fabs(s0[e] - s1[e]));
          }
      printf("check cfg %d gram A[k][m] splits 2 (lower triangle): max |dma - reg| = %.3e\n", cfg, md);
    }
  }
  for (int cfg = 1; cfg <= 5; ++cfg) {
    const char* name = cfg == 1 ? "128x128" : cfg == 2 ? "128x64" : cfg == 3 ? "64x64" : cfg == 4 ? "64x64/2st" : "128x64/2st";
    // 1. dense, A k-contiguous (Z = E L', (Z - m) P)
    GemmArgs g;
    g.A = A, g.B = B, g.lda = K, g.ldb = N, g.M = M, g.N = N, g.K = K, g.tri_mode = 0;
    float ms = time_it([&] { gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{C, N}, cfg); }, 20);
    printf("%-8s dense  A[m][k]  M=%d N=%d K=%d: %.1f us  %.2f TFLOP/s\n", name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    ms = time_it([&] { gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{C, N}, cfg, 1); }, 20);
    printf("%-8s dense  A[m][k]  (register-staged kernel): %.1f us  %.2f TFLOP/s\n", name, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    // 2. tri k-range
    g.tri_mode = 1;
    ms = time_it([&] { gemm_f64_launch<true>(st, g, 1, n_cu, EpiStore{C, N}, cfg); }, 20);
    printf("%-8s tri-k  A[m][k]  M=%d N=%d K=%d: %.1f us  %.2f TFLOP/s (dense convention)\n", name, M, N, K, ms * 1e3,
           2.0 * M * N * K / ms / 1e9);
    // 3. C = G' E: M = N = D, K = n rows; lower-triangular tiles, split-K
    GemmArgs g3;
    g3.A = A, g3.B = B, g3.lda = N, g3.ldb = N, g3.M = N, g3.N = N, g3.K = M, g3.tri_mode = 2;
    const long lower = gemm_count_blocks(g3, (cfg == 3 || cfg == 4) ? 64 : 128, cfg == 1 ? 128 : 64);
    for (int mult = 1; mult <= 4; ++mult) {
      int splits = (int)(mult * n_cu / lower);
      if (splits < 1) splits = 1;
      if ((int64_t)splits * N * N > (int64_t)2 * 4096 * 4096) continue;
      ms = time_it([&] { gemm_f64_launch<false>(st, g3, splits, n_cu, EpiSlab{C, N, (int64_t)N * N}, cfg); }, 20);
      printf("%-8s gram   A[k][m]  D=%d n=%d splits=%d: %.1f us  %.2f TFLOP/s (dense convention)\n", name, N, M, splits,
             ms * 1e3, 2.0 * M * N * N / ms / 1e9);
    }
  }
  return 0;
}
