// numpy's legacy generator on the host, in C++: MT19937 + the polar-method normals, Marsaglia-Tsang gammas,
// chi-square and Student-t draws of `numpy.random.RandomState` -- the streams the reference's families draw their
// noise from (approximations.py:203 `randn`, :273-274 `standard_t`, :342-345 `chisquare` then `randn`).  SURVEY 8(f)
// N2: with this the parity mode (`rng='numpy'`) does not depend on numpy for its noise, and the big normal matrices
// are produced several times faster than `RandomState.randn` does it:
//
//   * every attempt of the polar method consumes exactly four 32-bit words (two 53-bit doubles) whether it is
//     accepted or not, so attempt i reads words [4 i, 4 i + 4) of the stream and the k-th ACCEPTED attempt writes
//     outputs 2 k (= f x2, what legacy_gauss returns first) and 2 k + 1 (= f x1, the value it caches).  The words are
//     generated sequentially (MT19937 is one recurrence), the attempts are then evaluated by all host threads with
//     a prefix sum over the acceptance counts -- the log / sqrt / divide of the transform is 80 % of numpy's time;
//   * the number of attempts a request needs is random: rounds deliberately ask for a few standard deviations fewer
//     than expected, so that everything generated is consumed, and the short last round rewinds the generator to
//     just behind the attempt that produced the last value -- the state left behind (key, position, cached normal)
//     is exactly numpy's, call after call.
//
// The arithmetic per value is numpy's, operation by operation, in IEEE double with the C library's log / sqrt / pow
// (the same libm numpy calls); this file is compiled with -ffp-contract=off so that no product-sum is fused that
// numpy's build does not fuse.  tests/test_legacy_rng_cpu.py compares values AND generator state with numpy bit for
// bit.  Host code only: nothing here touches the GPU.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <atomic>
#include <vector>

#include <link.h>

#include "../../include/viabel_hip.h"
#include "vb_glibc_log.h"

namespace {

constexpr int kN = 624, kM = 397;
constexpr uint32_t kMatrixA = 0x9908b0dfu, kUpper = 0x80000000u, kLower = 0x7fffffffu;

}  // namespace

struct vb_legacy_rng {
  uint32_t key[kN];
  int pos;
  int has_gauss;
  double gauss;
  uint32_t out[kN];                  // key[] tempered (the block's output words), refreshed with it
  std::vector<uint32_t> buf[2];      // word buffers of the threaded normal draws (kept: no page faults per call)
  uint64_t uid = 0;                  // unique per created generator (an address may be handed out again after a destroy)
};

namespace {

inline uint32_t temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

void temper_block(vb_legacy_rng& s) {
  for (int i = 0; i < kN; ++i) s.out[i] = temper(s.key[i]);
}

void mt_seed(vb_legacy_rng& s, uint32_t seed) {          // numpy's mt19937_seed (Knuth's initialiser)
  for (int i = 0; i < kN; ++i) {
    s.key[i] = seed;
    seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
  }
  s.pos = kN;
  s.has_gauss = 0;
  s.gauss = 0.0;
  temper_block(s);
}

void mt_refresh(uint32_t* key) {                          // the next 624 words of the recurrence, in place
  int k = 0;
  for (; k < kN - kM; ++k) {
    const uint32_t y = (key[k] & kUpper) | (key[k + 1] & kLower);
    key[k] = key[k + kM] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrixA);
  }
  for (; k < kN - 1; ++k) {
    const uint32_t y = (key[k] & kUpper) | (key[k + 1] & kLower);
    key[k] = key[k + (kM - kN)] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrixA);
  }
  const uint32_t y = (key[kN - 1] & kUpper) | (key[0] & kLower);
  key[kN - 1] = key[kM - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrixA);
}

inline void next_block(vb_legacy_rng& s) {
  mt_refresh(s.key);
  temper_block(s);
  s.pos = 0;
}

inline uint32_t next_u32(vb_legacy_rng& s) {
  if (s.pos == kN) next_block(s);
  return s.out[s.pos++];
}

inline double words_to_double(uint32_t w0, uint32_t w1) {       // numpy's 53-bit double from two words
  const int32_t a = (int32_t)(w0 >> 5), b = (int32_t)(w1 >> 6);
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

inline double next_double(vb_legacy_rng& s) {
  const uint32_t w0 = next_u32(s), w1 = next_u32(s);
  return words_to_double(w0, w1);
}

// `count` tempered words into `out`, advancing the state
void fill_words(vb_legacy_rng& s, uint32_t* out, size_t count) {
  size_t done = 0;
  while (done < count) {
    if (s.pos == kN) next_block(s);
    const size_t take = std::min<size_t>(count - done, (size_t)(kN - s.pos));
    memcpy(out + done, s.out + s.pos, take * sizeof(uint32_t));
    s.pos += (int)take;
    done += take;
  }
}

void skip_words(vb_legacy_rng& s, size_t count) {
  while (count > 0) {
    if (s.pos == kN) next_block(s);
    const size_t take = std::min<size_t>(count, (size_t)(kN - s.pos));
    s.pos += (int)take;
    count -= take;
  }
}

// ---- numpy's legacy distributions (legacy-distributions.c), scalar ------------------------------------------
double legacy_gauss(vb_legacy_rng& s) {
  if (s.has_gauss) {
    const double tmp = s.gauss;
    s.has_gauss = 0;
    s.gauss = 0.0;
    return tmp;
  }
  double x1, x2, r2;
  do {
    x1 = 2.0 * next_double(s) - 1.0;
    x2 = 2.0 * next_double(s) - 1.0;
    r2 = x1 * x1 + x2 * x2;
  } while (r2 >= 1.0 || r2 == 0.0);
  const double f = std::sqrt(-2.0 * std::log(r2) / r2);       // polar method (Box-Muller without the trigonometry)
  s.gauss = f * x1;
  s.has_gauss = 1;
  return f * x2;
}

double legacy_standard_exponential(vb_legacy_rng& s) { return -std::log(1.0 - next_double(s)); }

double legacy_standard_gamma(vb_legacy_rng& s, double shape) {
  if (shape == 1.0) return legacy_standard_exponential(s);
  if (shape == 0.0) return 0.0;
  if (shape < 1.0) {
    for (;;) {
      const double U = next_double(s);
      const double V = legacy_standard_exponential(s);
      if (U <= 1.0 - shape) {
        const double X = std::pow(U, 1. / shape);
        if (X <= V) return X;
      } else {
        const double Y = -std::log((1 - U) / shape);
        const double X = std::pow(1.0 - shape + shape * Y, 1. / shape);
        if (X <= (V + Y)) return X;
      }
    }
  }
  const double b = shape - 1. / 3.;
  const double c = 1. / std::sqrt(9 * b);
  for (;;) {
    double X, V;
    do {
      X = legacy_gauss(s);
      V = 1.0 + c * X;
    } while (V <= 0.0);
    V = V * V * V;
    const double U = next_double(s);
    if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return (b * V);
    if (std::log(U) < 0.5 * X * X + b * (1. - V + std::log(V))) return (b * V);
  }
}

// ---- the big normal draws: attempts evaluated in parallel -------------------------------------------------------
struct Attempt {
  double x1, x2, r2;
  bool ok;
};

inline Attempt attempt_at(const uint32_t* w) {
  Attempt a;
  a.x1 = 2.0 * words_to_double(w[0], w[1]) - 1.0;
  a.x2 = 2.0 * words_to_double(w[2], w[3]) - 1.0;
  a.r2 = a.x1 * a.x1 + a.x2 * a.x2;
  a.ok = !(a.r2 >= 1.0 || a.r2 == 0.0);
  return a;
}

int thread_count(int asked) {
  if (asked <= 0) {
    const char* env = getenv("VIABEL_AMD_RNG_THREADS");
    asked = env ? atoi(env) : 0;
  }
  if (asked <= 0) {
    const unsigned hw = std::thread::hardware_concurrency();
    asked = hw == 0 ? 4 : (int)std::min(hw, 8u);     // (threads are started per chunk: beyond 8 the starts cost more than they save)
  }
  return asked;
}

template <class F>
void run_slices(int threads, size_t items, F&& body) {        // body(t, begin, end)
  if (threads <= 1) {
    body(0, (size_t)0, items);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve((size_t)threads - 1);
  const size_t per = (items + (size_t)threads - 1) / (size_t)threads;
  for (int t = 1; t < threads; ++t) {
    const size_t b = std::min(items, per * (size_t)t), e = std::min(items, b + per);
    pool.emplace_back([&body, t, b, e]() { body(t, b, e); });
  }
  body(0, (size_t)0, std::min(items, per));
  for (auto& th : pool) th.join();
}

constexpr double kAccept = 0.78539816339744830962;        // pi / 4: a point of the square falls inside the unit disc
constexpr size_t kChunk = (size_t)1 << 18;                 // attempts per chunk: 4 MiB of words, two chunks in flight
constexpr int64_t kParallelFrom = (int64_t)1 << 15;        // below this many values the scalar loop is as fast

struct ChunkResult {
  int64_t accepted = 0;      // accepted attempts in the chunk
  int64_t a_star = -1;       // index of the attempt that produced the request's last pair (-1: not in this chunk)
  double last_x1f = 0.0;     // f * x1 of that pair (the value numpy caches when n is odd)
};

// Evaluate `attempts` attempts on words `w`; the first accepted one is pair number `base` of the request.
ChunkResult process_chunk(const uint32_t* w, size_t attempts, int64_t base, int64_t pairs, int64_t n, double* out,
                          int threads) {
  const int use = attempts < 4096 ? 1 : std::max(1, threads);
  std::vector<int64_t> counts((size_t)use + 1, 0);
  run_slices(use, attempts, [&](int t, size_t b, size_t e) {
    int64_t c = 0;
    for (size_t i = b; i < e; ++i) c += attempt_at(w + 4 * i).ok ? 1 : 0;
    counts[(size_t)t + 1] = c;
  });
  for (int t = 0; t < use; ++t) counts[(size_t)t + 1] += counts[(size_t)t];      // exclusive prefix in counts[t]
  ChunkResult r;
  r.accepted = counts[(size_t)use];
  std::vector<int64_t> last_attempt((size_t)use, -1);
  std::vector<double> last_val((size_t)use, 0.0);
  run_slices(use, attempts, [&](int t, size_t b, size_t e) {
    int64_t q = base + counts[(size_t)t];
    for (size_t i = b; i < e && q < pairs; ++i) {
      const Attempt a = attempt_at(w + 4 * i);
      if (!a.ok) continue;
      const double f = std::sqrt(-2.0 * std::log(a.r2) / a.r2);
      out[2 * q] = f * a.x2;
      if (2 * q + 1 < n) out[2 * q + 1] = f * a.x1;
      if (q == pairs - 1) {            // one slice only reaches the last pair
        last_attempt[(size_t)t] = (int64_t)i;
        last_val[(size_t)t] = f * a.x1;
      }
      ++q;
    }
  });
  for (int t = 0; t < use; ++t)
    if (last_attempt[(size_t)t] >= 0) {
      r.a_star = last_attempt[(size_t)t];
      r.last_x1f = last_val[(size_t)t];
    }
  return r;
}

struct MtCore {
  uint32_t key[kN];
  int pos;
};

void randn_parallel(vb_legacy_rng& s, double* out, int64_t n, int threads) {
  if (n > 0 && s.has_gauss) {
    *out++ = s.gauss;
    s.has_gauss = 0;
    s.gauss = 0.0;
    --n;
  }
  if (n <= 0) return;
  const int64_t pairs = (n + 1) / 2;
  int64_t done = 0;
  double last_x1f = 0.0;
  auto save = [&s](MtCore& c) {
    memcpy(c.key, s.key, sizeof c.key);
    c.pos = s.pos;
  };
  auto restore = [&s](const MtCore& c) {
    memcpy(s.key, c.key, sizeof c.key);
    s.pos = c.pos;
    temper_block(s);
  };
  while (done < pairs) {
    const int64_t rem = pairs - done;
    const double expected = (double)rem / kAccept;
    const double sigma = std::sqrt((double)rem * (1.0 - kAccept)) / kAccept;
    // large remainders: fewer attempts than will be needed (8 sigma), so all of them are consumed and nothing has to
    // be rewound; the short tail asks for more than enough and rewinds a few thousand words at most
    const size_t attempts = rem > 2048 ? (size_t)std::max(1.0, expected - 8.0 * sigma) : (size_t)(expected * 1.5) + 64;
    // chunks of the round, double-buffered: while the workers evaluate chunk c the calling thread draws the words of
    // chunk c + 1 (the recurrence is the one sequential part: ~12 ms per 10.7 M words, hidden behind the transform)
    const size_t n_chunks = (attempts + kChunk - 1) / kChunk;
    MtCore snap[2];
    auto chunk_size = [&](size_t c) { return std::min(kChunk, attempts - c * kChunk); };
    for (int b = 0; b < 2; ++b) s.buf[b].resize(4 * std::min(kChunk, attempts));
    save(snap[0]);
    fill_words(s, s.buf[0].data(), 4 * chunk_size(0));
    bool finished = false;
    for (size_t c = 0; c < n_chunks && !finished; ++c) {
      const size_t m = chunk_size(c);
      const uint32_t* w = s.buf[c & 1].data();
      ChunkResult r;
      if (c + 1 < n_chunks) {
        std::thread stage([&]() { r = process_chunk(w, m, done, pairs, n, out, threads - 1); });
        save(snap[(c + 1) & 1]);
        fill_words(s, s.buf[(c + 1) & 1].data(), 4 * chunk_size(c + 1));
        stage.join();
      } else {
        r = process_chunk(w, m, done, pairs, n, out, threads);
      }
      if (r.a_star >= 0) {             // the request completed inside this chunk: rewind to just behind that attempt
        restore(snap[c & 1]);
        skip_words(s, 4 * (size_t)(r.a_star + 1));
        last_x1f = r.last_x1f;
        done = pairs;
        finished = true;
      } else {
        done += r.accepted;
      }
    }
  }
  if (n & 1) {
    s.gauss = last_x1f;
    s.has_gauss = 1;
  }
}

}  // namespace

// The attempts the device could not finish bit for bit (vb_legacy_dev.hip: the true log lies too close to a rounding
// boundary to know which neighbour THIS C library returns): list entries (q, x1, x2, r2) -> (q, f x2, f x1) with the
// host's own log, on all host threads.  Internal (declared in vb_common.h), not part of the C ABI.
namespace vb {
void vb_legacy_finish_pairs(const double* list, int64_t n, double* fixed) {
  run_slices(n < 4096 ? 1 : thread_count(0), (size_t)n, [&](int, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const double x1 = list[4 * i + 1], x2 = list[4 * i + 2], r2 = list[4 * i + 3];
      const double f = std::sqrt(-2.0 * std::log(r2) / r2);
      fixed[3 * i] = list[4 * i];
      fixed[3 * i + 1] = f * x2;
      fixed[3 * i + 2] = f * x1;
    }
  });
}
}  // namespace vb

// ---- the host libm's log table, found and proven (vb_glibc_log.h) -------------------------------------------------------
namespace {

struct LogSearch {
  std::vector<const vb::GlibcLogData*> found;
};

int log_search_cb(struct dl_phdr_info* info, size_t, void* user) {
  LogSearch* s = (LogSearch*)user;
  if (!info->dlpi_name || !strstr(info->dlpi_name, "libm")) return 0;
  // `__log_data` starts with (ln2hi, ln2lo) -- as does `__pow_log_data`, whose first coefficient is exactly -0.5
  const double pat[2] = {0x1.62e42fefa3800p-1, 0x1.ef35793c76730p-45};
  for (int i = 0; i < info->dlpi_phnum; ++i) {
    const ElfW(Phdr)& ph = info->dlpi_phdr[i];
    if (ph.p_type != PT_LOAD || !(ph.p_flags & PF_R) || ph.p_filesz < sizeof(vb::GlibcLogData)) continue;
    const char* base = (const char*)(info->dlpi_addr + ph.p_vaddr);
    const char* end = base + ph.p_filesz - sizeof(vb::GlibcLogData);
    for (const char* p = base; p <= end;) {
      const void* hit = memmem(p, (size_t)(end - p) + sizeof pat, pat, sizeof pat);
      if (!hit) break;
      if (((uintptr_t)hit & 7) == 0) {
        const vb::GlibcLogData* d = (const vb::GlibcLogData*)hit;
        if (d->A[0] < -0.49 && d->A[0] > -0.51 && d->A[0] != -0.5 && d->B[0] == -0.5) s->found.push_back(d);
      }
      p = (const char*)hit + 8;
    }
  }
  return 0;
}

// the restated sequence against the host's own log() on ~1.3 M arguments of the kinds the generator produces
__attribute__((target("fma"))) bool log_table_proven(const vb::GlibcLogData& T) {
  uint64_t x = 0x9e3779b97f4a7c15ull;
  auto next = [&x]() {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    return x;
  };
  auto same = [&T](double v) __attribute__((target("fma"))) {
    const double a = vb::glibc_log(v, T), b = std::log(v);
    return memcmp(&a, &b, sizeof a) == 0;
  };
  for (int i = 0; i < 600000; ++i) {               // 53-bit uniforms in (0, 1): r2 and U
    const double u = (double)(next() >> 11) / 9007199254740992.0;
    if (u > 0.0 && !same(u)) return false;
  }
  for (int i = 0; i < 300000; ++i) {               // around 1, both branches' borders included
    const double u = 0.92 + 0.16 * ((double)(next() >> 11) / 9007199254740992.0);
    if (!same(u)) return false;
  }
  for (int i = 0; i < 300000; ++i) {               // V = (1 + c X)^3: 2^-160 ... 2^7
    const double m = 1.0 + (double)(next() >> 12) / 4503599627370496.0;
    const int e = (int)(next() % 168) - 160;
    if (!same(std::ldexp(m, e))) return false;
  }
  for (int i = 0; i < 128; ++i)                    // every table interval's ends
    for (int j = -2; j <= 2; ++j) {
      uint64_t bits = 0x3fe6000000000000ull + ((uint64_t)i << 45) + (uint64_t)(int64_t)j;
      double v;
      memcpy(&v, &bits, sizeof v);
      if (!same(v) || !same(0.5 * v) || !same(0x1p-30 * v)) return false;
    }
  for (int k = 1; k < 4096; ++k)                   // the smallest uniforms
    if (!same((double)k / 9007199254740992.0)) return false;
  return true;
}

}  // namespace

namespace vb {
const GlibcLogData* vb_glibc_log_locate() {
  static const GlibcLogData* proven = []() -> const GlibcLogData* {
    if (getenv("VIABEL_AMD_NO_GLIBC_LOG")) return nullptr;      // (tests: force the double-double + host path)
    if (!__builtin_cpu_supports("fma")) return nullptr;          // libm runs another variant of its log here
    LogSearch s;
    dl_iterate_phdr(log_search_cb, &s);
    for (const GlibcLogData* d : s.found)
      if (log_table_proven(*d)) return d;
    return nullptr;
  }();
  return proven;
}
}  // namespace vb

extern "C" {

// 1 when the host libm's log has been restated and proven bit for bit (the device draws need no host round trip)
int vb_legacy_rng_log_proven(void) { return vb::vb_glibc_log_locate() ? 1 : 0; }

int vb_legacy_rng_create(uint32_t seed, vb_legacy_rng** out) {
  if (!out) return VB_ERR_INVALID;
  vb_legacy_rng* s = new (std::nothrow) vb_legacy_rng;
  if (!s) return VB_ERR_HIP;          // (out of host memory)
  mt_seed(*s, seed);
  static std::atomic<uint64_t> next_uid{1};
  s->uid = next_uid.fetch_add(1);
  *out = s;
  return VB_OK;
}

uint64_t vb_legacy_rng_uid(const vb_legacy_rng* s) { return s ? s->uid : 0; }

void vb_legacy_rng_destroy(vb_legacy_rng* s) { delete s; }

int vb_legacy_rng_get_state(const vb_legacy_rng* s, uint32_t key[624], int* pos, int* has_gauss, double* gauss) {
  if (!s || !key || !pos || !has_gauss || !gauss) return VB_ERR_INVALID;
  memcpy(key, s->key, sizeof s->key);
  *pos = s->pos;
  *has_gauss = s->has_gauss;
  *gauss = s->gauss;
  return VB_OK;
}

int vb_legacy_rng_set_state(vb_legacy_rng* s, const uint32_t key[624], int pos, int has_gauss, double gauss) {
  if (!s || !key || pos < 0 || pos > kN) return VB_ERR_INVALID;
  memcpy(s->key, key, sizeof s->key);
  s->pos = pos;
  s->has_gauss = has_gauss ? 1 : 0;
  s->gauss = has_gauss ? gauss : 0.0;
  temper_block(*s);
  return VB_OK;
}

int vb_legacy_rng_randn(vb_legacy_rng* s, double* out, int64_t n, int threads) {
  if (!s || n < 0 || (n > 0 && !out)) return VB_ERR_INVALID;
  if (n < kParallelFrom) {
    for (int64_t i = 0; i < n; ++i) out[i] = legacy_gauss(*s);
    return VB_OK;
  }
  randn_parallel(*s, out, n, thread_count(threads));
  return VB_OK;
}

int vb_legacy_rng_standard_t(vb_legacy_rng* s, double df, double* out, int64_t n) {
  if (!s || n < 0 || (n > 0 && !out) || !(df > 0.0)) return VB_ERR_INVALID;
  for (int64_t i = 0; i < n; ++i) {
    const double num = legacy_gauss(*s);
    const double denom = legacy_standard_gamma(*s, df / 2);
    out[i] = std::sqrt(df / 2) * num / std::sqrt(denom);
  }
  return VB_OK;
}

int vb_legacy_rng_chisquare(vb_legacy_rng* s, double df, double* out, int64_t n) {
  if (!s || n < 0 || (n > 0 && !out) || !(df > 0.0)) return VB_ERR_INVALID;
  for (int64_t i = 0; i < n; ++i) out[i] = 2.0 * legacy_standard_gamma(*s, df / 2.0);
  return VB_OK;
}

int vb_legacy_rng_random_sample(vb_legacy_rng* s, double* out, int64_t n) {
  if (!s || n < 0 || (n > 0 && !out)) return VB_ERR_INVALID;
  for (int64_t i = 0; i < n; ++i) out[i] = next_double(*s);
  return VB_OK;
}

}  // extern "C"
