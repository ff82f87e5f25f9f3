// tables.hip — host-side generation + per-device caching of the transform tables.
#include "tables.h"
#include "arith.h"

#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace tfhe_hip {

static uint32_t ilog2(uint32_t x) {
  uint32_t l = 0;
  while ((1u << l) < x) ++l;
  return l;
}
static uint32_t bitrev(uint32_t x, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; ++i) {
    r = (r << 1) | (x & 1);
    x >>= 1;
  }
  return r;
}

// DESIGN.md §4.  Angles are evaluated in long double and rounded once to double.
void fill_fft_tables_host(uint32_t N, double *fwd, double *inv, double *untw) {
  const uint32_t n = N / 2, D = ilog2(n);
  const long double PI = 3.14159265358979323846264338327950288L;
  // forward: even g evaluated, odd sibling = i * even (an exact quarter turn)
  fwd[0] = fwd[1] = 0.0;
  for (uint32_t d = 0; d < D; ++d)
    for (uint32_t g = 0; g < (1u << d); ++g) {
      const uint32_t idx = (1u << d) + g;
      if (g & 1) {
        fwd[2 * idx] = -fwd[2 * (idx - 1) + 1];
        fwd[2 * idx + 1] = fwd[2 * (idx - 1)];
      } else {
        const uint32_t r = 1 + 4 * bitrev(g, d);
        const long double ang = PI * (long double)r / (long double)(1u << (d + 2));
        fwd[2 * idx] = (double)cosl(ang);
        fwd[2 * idx + 1] = (double)sinl(ang);
      }
    }
  // backward: w = exp(-2*pi*i*j/(2*half)); j >= half/2 derived as -i * w[j - half/2]
  inv[0] = inv[1] = 0.0;
  for (uint32_t half = 1; half < n; half *= 2)
    for (uint32_t j = 0; j < half; ++j) {
      double c, s;
      if (j == 0) {
        c = 1.0;
        s = 0.0;
      } else if (half >= 2 && j >= half / 2) {
        c = inv[2 * (half + j - half / 2) + 1];
        s = -inv[2 * (half + j - half / 2)];
      } else {
        const long double ang = -PI * (long double)j / (long double)half;
        c = (double)cosl(ang);
        s = (double)sinl(ang);
      }
      inv[2 * (half + j)] = c;
      inv[2 * (half + j) + 1] = s;
    }
  // untwist: conj(exp(i*pi*j/N))/n for j <= n/2, mirrored (cos <-> sin) above
  for (uint32_t j = 0; j < n; ++j) {
    if (j <= n / 2) {
      const long double ang = PI * (long double)j / (long double)N;
      untw[2 * j] = (double)cosl(ang) / (double)n;
      untw[2 * j + 1] = -((double)sinl(ang)) / (double)n;
    } else {
      untw[2 * j] = -untw[2 * (n - j) + 1];
      untw[2 * j + 1] = -untw[2 * (n - j)];
    }
  }
}

// e^{i pi j / N} for j < 2N: the first octant evaluated (long double, rounded once), the rest by exact symmetries,
// so that the quarter turns are exact and the table is closed under conjugation and multiplication by i
void fill_monomial_table_host(uint32_t N, double *z) {
  const long double PI = 3.14159265358979323846264338327950288L;
  for (uint32_t j = 0; j <= N / 4; ++j) {
    const long double ang = PI * (long double)j / (long double)N;
    z[2 * j] = (double)cosl(ang);
    z[2 * j + 1] = (double)sinl(ang);
  }
  z[0] = 1.0;
  z[1] = 0.0;
  for (uint32_t j = N / 4 + 1; j <= N / 2; ++j) {  // reflection about pi/4
    z[2 * j] = z[2 * (N / 2 - j) + 1];
    z[2 * j + 1] = z[2 * (N / 2 - j)];
  }
  for (uint32_t j = N / 2 + 1; j <= N; ++j) {  // times i
    z[2 * j] = -z[2 * (j - N / 2) + 1];
    z[2 * j + 1] = z[2 * (j - N / 2)];
  }
  for (uint32_t j = N + 1; j < 2 * N; ++j) {  // times -1
    z[2 * j] = -z[2 * (j - N)];
    z[2 * j + 1] = -z[2 * (j - N) + 1];
  }
}

// tfhe-fft/src/fft_simd.rs:239-295 (sincospi64, after https://stackoverflow.com/a/42792940): every operation as
// there, fused where it says mul_add
static void ref_sincospi64(double a, double *s_out, double *c_out) {
  const double az = a * 0.0;
  a = std::fabs(a) < 9007199254740992.0 ? a : az;
  double r = std::round(a + a);
  const int64_t i = (int64_t)r;
  const double t = std::fma(-0.5, r, a);
  double s = t * t;
  r = -1.0369917389758117e-4;
  r = std::fma(r, s, 1.9294935641298806e-3);
  r = std::fma(r, s, -2.5806887942825395e-2);
  r = std::fma(r, s, 2.3533063028328211e-1);
  r = std::fma(r, s, -1.3352627688538006e+0);
  r = std::fma(r, s, 4.0587121264167623e+0);
  r = std::fma(r, s, -4.9348022005446790e+0);
  double c = std::fma(r, s, 1.0000000000000000e+0);
  r = 4.6151442520157035e-4;
  r = std::fma(r, s, -7.3700183130883555e-3);
  r = std::fma(r, s, 8.2145868949323936e-2);
  r = std::fma(r, s, -5.9926452893214921e-1);
  r = std::fma(r, s, 2.5501640398732688e+0);
  r = std::fma(r, s, -5.1677127800499516e+0);
  s = s * t;
  r *= s;
  s = std::fma(t, 3.14159265358979323846264338327950288, r);
  if (i & 2) {
    s = 0.0 - s;
    c = 0.0 - c;
  }
  if (i & 1) {
    const double tt = 0.0 - s;
    s = c;
    c = tt;
  }
  if (a == std::floor(a)) s = az;
  *s_out = s;
  *c_out = c;
}
void fill_ref_tables_host(uint32_t N, double *twist, double *w, double *w_inv) {
  const uint32_t n = N / 2, nr = n / 4;
  const double unit = 3.14159265358979323846264338327950288 / (2.0 * (double)n);  // fft/mod.rs:68
  for (uint32_t i = 0; i < n; ++i) {
    twist[2 * i] = std::cos((double)i * unit);
    twist[2 * i + 1] = std::sin((double)i * unit);
  }
  for (uint32_t i = 0; i < 4 * n; ++i) w[i] = w_inv[i] = std::nan("");
  const double theta = -2.0 / (double)n;
  for (uint32_t q = 0; q < nr; ++q)
    for (uint32_t k = 1; k < 4; ++k) {
      double s, c;
      ref_sincospi64(theta * (double)(k * q), &s, &c);
      for (const uint32_t at : {q + k * nr, n + 4 * q + k}) {
        w[2 * at] = c;
        w[2 * at + 1] = s;
        w_inv[2 * at] = c;
        w_inv[2 * at + 1] = -s;
      }
    }
}

static uint64_t gl_pow_host(uint64_t a, uint64_t e) {
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, a);
    a = gl_mul(a, a);
    e >>= 1;
  }
  return r;
}

// psi: the reference's fixed Goldilocks roots (tfhe-ntt/src/prime64.rs:162-179), so the
// NTT-domain key equals the reference's NttLweBootstrapKey value for value.
static uint64_t gl_root_2N(uint32_t N) {
  switch (N) {
    case 256: return 14430643036723656017ull;
    case 512: return 4440654710286119610ull;
    case 1024: return 8816101479115663336ull;
    case 2048: return 10974926054405199669ull;
    case 4096: return 1206500561358145487ull;
    default: return gl_pow_host(7, (GL_P - 1) / (2ull * N));  // 7 generates Z_p^*
  }
}
void fill_ntt_tables_host(uint32_t N, uint64_t *tw, uint64_t *itw, uint64_t *n_inv) {
  const uint32_t lg = ilog2(N);
  const uint64_t psi = gl_root_2N(N);
  const uint64_t psi_inv = gl_pow_host(psi, GL_P - 2);
  for (uint32_t i = 0; i < N; ++i) {
    const uint32_t e = bitrev(i, lg);
    tw[i] = gl_pow_host(psi, e);
    itw[i] = gl_pow_host(psi_inv, e);
  }
  *n_inv = gl_pow_host(N, GL_P - 2);
}

namespace {
struct FftEntry {
  double *fwd, *inv, *untw, *mono, *mono_lane;
};
struct NttEntry {
  uint64_t *tw, *itw;
  uint64_t n_inv;
};
struct RefEntry {
  double *twist, *w, *w_inv;
};
std::mutex g_mu;
std::map<std::pair<uint32_t, uint32_t>, FftEntry> g_fft;
std::map<std::pair<uint32_t, uint32_t>, NttEntry> g_ntt;
std::map<std::pair<uint32_t, uint32_t>, RefEntry> g_ref;
}  // namespace

RefTables get_ref_tables(uint32_t gpu_index, hipStream_t stream, uint32_t N) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto key = std::make_pair(gpu_index, N);
  auto it = g_ref.find(key);
  if (it == g_ref.end()) {
    const size_t n = N / 2;
    std::vector<double> twist(2 * n), w(4 * n), w_inv(4 * n);
    fill_ref_tables_host(N, twist.data(), w.data(), w_inv.data());
    RefEntry e;
    HX_CHECK(hipSetDevice((int)gpu_index));
    HX_CHECK(hipMalloc((void **)&e.twist, sizeof(double) * 2 * n));
    HX_CHECK(hipMalloc((void **)&e.w, sizeof(double) * 4 * n));
    HX_CHECK(hipMalloc((void **)&e.w_inv, sizeof(double) * 4 * n));
    HX_CHECK(hipMemcpy(e.twist, twist.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    HX_CHECK(hipMemcpy(e.w, w.data(), sizeof(double) * 4 * n, hipMemcpyHostToDevice));
    HX_CHECK(hipMemcpy(e.w_inv, w_inv.data(), sizeof(double) * 4 * n, hipMemcpyHostToDevice));
    (void)stream;
    it = g_ref.emplace(key, e).first;
  }
  return RefTables{it->second.twist, it->second.w, it->second.w_inv};
}

// pbs_fft_wave.hip: the twiddles that kernel carries as literals of its instruction stream
bool wave_literal_twiddles_match(const double *fwd, const double *inv);
bool wave3_literal_twiddles_match(const double *fwd, const double *inv);  // pbs_fft_wave3.hip, N = 1024

FftTables get_fft_tables(uint32_t gpu_index, hipStream_t stream, uint32_t N, bool with_mono_lane) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto key = std::make_pair(gpu_index, N);
  auto it = g_fft.find(key);
  if (it == g_fft.end()) {
    std::vector<double> fwd(N), inv(N), untw(N), mono(4 * (size_t)N);
    fill_fft_tables_host(N, fwd.data(), inv.data(), untw.data());
    fill_monomial_table_host(N, mono.data());
    if (N == 2048)
      HX_PANIC_IF_FALSE(wave_literal_twiddles_match(fwd.data(), inv.data()),
                        "the literal twiddles of the N = 2048 throughput kernel differ from the host tables");
    if (N == 1024)
      HX_PANIC_IF_FALSE(wave3_literal_twiddles_match(fwd.data(), inv.data()),
                        "the literal twiddles of the N = 1024 throughput kernel differ from the host tables");
    FftEntry e;
    HX_CHECK(hipSetDevice((int)gpu_index));
    HX_CHECK(hipMalloc((void **)&e.fwd, sizeof(double) * N));
    HX_CHECK(hipMalloc((void **)&e.inv, sizeof(double) * N));
    HX_CHECK(hipMalloc((void **)&e.untw, sizeof(double) * N));
    HX_CHECK(hipMalloc((void **)&e.mono, sizeof(double) * 4 * N));
    HX_CHECK(hipMemcpy(e.mono, mono.data(), sizeof(double) * 4 * N, hipMemcpyHostToDevice));
    e.mono_lane = nullptr;
    // synchronous copies from pageable host memory: complete before we return
    HX_CHECK(hipMemcpy(e.fwd, fwd.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    HX_CHECK(hipMemcpy(e.inv, inv.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    HX_CHECK(hipMemcpy(e.untw, untw.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    (void)stream;
    it = g_fft.emplace(key, e).first;
  }
  if (with_mono_lane && N == 2048 && it->second.mono_lane == nullptr) {
    // tables.h: the base factors of the multi-bit kernels in lane order (4 MB): built for the first multi-bit scratch
    // of the device, not for callers of the classic PBS
    std::vector<double> mono(4 * (size_t)N);
    fill_monomial_table_host(N, mono.data());
    std::vector<double> ml((size_t)2 * N * 64 * 2);
    for (uint32_t d = 0; d < 2 * N; ++d)
      for (uint32_t h = 0; h < 64; ++h) {
        uint32_t br = 0;
        for (int b = 0; b < 6; ++b) br |= ((h >> b) & 1u) << (5 - b);
        const uint32_t j = ((1u + 4u * br) * d) & (2u * N - 1u);
        ml[((size_t)d * 64 + h) * 2] = mono[2 * (size_t)j];
        ml[((size_t)d * 64 + h) * 2 + 1] = mono[2 * (size_t)j + 1];
      }
    HX_CHECK(hipSetDevice((int)gpu_index));
    HX_CHECK(hipMalloc((void **)&it->second.mono_lane, sizeof(double) * ml.size()));
    HX_CHECK(hipMemcpy(it->second.mono_lane, ml.data(), sizeof(double) * ml.size(), hipMemcpyHostToDevice));
  }
  return FftTables{it->second.fwd, it->second.inv, it->second.untw, it->second.mono, it->second.mono_lane};
}


NttTables get_ntt_tables(uint32_t gpu_index, hipStream_t stream, uint32_t N) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto key = std::make_pair(gpu_index, N);
  auto it = g_ntt.find(key);
  if (it == g_ntt.end()) {
    std::vector<uint64_t> tw(N), itw(N);
    NttEntry e;
    fill_ntt_tables_host(N, tw.data(), itw.data(), &e.n_inv);
    HX_CHECK(hipSetDevice((int)gpu_index));
    HX_CHECK(hipMalloc((void **)&e.tw, sizeof(uint64_t) * N));
    HX_CHECK(hipMalloc((void **)&e.itw, sizeof(uint64_t) * N));
    HX_CHECK(hipMemcpy(e.tw, tw.data(), sizeof(uint64_t) * N, hipMemcpyHostToDevice));
    HX_CHECK(hipMemcpy(e.itw, itw.data(), sizeof(uint64_t) * N, hipMemcpyHostToDevice));
    (void)stream;
    it = g_ntt.emplace(key, e).first;
  }
  return NttTables{it->second.tw, it->second.itw, it->second.n_inv};
}

}  // namespace tfhe_hip
