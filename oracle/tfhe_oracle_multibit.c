/*
 * tfhe_oracle_multibit.c — multi-bit PBS restatement.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows cc/algorithms/lwe_multi_bit_programmable_bootstrapping.rs (deterministic
 * variant, :647-880) and cc/algorithms/lwe_multi_bit_bootstrap_key_generation.rs.
 *
 * f64 engine: like the CPU reference, the group's 2^g GGSWs are kept in the FOURIER domain and
 * combined there (prepare_multi_bit_ggsw_mem_optimized :116-156):
 *     GGSW_comb = GGSW_0 + sum_{s >= 1} GGSW_s (.) FFT(X^{deg_s})
 * where the transform of a monomial is a vector of roots of unity (fft/mod.rs:411-446
 * incomplete_monomial_forward_as_integer + tfhe-fft unordered.rs fwd_monomial).  In this repository's
 * fixed transform order (position p holds the evaluation at zeta^(1 + 4 bitrev(p)),
 * zeta = e^{i pi/N}) the monomial factor at position p is, bit for bit,
 *     M_d[p] = cmul( Z[((1 + 4 bitrev_{L-4}(p >> 4)) d) mod 2N] , Z[(N/8) ((bitrev_4(p & 15) d) mod 16)] )
 * with Z[j] = e^{i pi j / N} (2N entries, octant-symmetric, evaluated in long double and rounded once),
 * L = log2(N/2), cmul(x, y) = (fma(-x.im, y.im, x.re y.re), fma(x.im, y.re, x.re y.im)); the split into a
 * per-16-positions base and a 16th root of unity is what lets a GPU wave derive its 16 factors per lane
 * from one table entry.  The combine accumulates subsets in increasing s:
 *     KB[p] <- K_0[p];  KB[p] <- (fma(-k.im, m.im, fma(k.re, m.re, KB.re)), fma(k.im, m.re, fma(k.re, m.im, KB.im)))
 * exact engine: the reference's GPU backend combines in the STANDARD (u64) domain with exact monomial
 * products (backends/tfhe-cuda-backend/cuda/src/pbs/programmable_bootstrap_multibit.cuh:40-330); that
 * form is kept for the exact-integer verification engine, where it is error free.
 */
#include "tfhe_oracle.h"
#include "tfhe_oracle_internal.h"
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#endif
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

/* lwe_multi_bit_bootstrap_key_generation.rs:504-530 (combine_key_bits) */
static uint64_t combine_key_bits(uint32_t selector, const uint64_t *key_bits, uint32_t g) {
  uint64_t p = 1;
  for (uint32_t m = 0; m < g; ++m) {
    uint32_t pos = g - (m + 1);
    uint64_t inv = ((selector >> pos) & 1) ^ 1;
    p *= key_bits[m] ^ inv;
  }
  return p;
}


/* layout [n/g groups][2^g GGSWs][level l..1][k+1 rows][k+1 polys][N]
 * (cc/entities/lwe_multi_bit_bootstrap_key) */
void orc_gen_multi_bit_bsk(uint64_t seed, uint64_t *bsk, const uint64_t *lwe_sk, uint32_t n,
                           const uint64_t *glwe_sk, uint32_t k, uint32_t N, uint32_t base_log,
                           uint32_t level, uint32_t g, uint32_t noise_bound_log2) {
  size_t ggsw_sz = (size_t)level * (k + 1) * (k + 1) * N;
  uint32_t groups = n / g, per = 1u << g;
#pragma omp parallel for schedule(dynamic)
  for (uint32_t t = 0; t < groups * per; ++t) {
    uint32_t grp = t / per, s = t % per;
    orc_rng r;
    orc_rng_seed(&r, seed * 0x100000001B3ull + 0xABCD00 + t);
    uint64_t pt = combine_key_bits(s, lwe_sk + (size_t)grp * g, g);
    orc_ggsw_encrypt(&r, bsk + (size_t)t * ggsw_sz, pt, glwe_sk, k, N, base_log, level,
                     noise_bound_log2);
  }
}

/* lwe_multi_bit_programmable_bootstrapping.rs:30-65,78-114
 * degrees[grp*2^g + s] for s = 1..2^g-1 (slot s = 0 unused, set to 0) */
void orc_multi_bit_modulus_switch(const uint64_t *lwe, uint32_t n, uint32_t log_modulus, uint32_t g,
                                  uint64_t *degrees, uint64_t *body_hat) {
  uint32_t groups = n / g, per = 1u << g;
  for (uint32_t grp = 0; grp < groups; ++grp) {
    degrees[(size_t)grp * per] = 0;
    for (uint32_t s = 1; s < per; ++s) {
      uint64_t sum = 0;
      for (uint32_t m = 0; m < g; ++m) {
        uint32_t pos = g - (m + 1);
        if ((s >> pos) & 1) sum += lwe[(size_t)grp * g + m];
      }
      degrees[(size_t)grp * per + s] = orc_modulus_switch(sum, log_modulus);
    }
  }
  *body_hat = orc_modulus_switch(lwe[n], log_modulus);
}

/* keybundle = GGSW_0 + sum_{s>=1} GGSW_s * X^{deg_s}, exact mod 2^64 */
static void build_keybundle(uint64_t *kb, const uint64_t *group, const uint64_t *deg, uint32_t k,
                            uint32_t N, uint32_t level, uint32_t g, uint64_t *tmp) {
  size_t ggsw_sz = (size_t)level * (k + 1) * (k + 1) * N;
  size_t polys = ggsw_sz / N;
  memcpy(kb, group, sizeof(uint64_t) * ggsw_sz);
  for (uint32_t s = 1; s < (1u << g); ++s) {
    const uint64_t *src = group + (size_t)s * ggsw_sz;
    for (size_t p = 0; p < polys; ++p) {
      orc_monomial_mul(tmp, src + p * N, N, deg[s]);
      for (uint32_t j = 0; j < N; ++j) kb[p * N + j] += tmp[j];
    }
  }
}

static void multi_bit_core(uint64_t *lwe_out, const uint64_t *lwe_in,
                           const uint64_t *lut, const uint64_t *bsk_std, uint32_t n, uint32_t k,
                           uint32_t N, uint32_t base_log, uint32_t level, uint32_t g) {
  size_t gl = (size_t)(k + 1) * N;
  size_t ggsw_sz = (size_t)level * (k + 1) * gl;
  uint32_t groups = n / g, per = 1u << g;
  uint64_t *deg = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)groups * per);
  uint64_t body_hat;
  orc_multi_bit_modulus_switch(lwe_in, n, orc_log2_u32(2 * N), g, deg, &body_hat);

  uint64_t *buf = (uint64_t *)malloc(sizeof(uint64_t) * (gl * 3 + ggsw_sz + N));
  uint64_t *ct0 = buf, *ct1 = buf + gl, *states = buf + 2 * gl, *kb = buf + 3 * gl,
           *tmp = kb + ggsw_sz;
  int64_t *digits = (int64_t *)malloc(sizeof(int64_t) * gl);
  /* acc <- LUT * X^{-b_hat}  (:700-720 of the reference's blind rotate) */
  for (uint32_t p = 0; p <= k; ++p) orc_monomial_div(ct0 + (size_t)p * N, lut + (size_t)p * N, N, body_hat);

  uint64_t *src = ct0, *dst = ct1;
  for (uint32_t grp = 0; grp < groups; ++grp) {
    build_keybundle(kb, bsk_std + (size_t)grp * per * ggsw_sz, deg + (size_t)grp * per, k, N, level, g, tmp);
    memset(dst, 0, sizeof(uint64_t) * gl);
    orc_ext_product_exact(dst, src, kb, k, N, base_log, level, digits, states);
    uint64_t *t = src; src = dst; dst = t;
  }
  orc_sample_extract(lwe_out, src, k, N, 0);
  free(deg); free(buf); free(digits);
}

void orc_pbs_multi_bit_exact(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                             const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N,
                             uint32_t base_log, uint32_t level, uint32_t g) {
  multi_bit_core(lwe_out, lwe_in, lut, bsk_std, n, k, N, base_log, level, g);
}

/* Z[j] = e^{i pi j / N}, j < 2N: octant [0, N/4] evaluated, the rest by exact symmetries */
void orc_monomial_table(uint32_t N, double *z) {
  const long double PI = 3.14159265358979323846264338327950288L;
  for (uint32_t j = 0; j <= N / 4; ++j) {
    long double ang = PI * (long double)j / (long double)N;
    z[2 * j] = (double)cosl(ang);
    z[2 * j + 1] = (double)sinl(ang);
  }
  z[0] = 1.0; z[1] = 0.0;
  for (uint32_t j = N / 4 + 1; j <= N / 2; ++j) { z[2 * j] = z[2 * (N / 2 - j) + 1]; z[2 * j + 1] = z[2 * (N / 2 - j)]; }
  for (uint32_t j = N / 2 + 1; j <= N; ++j) { z[2 * j] = -z[2 * (j - N / 2) + 1]; z[2 * j + 1] = z[2 * (j - N / 2)]; }
  for (uint32_t j = N + 1; j < 2 * N; ++j) { z[2 * j] = -z[2 * (j - N)]; z[2 * j + 1] = -z[2 * (j - N) + 1]; }
}

static uint32_t bitrev_u32(uint32_t x, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

/* M_d[p] for p < N/2 (re, im interleaved).  The two index factors of position p depend on N alone: tabulated once per
 * bootstrap (the bit reversals cost more than the products they feed) */
static void monomial_index_tables(uint32_t N, uint32_t *tab_b, uint32_t *tab_w) {
  uint32_t n = N / 2, L = orc_log2_u32(n);
  for (uint32_t p = 0; p < n; ++p) { tab_b[p] = 1 + 4 * bitrev_u32(p >> 4, L - 4); tab_w[p] = bitrev_u32(p & 15, 4); }
}
static void monomial_fourier_tab(uint32_t N, uint64_t degree, const double *z, const uint32_t *tab_b, const uint32_t *tab_w,
                                 double *m) {
  const uint32_t n = N / 2, d = (uint32_t)(degree % (2 * N)), mask = 2 * N - 1;  /* 2N is a power of two */
  for (uint32_t p = 0; p < n; ++p) {
    uint32_t jb = (tab_b[p] * d) & mask;                 /* ((1 + 4 bitrev(p >> 4)) * degree) mod 2N */
    uint32_t jw = (N / 8) * ((tab_w[p] * d) & 15);       /* (N / 8) * ((bitrev(p & 15) * degree) mod 16) */
    double xr = z[2 * jb], xi = z[2 * jb + 1], yr = z[2 * jw], yi = z[2 * jw + 1];
    m[2 * p] = fma(-xi, yi, xr * yr);
    m[2 * p + 1] = fma(xi, yr, xr * yi);
  }
}
void orc_monomial_fourier(uint32_t N, uint64_t degree, const double *z, double *m) {
  uint32_t *tab = (uint32_t *)malloc(sizeof(uint32_t) * N);
  monomial_index_tables(N, tab, tab + N / 2);
  monomial_fourier_tab(N, degree, z, tab, tab + N / 2, m);
  free(tab);
}

/* standard-domain multi-bit key -> Fourier domain, same nesting [group][subset][level][row][col], each
 * polynomial N/2 complex in transform-position order (cc/algorithms/lwe_multi_bit_bootstrap_key_conversion.rs) */
void orc_convert_multi_bit_bsk_fft(double *bsk_f, const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N,
                                   uint32_t level, uint32_t g) {
  size_t polys = (size_t)(n / g) * ((size_t)1 << g) * level * (k + 1) * (k + 1);
  orc_fft_forward_torus(bsk_f, bsk_std, N); /* builds the plan outside the parallel region */
#pragma omp parallel for schedule(static)
  for (size_t p = 0; p < polys; ++p) orc_fft_forward_torus(bsk_f + p * N, bsk_std + p * N, N);
}

/* lwe_multi_bit_programmable_bootstrapping.rs:116-156 + :647-880, f64 engine; bsk_f from
 * orc_convert_multi_bit_bsk_fft */
void orc_pbs_multi_bit_fft(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                           const double *bsk_f, uint32_t n, uint32_t k, uint32_t N,
                           uint32_t base_log, uint32_t level, uint32_t g) {
  size_t gl = (size_t)(k + 1) * N;
  size_t ggsw_sz = (size_t)level * (k + 1) * gl; /* doubles per Fourier GGSW = u64 per standard GGSW */
  uint32_t groups = n / g, per = 1u << g, nn = N / 2;
  uint64_t *deg = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)groups * per);
  uint64_t body_hat;
  orc_multi_bit_modulus_switch(lwe_in, n, orc_log2_u32(2 * N), g, deg, &body_hat);
  uint64_t *buf = (uint64_t *)malloc(sizeof(uint64_t) * gl * 3);
  uint64_t *ct0 = buf, *ct1 = buf + gl, *states = buf + 2 * gl;
  int64_t *digits = (int64_t *)malloc(sizeof(int64_t) * gl);
  double *kb_f = (double *)malloc(sizeof(double) * ggsw_sz);
  double *fbuf = (double *)malloc(sizeof(double) * (N + gl));
  double *outbuf = fbuf + N;
  double *z = (double *)malloc(sizeof(double) * 4 * N);
  double *mono = (double *)malloc(sizeof(double) * N);
  orc_monomial_table(N, z);
  uint32_t *tab = (uint32_t *)malloc(sizeof(uint32_t) * N);
  monomial_index_tables(N, tab, tab + N / 2);
  for (uint32_t p = 0; p <= k; ++p) orc_monomial_div(ct0 + (size_t)p * N, lut + (size_t)p * N, N, body_hat);
  uint64_t *src = ct0, *dst = ct1;
  for (uint32_t grp = 0; grp < groups; ++grp) {
    const double *group = bsk_f + (size_t)grp * per * ggsw_sz;
    memcpy(kb_f, group, sizeof(double) * ggsw_sz);
    for (uint32_t s = 1; s < per; ++s) {
      const double *ks = group + (size_t)s * ggsw_sz;
      monomial_fourier_tab(N, deg[(size_t)grp * per + s], z, tab, tab + N / 2, mono);
      for (size_t poly = 0; poly < ggsw_sz / N; ++poly) {
        uint32_t p = 0;
#if defined(__AVX2__) && defined(__FMA__)
        /* two points per register, lane by lane the scalar operations below in the same order: inner = fma(kr, m, o),
         * then fma(ki, [-mi, mr], inner) — (-ki) * mi and ki * (-mi) are the same product */
        const __m256d sgn = _mm256_setr_pd(-0.0, 0.0, -0.0, 0.0);
        for (; p + 2 <= nn; p += 2) {
          const __m256d kv = _mm256_loadu_pd(ks + poly * N + 2 * p), mv = _mm256_loadu_pd(mono + 2 * p);
          double *o = kb_f + poly * N + 2 * p;
          const __m256d krr = _mm256_movedup_pd(kv), kii = _mm256_permute_pd(kv, 0xF);
          const __m256d msw = _mm256_xor_pd(_mm256_permute_pd(mv, 0x5), sgn);  /* [-mi mr] */
          const __m256d inner = _mm256_fmadd_pd(krr, mv, _mm256_loadu_pd(o));
          _mm256_storeu_pd(o, _mm256_fmadd_pd(kii, msw, inner));
        }
#endif
        for (; p < nn; ++p) {
          double kr = ks[poly * N + 2 * p], ki = ks[poly * N + 2 * p + 1];
          double mr = mono[2 * p], mi = mono[2 * p + 1];
          double *o = kb_f + poly * N + 2 * p;
          o[0] = fma(-ki, mi, fma(kr, mr, o[0]));
          o[1] = fma(ki, mr, fma(kr, mi, o[1]));
        }
      }
    }
    memset(dst, 0, sizeof(uint64_t) * gl);
    orc_ext_product_fft(dst, src, kb_f, k, N, base_log, level, states, digits, fbuf, outbuf);
    uint64_t *t = src; src = dst; dst = t;
  }
  orc_sample_extract(lwe_out, src, k, N, 0);
  free(deg); free(buf); free(digits); free(kb_f); free(fbuf); free(z); free(mono); free(tab);
}

void orc_pbs_multi_bit_fft_batch(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                                 const double *bsk_f, uint32_t n, uint32_t k, uint32_t N, uint32_t base_log,
                                 uint32_t level, uint32_t g, uint32_t count, uint32_t threads) {
  size_t out_sz = (size_t)k * N + 1;
  (void)threads;
#pragma omp parallel for schedule(dynamic) num_threads(threads ? (int)threads : omp_get_max_threads())
  for (uint32_t i = 0; i < count; ++i)
    orc_pbs_multi_bit_fft(lwe_out + (size_t)i * out_sz, lwe_in + (size_t)i * (n + 1), lut, bsk_f, n, k, N,
                          base_log, level, g);
}
