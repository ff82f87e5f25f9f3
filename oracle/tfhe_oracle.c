/*
 * tfhe_oracle.c — CPU restatement of the tfhe-rs core_crypto PBS hot path.
 * TEST INFRASTRUCTURE ONLY (see tfhe_oracle.h header comment).
 *
 * Reference paths are relative to the tfhe-rs tree; "cc/" abbreviates
 * tfhe/src/core_crypto/.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).  The f64
 * transform below is compiled with contraction OFF and uses explicit fma() so
 * that its operation order is exactly the one DESIGN.md §4 specifies.
 */
#include "tfhe_oracle.h"
#include "tfhe_oracle_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ PRNG */
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void orc_rng_seed(orc_rng *r, uint64_t seed) {
  /* splitmix64 expansion */
  for (int i = 0; i < 4; ++i) {
    seed += 0x9E3779B97F4A7C15ull;
    uint64_t z = seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    r->s[i] = z ^ (z >> 31);
  }
}

uint64_t orc_rng_next(orc_rng *r) {
  uint64_t *s = r->s;
  const uint64_t result = rotl64(s[1] * 5, 7) * 9;
  const uint64_t t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
  s[2] ^= t; s[3] = rotl64(s[3], 45);
  return result;
}

/* TUniform(b): integers in [-2^b, 2^b], the two ends with half weight
 * (cc/commons/math/random/t_uniform.rs:40-48 — distribution only; the byte
 * stream is ours). */
int64_t orc_rng_tuniform(orc_rng *r, uint32_t b) {
  uint64_t u = orc_rng_next(r) & ((1ull << (b + 2)) - 1);
  return (int64_t)((u + 1) >> 1) - ((int64_t)1 << b);
}

/* ------------------------------------------------------ modulus switch */
/* cc/fft_impl/common.rs:10-23 */
uint64_t orc_modulus_switch(uint64_t x, uint32_t log_modulus) {
  if (log_modulus == 64) return x;
  uint64_t t = x + (1ull << (64 - log_modulus - 1)); /* wrapping */
  return t >> (64 - log_modulus);
}

/* cc/algorithms/modulus_switch.rs:57-103 */
uint64_t orc_centered_ms_body_correction(const uint64_t *lwe, uint32_t n, uint32_t log_modulus) {
  uint64_t sum_half = 0;
  int64_t sum_halving_doubled = 0;
  for (uint32_t i = 0; i < n; ++i) {
    uint64_t a = lwe[i];
    uint64_t rounded = orc_modulus_switch(a, log_modulus) << (64 - log_modulus);
    int64_t err = (int64_t)(rounded - a);
    int64_t half = err / 2; /* truncation toward zero, as Rust's `/` */
    int64_t halving_doubled = 2 * half - err;
    sum_half += (uint64_t)half;
    sum_halving_doubled += halving_doubled;
  }
  uint64_t sum_halving = (uint64_t)(sum_halving_doubled / 2);
  sum_half -= sum_halving;
  uint64_t half_case = 1ull << (64 - log_modulus - 1);
  return sum_half - half_case;
}

/* cc/entities/modulus_switched_lwe_ciphertext.rs:143-176 */
void orc_lwe_modulus_switch(const uint64_t *lwe, uint32_t n, uint32_t log_modulus,
                            uint32_t ms_type, uint64_t *out) {
  uint64_t corr = ms_type == 1 ? orc_centered_ms_body_correction(lwe, n, log_modulus) : 0;
  for (uint32_t i = 0; i < n; ++i) out[i] = orc_modulus_switch(lwe[i], log_modulus);
  out[n] = orc_modulus_switch(lwe[n] + corr, log_modulus);
}

/* --------------------------------------------------------- decomposer */
/* cc/commons/math/decomposition/decomposer.rs:156-185 (+ :60-95 bit trick) */
uint64_t orc_decomp_init_state(uint64_t x, uint32_t base_log, uint32_t level) {
  uint32_t rep = base_log * level;
  uint32_t non_rep = 64 - rep;
  uint64_t res = x >> (non_rep - 1);
  uint64_t rounding_bit = res & 1;
  res += 1;
  res >>= 1;
  uint64_t mod_mask = ~0ull >> (64 - rep);
  res &= mod_mask;
  /* balanced_rounding_condition_bit_trick */
  uint64_t need_balance = (((res - 1) | (rounding_bit << (rep - 1))) & res) >> (rep - 1);
  return res - (need_balance << rep);
}

/* cc/commons/math/decomposition/decomposer.rs (native_closest_representable) */
uint64_t orc_closest_representable(uint64_t x, uint32_t base_log, uint32_t level) {
  uint32_t non_rep = 64 - base_log * level;
  uint64_t res = x >> (non_rep - 1);
  res += 1;
  res &= ~1ull;
  return res << (non_rep - 1);
}

/* cc/commons/math/decomposition/iter.rs:122-151 */
uint64_t orc_decompose_one_level(uint32_t base_log, uint64_t *state) {
  uint64_t mask = (1ull << base_log) - 1;
  uint64_t res = *state & mask;
  *state = (uint64_t)((int64_t)*state >> base_log);
  uint64_t carry = (((res - 1) | *state) & res) >> (base_log - 1);
  *state += carry;
  return res - (carry << base_log);
}

void orc_decompose(uint64_t x, uint32_t base_log, uint32_t level, int64_t *digits) {
  uint64_t st = orc_decomp_init_state(x, base_log, level);
  for (uint32_t i = 0; i < level; ++i) digits[i] = (int64_t)orc_decompose_one_level(base_log, &st);
}

/* -------------------------------------------------------- monomial ops */
/* cc/algorithms/polynomial_algorithms.rs:544-583 */
void orc_monomial_div(uint64_t *out, const uint64_t *in, uint32_t N, uint64_t degree) {
  uint32_t r = (uint32_t)(degree % N);
  int odd = (degree / N) & 1;
  for (uint32_t j = 0; j < N - r; ++j) out[j] = odd ? (uint64_t)0 - in[j + r] : in[j + r];
  for (uint32_t j = N - r; j < N; ++j) out[j] = odd ? in[j - (N - r)] : (uint64_t)0 - in[j - (N - r)];
}

/* cc/algorithms/polynomial_algorithms.rs (polynomial_wrapping_monic_monomial_mul) */
void orc_monomial_mul(uint64_t *out, const uint64_t *in, uint32_t N, uint64_t degree) {
  uint32_t r = (uint32_t)(degree % N);
  int odd = (degree / N) & 1;
  for (uint32_t j = 0; j < r; ++j) out[j] = odd ? in[N - r + j] : (uint64_t)0 - in[N - r + j];
  for (uint32_t j = r; j < N; ++j) out[j] = odd ? (uint64_t)0 - in[j - r] : in[j - r];
}

/* cc/algorithms/polynomial_algorithms.rs:662-727 */
void orc_monomial_mul_and_sub(uint64_t *out, const uint64_t *in, uint32_t N, uint64_t degree) {
  uint32_t r = (uint32_t)(degree % N);
  int odd = (degree / N) & 1;
  for (uint32_t j = 0; j < r; ++j) {
    uint64_t s = in[N - r + j];
    out[j] = (odd ? s : (uint64_t)0 - s) - in[j];
  }
  for (uint32_t j = r; j < N; ++j) {
    uint64_t s = in[j - r];
    out[j] = (odd ? (uint64_t)0 - s : s) - in[j];
  }
}

/* ------------------------------------- sample extract / keyswitch / LUT */
/* cc/algorithms/glwe_sample_extraction.rs:119-146 */
void orc_sample_extract(uint64_t *lwe_out, const uint64_t *glwe, uint32_t k, uint32_t N, uint32_t nth) {
  lwe_out[k * N] = glwe[k * N + nth];
  for (uint32_t p = 0; p < k; ++p) {
    const uint64_t *A = glwe + (size_t)p * N;
    uint64_t *o = lwe_out + (size_t)p * N;
    for (uint32_t j = 0; j < N; ++j) o[j] = A[j];
    /* reverse */
    for (uint32_t j = 0; j < N / 2; ++j) { uint64_t t = o[j]; o[j] = o[N - 1 - j]; o[N - 1 - j] = t; }
    uint32_t opp = N - nth - 1;
    for (uint32_t j = 0; j < opp; ++j) o[j] = (uint64_t)0 - o[j];
    /* rotate_left(opp) */
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * N);
    for (uint32_t j = 0; j < N; ++j) tmp[j] = o[(j + opp) % N];
    memcpy(o, tmp, sizeof(uint64_t) * N);
    free(tmp);
  }
}

/* cc/algorithms/lwe_keyswitch.rs:186-227 ; KSK layout [n_in][level l..1][n_out+1]
 * (cc/algorithms/lwe_keyswitch_key_generation.rs:165-195) */
void orc_keyswitch(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *ksk,
                   uint32_t n_in, uint32_t n_out, uint32_t base_log, uint32_t level) {
  memset(lwe_out, 0, sizeof(uint64_t) * (n_out + 1));
  lwe_out[n_out] = lwe_in[n_in];
  int64_t digits[64];
  for (uint32_t i = 0; i < n_in; ++i) {
    orc_decompose(lwe_in[i], base_log, level, digits);
    for (uint32_t lv = 0; lv < level; ++lv) {
      const uint64_t *row = ksk + ((size_t)i * level + lv) * (n_out + 1);
      uint64_t d = (uint64_t)digits[lv];
      for (uint32_t j = 0; j <= n_out; ++j) lwe_out[j] -= row[j] * d;
    }
  }
}

/* cc/algorithms/lwe_keyswitch.rs:331-447 (keyswitch_lwe_ciphertext_with_scalar_change, u64 -> u32: the
 * "KS32" atomic pattern, shortint/atomic_pattern/ks32.rs): the body is the input body rounded to the
 * output width (closest representable on 32 bits, one level, then shifted down), the mask digits come
 * from the u64 decomposer and multiply a u32 key, everything accumulates mod 2^32. */
void orc_keyswitch_64_32(uint32_t *lwe_out, const uint64_t *lwe_in, const uint32_t *ksk,
                         uint32_t n_in, uint32_t n_out, uint32_t base_log, uint32_t level) {
  memset(lwe_out, 0, sizeof(uint32_t) * (n_out + 1));
  lwe_out[n_out] = (uint32_t)(orc_closest_representable(lwe_in[n_in], 32, 1) >> 32);
  int64_t digits[64];
  for (uint32_t i = 0; i < n_in; ++i) {
    orc_decompose(lwe_in[i], base_log, level, digits);
    for (uint32_t lv = 0; lv < level; ++lv) {
      const uint32_t *row = ksk + ((size_t)i * level + lv) * (n_out + 1);
      uint32_t d = (uint32_t)(uint64_t)digits[lv];
      for (uint32_t j = 0; j <= n_out; ++j) lwe_out[j] -= row[j] * d;
    }
  }
}

/* cc/algorithms/lwe_programmable_bootstrapping/mod.rs:26-79 */
void orc_generate_lut(uint64_t *glwe_out, uint32_t k, uint32_t N, uint32_t message_modulus,
                      uint64_t delta, const uint64_t *f_table) {
  memset(glwe_out, 0, sizeof(uint64_t) * (size_t)(k + 1) * N);
  uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * N);
  uint32_t box = N / message_modulus;
  for (uint32_t i = 0; i < message_modulus; ++i)
    for (uint32_t j = 0; j < box; ++j) acc[i * box + j] = f_table[i] * delta;
  uint32_t half = box / 2;
  for (uint32_t j = 0; j < half; ++j) acc[j] = (uint64_t)0 - acc[j];
  uint64_t *body = glwe_out + (size_t)k * N;
  for (uint32_t j = 0; j < N; ++j) body[j] = acc[(j + half) % N];
  free(acc);
}

/* ------------------------------------------ exact negacyclic product */
/* out += small * big  mod (X^N+1, 2^64).  Plain schoolbook — the defining
 * formula (cc/algorithms/polynomial_algorithms.rs polynomial_wrapping_add_mul_assign). */
void orc_negacyclic_mul_add_naive(uint64_t *out, const int64_t *small, const uint64_t *big, uint32_t N) {
  for (uint32_t i = 0; i < N; ++i) {
    uint64_t a = (uint64_t)small[i];
    if (!a) continue;
    for (uint32_t j = 0; j < N; ++j) {
      uint32_t d = i + j;
      uint64_t p = a * big[j];
      if (d < N) out[d] += p; else out[d - N] -= p;
    }
  }
}

/* Karatsuba on plain (acyclic) products, then fold negacyclically
 * (cc/algorithms/polynomial_algorithms.rs:1106-1206 — same result as schoolbook
 * because everything is exact mod 2^64). */
static void kara_rec(uint64_t *res /* 2n, overwritten */, const uint64_t *a, const uint64_t *b,
                     uint32_t n, uint64_t *scratch) {
  if (n <= 32) {
    memset(res, 0, sizeof(uint64_t) * 2 * n);
    for (uint32_t i = 0; i < n; ++i) {
      uint64_t ai = a[i];
      for (uint32_t j = 0; j < n; ++j) res[i + j] += ai * b[j];
    }
    return;
  }
  uint32_t h = n / 2;
  uint64_t *sa = scratch, *sb = scratch + h, *mid = scratch + 2 * h, *next = scratch + 2 * h + 2 * h;
  kara_rec(res, a, b, h, next);
  kara_rec(res + n, a + h, b + h, h, next);
  for (uint32_t i = 0; i < h; ++i) { sa[i] = a[i] + a[i + h]; sb[i] = b[i] + b[i + h]; }
  kara_rec(mid, sa, sb, h, next);
  for (uint32_t i = 0; i < n; ++i) mid[i] -= res[i] + res[n + i];
  for (uint32_t i = 0; i < n; ++i) res[h + i] += mid[i];
}

void orc_negacyclic_mul_add(uint64_t *out, const int64_t *small, const uint64_t *big, uint32_t N) {
  uint64_t *buf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N * 8);
  uint64_t *res = buf, *scratch = buf + 2 * N;
  kara_rec(res, (const uint64_t *)small, big, N, scratch);
  for (uint32_t i = 0; i < N; ++i) out[i] += res[i] - res[i + N];
  free(buf);
}

/* --------------------------------------------- key generation / crypto */
void orc_gen_binary_key(orc_rng *r, uint64_t *sk, uint32_t len) {
  for (uint32_t i = 0; i < len; ++i) sk[i] = orc_rng_next(r) >> 63;
}

void orc_lwe_encrypt(orc_rng *r, uint64_t *ct, const uint64_t *sk, uint32_t n,
                     uint64_t plaintext, uint32_t noise_bound_log2) {
  uint64_t b = plaintext + (uint64_t)orc_rng_tuniform(r, noise_bound_log2);
  for (uint32_t i = 0; i < n; ++i) {
    ct[i] = orc_rng_next(r);
    if (sk[i]) b += ct[i];
  }
  ct[n] = b;
}

uint64_t orc_lwe_decrypt(const uint64_t *ct, const uint64_t *sk, uint32_t n) {
  uint64_t b = ct[n];
  for (uint32_t i = 0; i < n; ++i) if (sk[i]) b -= ct[i];
  return b;
}

/* body += sum_j A_j * S_j + E with fresh uniform masks
 * (cc/algorithms/glwe_encryption.rs encrypt_glwe_ciphertext_assign) */
void orc_glwe_encrypt_assign(orc_rng *r, uint64_t *glwe, const uint64_t *glwe_sk, uint32_t k,
                             uint32_t N, uint32_t noise_bound_log2) {
  uint64_t *body = glwe + (size_t)k * N;
  for (uint32_t j = 0; j < N; ++j) body[j] += (uint64_t)orc_rng_tuniform(r, noise_bound_log2);
  for (uint32_t p = 0; p < k; ++p) {
    uint64_t *A = glwe + (size_t)p * N;
    const uint64_t *S = glwe_sk + (size_t)p * N;
    for (uint32_t j = 0; j < N; ++j) A[j] = orc_rng_next(r);
    for (uint32_t t = 0; t < N; ++t) {
      if (!S[t]) continue;
      /* body += A * X^t */
      for (uint32_t j = 0; j < N - t; ++j) body[j + t] += A[j];
      for (uint32_t j = N - t; j < N; ++j) body[j + t - N] -= A[j];
    }
  }
}

/* One GGSW encrypting `cleartext` in the constant coefficient; layout
 * [level l..1][k+1 rows][k+1 polys][N]  (cc/algorithms/ggsw_encryption.rs:20-44,141-160,361-413) */
void orc_ggsw_encrypt(orc_rng *r, uint64_t *ggsw, uint64_t cleartext, const uint64_t *glwe_sk,
                         uint32_t k, uint32_t N, uint32_t base_log, uint32_t level,
                         uint32_t noise_bound_log2) {
  size_t row_sz = (size_t)(k + 1) * N;
  for (uint32_t idx = 0; idx < level; ++idx) {
    uint32_t lv = level - idx;
    uint64_t factor = ((uint64_t)0 - cleartext) << (64 - base_log * lv);
    for (uint32_t row = 0; row <= k; ++row) {
      uint64_t *g = ggsw + ((size_t)idx * (k + 1) + row) * row_sz;
      memset(g, 0, sizeof(uint64_t) * row_sz);
      uint64_t *body = g + (size_t)k * N;
      if (row < k) {
        const uint64_t *S = glwe_sk + (size_t)row * N;
        for (uint32_t j = 0; j < N; ++j) body[j] = S[j] * factor;
      } else {
        body[0] = (uint64_t)0 - factor;
      }
      orc_glwe_encrypt_assign(r, g, glwe_sk, k, N, noise_bound_log2);
    }
  }
}

/* BSK = n GGSWs of the LWE key bits, mask order
 * (cc/algorithms/lwe_bootstrap_key_generation.rs) */
void orc_gen_bsk(uint64_t seed, uint64_t *bsk, const uint64_t *lwe_sk, uint32_t n,
                 const uint64_t *glwe_sk, uint32_t k, uint32_t N, uint32_t base_log,
                 uint32_t level, uint32_t noise_bound_log2) {
  size_t ggsw_sz = (size_t)level * (k + 1) * (k + 1) * N;
#pragma omp parallel for schedule(dynamic)
  for (uint32_t i = 0; i < n; ++i) {
    orc_rng r;
    orc_rng_seed(&r, seed * 0x100000001B3ull + i);
    orc_ggsw_encrypt(&r, bsk + (size_t)i * ggsw_sz, lwe_sk[i], glwe_sk, k, N, base_log, level,
                 noise_bound_log2);
  }
}

/* cc/algorithms/lwe_keyswitch_key_generation.rs:165-195: block i, level index
 * idx (level l-idx) encrypts  -s_in[i] * 2^(64 - base_log*level)  ... the
 * reference encrypts +s*q/B^lvl and the keyswitch SUBTRACTS (lwe_keyswitch.rs) */
void orc_gen_ksk(uint64_t seed, uint64_t *ksk, const uint64_t *sk_in, uint32_t n_in,
                 const uint64_t *sk_out, uint32_t n_out, uint32_t base_log, uint32_t level,
                 uint32_t noise_bound_log2) {
#pragma omp parallel for schedule(dynamic, 16)
  for (uint32_t i = 0; i < n_in; ++i) {
    orc_rng r;
    orc_rng_seed(&r, seed * 0x100000001B3ull + 0x5151 + i);
    for (uint32_t idx = 0; idx < level; ++idx) {
      uint32_t lv = level - idx;
      uint64_t pt = sk_in[i] << (64 - base_log * lv);
      orc_lwe_encrypt(&r, ksk + ((size_t)i * level + idx) * (n_out + 1), sk_out, n_out, pt,
                      noise_bound_log2);
    }
  }
}

/* ----------------------------------------------- exact ("karatsuba") PBS */
/* acc += ct1 (x) GGSW, exact mod 2^64
 * (cc/algorithms/lwe_programmable_bootstrapping/karatsuba_pbs.rs:313-413) */
void orc_ext_product_exact(uint64_t *acc, const uint64_t *ct1, const uint64_t *ggsw, uint32_t k,
                              uint32_t N, uint32_t base_log, uint32_t level, int64_t *digit_buf,
                              uint64_t *states) {
  size_t gl = (size_t)(k + 1) * N;
  for (size_t j = 0; j < gl; ++j) states[j] = orc_decomp_init_state(ct1[j], base_log, level);
  for (uint32_t idx = 0; idx < level; ++idx) {
    for (size_t j = 0; j < gl; ++j) digit_buf[j] = (int64_t)orc_decompose_one_level(base_log, &states[j]);
    for (uint32_t row = 0; row <= k; ++row) {
      const uint64_t *grow = ggsw + ((size_t)idx * (k + 1) + row) * gl;
      for (uint32_t c = 0; c <= k; ++c)
        orc_negacyclic_mul_add(acc + (size_t)c * N, digit_buf + (size_t)row * N, grow + (size_t)c * N, N);
    }
  }
}

/* cc/fft_impl/fft64/crypto/bootstrap.rs:312-365 order (shared by karatsuba_pbs.rs:199-311) */
void orc_blind_rotate_exact(uint64_t *acc, const uint64_t *msed, const uint64_t *bsk_std, uint32_t n,
                            uint32_t k, uint32_t N, uint32_t base_log, uint32_t level) {
  size_t gl = (size_t)(k + 1) * N;
  size_t ggsw_sz = (size_t)level * (k + 1) * gl;
  uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * gl * 2);
  int64_t *digits = (int64_t *)malloc(sizeof(int64_t) * gl);
  uint64_t *ct1 = tmp, *states = tmp + gl;
  for (uint32_t p = 0; p <= k; ++p) {
    memcpy(ct1, acc + (size_t)p * N, sizeof(uint64_t) * N);
    orc_monomial_div(acc + (size_t)p * N, ct1, N, msed[n]);
  }
  for (uint32_t i = 0; i < n; ++i) {
    uint64_t a = msed[i];
    if (a == 0) continue;
    for (uint32_t p = 0; p <= k; ++p)
      orc_monomial_mul_and_sub(ct1 + (size_t)p * N, acc + (size_t)p * N, N, a);
    orc_ext_product_exact(acc, ct1, bsk_std + (size_t)i * ggsw_sz, k, N, base_log, level, digits, states);
  }
  free(tmp); free(digits);
}

uint32_t orc_log2_u32(uint32_t x) { uint32_t l = 0; while ((1u << l) < x) ++l; return l; }

void orc_pbs_exact(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                   const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N, uint32_t base_log,
                   uint32_t level, uint32_t ms_type) {
  size_t gl = (size_t)(k + 1) * N;
  uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * gl);
  uint64_t *msed = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
  memcpy(acc, lut, sizeof(uint64_t) * gl);
  orc_lwe_modulus_switch(lwe_in, n, orc_log2_u32(2 * N), ms_type, msed);
  orc_blind_rotate_exact(acc, msed, bsk_std, n, k, N, base_log, level);
  orc_sample_extract(lwe_out, acc, k, N, 0);
  free(acc); free(msed);
}

/* ------------------------------------------------------- Goldilocks NTT */
/* tfhe-ntt/src/prime64/generic_solinas.rs:77-129 */
/* Branch-free forms (the conditions of the reference's code are data-dependent coin flips: as branches they cost a
 * misprediction every other butterfly — 13 ns per butterfly against 4); the value returned is the same canonical
 * residue in [0, p) */
static inline uint64_t gl_add(uint64_t a, uint64_t b) {
  const uint64_t p = ORC_GOLDILOCKS_P;
  uint64_t s = a + b;                                   /* a, b < p < 2^64: at most one wrap */
  uint64_t wrapped = (uint64_t)0 - (uint64_t)(s < a);   /* all ones when a + b >= 2^64 */
  s -= wrapped & p;                                     /* 2^64 = p + (2^32 - 1): subtracting p mod 2^64 == adding 2^32 - 1 */
  s -= ((uint64_t)0 - (uint64_t)(s >= p)) & p;
  return s;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) {
  const uint64_t p = ORC_GOLDILOCKS_P;
  uint64_t d = a - b;
  return d + (((uint64_t)0 - (uint64_t)(a < b)) & p);
}
static inline uint64_t gl_mul(uint64_t a, uint64_t b) {
  const uint64_t p = ORC_GOLDILOCKS_P;
  u128 wide = (u128)a * b;
  uint64_t lo = (uint64_t)wide;
  uint64_t hi = (uint64_t)(wide >> 64);
  uint64_t mid = hi & 0xFFFFFFFFull;
  hi = hi >> 32;
  uint64_t low2 = lo - hi;
  low2 += ((uint64_t)0 - (uint64_t)(hi > lo)) & p;
  uint64_t product = (mid << 32) - mid;
  uint64_t result = low2 + product;
  result -= ((uint64_t)0 - (uint64_t)((result < product) | (result >= p))) & p;
  return result;
}
uint64_t orc_gl_add(uint64_t a, uint64_t b) { return gl_add(a, b); }
uint64_t orc_gl_sub(uint64_t a, uint64_t b) { return gl_sub(a, b); }
uint64_t orc_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a, b); }
uint64_t orc_gl_pow(uint64_t a, uint64_t e) {
  uint64_t r = 1;
  while (e) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); e >>= 1; }
  return r;
}

/* The primitive 2N-th root of unity the reference uses for the Goldilocks prime:
 * fixed constants of tfhe-ntt/src/prime64.rs:162-179 (so the NTT-domain key is the
 * reference's, value for value); other sizes fall back to 7^((p-1)/2N) (7 generates Z_p^*). */
uint64_t orc_gl_primitive_root_2N(uint32_t N) {
  switch (N) {
    case 32: return 8ull;
    case 64: return 2198989700608ull;
    case 128: return 14041890976876060974ull;
    case 256: return 14430643036723656017ull;
    case 512: return 4440654710286119610ull;
    case 1024: return 8816101479115663336ull;
    case 2048: return 10974926054405199669ull;
    case 4096: return 1206500561358145487ull;
    case 8192: return 10930245224889659871ull;
    default: return orc_gl_pow(7, (ORC_GOLDILOCKS_P - 1) / (2ull * N));
  }
}

static uint32_t bitrev(uint32_t x, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

typedef struct { uint32_t N; uint64_t *tw, *itw; uint64_t n_inv; } ntt_plan;
static ntt_plan g_plans[8];
static int g_nplans = 0;

static const ntt_plan *ntt_get_plan(uint32_t N) {
  const ntt_plan *res = NULL;
#pragma omp critical(orc_ntt_plan)
  {
    for (int i = 0; i < g_nplans; ++i) if (g_plans[i].N == N) res = &g_plans[i];
    if (!res) {
      ntt_plan *p = &g_plans[g_nplans];
      uint32_t lg = orc_log2_u32(N);
      p->N = N;
      p->tw = (uint64_t *)malloc(sizeof(uint64_t) * N);
      p->itw = (uint64_t *)malloc(sizeof(uint64_t) * N);
      uint64_t psi = orc_gl_primitive_root_2N(N);
      uint64_t psi_inv = orc_gl_pow(psi, ORC_GOLDILOCKS_P - 2);
      /* twiddles in bit-reversed order: tw[m + g] = psi^bitrev_lg(m + g) ... the
       * classic merged negacyclic layout (tfhe-ntt/src/prime64.rs:764-862) */
      for (uint32_t i = 0; i < N; ++i) {
        uint32_t e = bitrev(i, lg);
        p->tw[i] = orc_gl_pow(psi, e);
        p->itw[i] = orc_gl_pow(psi_inv, e);
      }
      p->n_inv = orc_gl_pow(N, ORC_GOLDILOCKS_P - 2);
      ++g_nplans;
      res = p;
    }
  }
  return res;
}

/* tfhe-ntt/src/prime64/generic_solinas.rs:449-481 (Cooley–Tukey, output bit-reversed) */
void orc_ntt_forward(uint64_t *data, uint32_t N) {
  const ntt_plan *pl = ntt_get_plan(N);
  uint32_t t = N / 2, m = 1;
  while (m < N) {
    for (uint32_t g = 0; g < m; ++g) {
      uint64_t w = pl->tw[m + g];
      uint64_t *z0 = data + (size_t)2 * g * t, *z1 = z0 + t;
      for (uint32_t j = 0; j < t; ++j) {
        uint64_t zw = gl_mul(z1[j], w);
        uint64_t a = z0[j];
        z0[j] = gl_add(a, zw);
        z1[j] = gl_sub(a, zw);
      }
    }
    t /= 2; m *= 2;
  }
}

/* tfhe-ntt/src/prime64/generic_solinas.rs:483-514 (Gentleman–Sande) */
void orc_ntt_inverse(uint64_t *data, uint32_t N) {
  const ntt_plan *pl = ntt_get_plan(N);
  uint32_t t = 1, m = N;
  while (m > 1) {
    m /= 2;
    for (uint32_t g = 0; g < m; ++g) {
      uint64_t w = pl->itw[m + g];
      uint64_t *z0 = data + (size_t)2 * g * t, *z1 = z0 + t;
      for (uint32_t j = 0; j < t; ++j) {
        uint64_t a = z0[j], b = z1[j];
        z0[j] = gl_add(a, b);
        z1[j] = gl_mul(gl_sub(a, b), w);
      }
    }
    t *= 2;
  }
}

/* tfhe-ntt/src/prime64.rs:1137-1179 */
void orc_ntt_normalize(uint64_t *data, uint32_t N) {
  const ntt_plan *pl = ntt_get_plan(N);
  for (uint32_t i = 0; i < N; ++i) data[i] = gl_mul(data[i], pl->n_inv);
}

/* cc/commons/math/ntt/ntt64.rs:144-160 with input_modulus_width = 64 */
uint64_t orc_modswitch_pow2_to_prime(uint64_t x) {
  u128 v = (u128)x * ORC_GOLDILOCKS_P + ((u128)1 << 63);
  return (uint64_t)(v >> 64);
}
/* cc/commons/math/ntt/ntt64.rs:162-177 with output_modulus_width = 64 */
uint64_t orc_modswitch_prime_to_pow2(uint64_t v) {
  u128 num = ((u128)v << 64) | (u128)(ORC_GOLDILOCKS_P >> 1);
  return (uint64_t)(num / ORC_GOLDILOCKS_P);
}

/* cc/algorithms/lwe_bootstrap_key_conversion.rs:367-434 (option Raw: modswitch, forward NTT,
 * no normalisation) */
void orc_convert_bsk_ntt(uint64_t *bsk_ntt, const uint64_t *bsk_std, uint32_t n, uint32_t k,
                         uint32_t N, uint32_t level) {
  size_t polys = (size_t)n * level * (k + 1) * (k + 1);
  ntt_get_plan(N);
#pragma omp parallel for schedule(static)
  for (size_t p = 0; p < polys; ++p) {
    uint64_t *o = bsk_ntt + p * N;
    const uint64_t *s = bsk_std + p * N;
    for (uint32_t j = 0; j < N; ++j) o[j] = orc_modswitch_pow2_to_prime(s[j]);
    orc_ntt_forward(o, N);
  }
}

/* cc/algorithms/lwe_programmable_bootstrapping/ntt64_bnf_pbs.rs:541-681 */
static void ext_product_ntt_bnf(uint64_t *acc, const uint64_t *ct1, const uint64_t *ggsw_ntt,
                                uint32_t k, uint32_t N, uint32_t base_log, uint32_t level,
                                uint64_t *states, uint64_t *polybuf, uint64_t *outbuf) {
  size_t gl = (size_t)(k + 1) * N;
  for (size_t j = 0; j < gl; ++j) states[j] = orc_decomp_init_state(ct1[j], base_log, level);
  memset(outbuf, 0, sizeof(uint64_t) * gl);
  for (uint32_t idx = 0; idx < level; ++idx) {
    for (uint32_t row = 0; row <= k; ++row) {
      /* forward_from_decomp: ntt64.rs:199-220 */
      for (uint32_t j = 0; j < N; ++j) {
        uint64_t d = orc_decompose_one_level(base_log, &states[(size_t)row * N + j]);
        polybuf[j] = ((int64_t)d < 0) ? d + ORC_GOLDILOCKS_P : d;
      }
      orc_ntt_forward(polybuf, N);
      const uint64_t *grow = ggsw_ntt + ((size_t)idx * (k + 1) + row) * gl;
      for (uint32_t c = 0; c <= k; ++c) {
        uint64_t *o = outbuf + (size_t)c * N;
        const uint64_t *g = grow + (size_t)c * N;
        /* mul_accumulate: tfhe-ntt/src/prime64.rs:1182-1222 */
        for (uint32_t j = 0; j < N; ++j) o[j] = gl_add(o[j], gl_mul(g[j], polybuf[j]));
      }
    }
  }
  for (uint32_t c = 0; c <= k; ++c) {
    uint64_t *o = outbuf + (size_t)c * N;
    orc_ntt_normalize(o, N);
    orc_ntt_inverse(o, N);
    for (uint32_t j = 0; j < N; ++j) acc[(size_t)c * N + j] += orc_modswitch_prime_to_pow2(o[j]);
  }
}

/* cc/algorithms/lwe_programmable_bootstrapping/ntt64_bnf_pbs.rs:208-280,469-539,683-705
 * NOTE the order: no initial division; ct1 = acc*X^a; ct1 -= acc; rotation by -b LAST. */
void orc_pbs_ntt_bnf(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                     const uint64_t *bsk_ntt, uint32_t n, uint32_t k, uint32_t N, uint32_t base_log,
                     uint32_t level, uint32_t ms_type) {
  size_t gl = (size_t)(k + 1) * N;
  size_t ggsw_sz = (size_t)level * (k + 1) * gl;
  uint64_t *buf = (uint64_t *)malloc(sizeof(uint64_t) * (gl * 4 + N + n + 1));
  uint64_t *acc = buf, *ct1 = buf + gl, *states = buf + 2 * gl, *outbuf = buf + 3 * gl,
           *polybuf = buf + 4 * gl, *msed = polybuf + N;
  memcpy(acc, lut, sizeof(uint64_t) * gl);
  orc_lwe_modulus_switch(lwe_in, n, orc_log2_u32(2 * N), ms_type, msed);
  for (uint32_t i = 0; i < n; ++i) {
    uint64_t a = msed[i];
    if (a == 0) continue;
    for (uint32_t p = 0; p <= k; ++p) {
      orc_monomial_mul(ct1 + (size_t)p * N, acc + (size_t)p * N, N, a);
      for (uint32_t j = 0; j < N; ++j) ct1[(size_t)p * N + j] -= acc[(size_t)p * N + j];
    }
    ext_product_ntt_bnf(acc, ct1, bsk_ntt + (size_t)i * ggsw_sz, k, N, base_log, level, states,
                        polybuf, outbuf);
  }
  for (uint32_t p = 0; p <= k; ++p) {
    memcpy(ct1, acc + (size_t)p * N, sizeof(uint64_t) * N);
    orc_monomial_div(acc + (size_t)p * N, ct1, N, msed[n]);
  }
  orc_sample_extract(lwe_out, acc, k, N, 0);
  free(buf);
}

/* ------------------------------------------- fixed-order f64 transform */
/*
 * Semantics = cc/fft_impl/fft64/math/fft/mod.rs:63-74,201-330,498-559 (fold the
 * N reals into N/2 complex, twist by e^{i*pi*j/N}, size-N/2 DFT; backward =
 * inverse DFT, conj-twist, 1/(N/2), fractional part, *2^64, round, wrapping
 * add).  The reference leaves the DFT's internal order to tfhe-fft's runtime
 * planner, so ITS bits are not reproducible; this restatement fixes
 * one order:
 *   forward : merged-twist decimation tree.  Node (d,g) over positions
 *             [g*m,(g+1)*m), m = n>>d, twiddle s = exp(i*pi*(1+4*bitrev_d(g))/2^(d+2));
 *             (a,b) -> (a + s*b, 2a - (a + s*b)) with the fma chains below.
 *             Output position p holds the evaluation at zeta^(1+4*bitrev(p)).
 *   backward: radix-2 DIT on that order with w = exp(-2*pi*i*j/m); stages m=2,4
 *             use plain add/sub, stages m>=8 the same fma butterfly; then
 *             t = y * (conj(twist_j)/n), frac via nearest-even rint, *2^64,
 *             rint, saturating i64 conversion, wrapping add.
 */
typedef struct { uint32_t N; double *fwd, *inv, *untw; } fft_plan;
static fft_plan g_fplans[8];
static int g_nfplans = 0;

static void fft_fill_tables(uint32_t N, double *fwd, double *inv, double *untw) {
  uint32_t n = N / 2, D = orc_log2_u32(n);
  const long double PI = 3.14159265358979323846264338327950288L;
  /* forward: even g evaluated, odd sibling = i * even (exact quarter turn) */
  fwd[0] = fwd[1] = 0.0;
  for (uint32_t d = 0; d < D; ++d)
    for (uint32_t g = 0; g < (1u << d); ++g) {
      double c, s;
      if (g & 1) { c = -fwd[2 * ((1u << d) + g - 1) + 1]; s = fwd[2 * ((1u << d) + g - 1)]; }
      else {
        uint32_t r = 1 + 4 * bitrev(g, d);
        long double ang = PI * (long double)r / (long double)(1u << (d + 2));
        c = (double)cosl(ang); s = (double)sinl(ang);
      }
      fwd[2 * ((1u << d) + g)] = c;
      fwd[2 * ((1u << d) + g) + 1] = s;
    }
  /* backward: w = exp(-2*pi*i*j/(2*half)); j >= half/2 derived as -i * w[j - half/2] */
  inv[0] = inv[1] = 0.0;
  for (uint32_t half = 1; half < n; half *= 2)
    for (uint32_t j = 0; j < half; ++j) {
      double c, s;
      if (j == 0) { c = 1.0; s = 0.0; }
      else if (half >= 2 && j >= half / 2) {
        c = inv[2 * (half + j - half / 2) + 1]; s = -inv[2 * (half + j - half / 2)];
      } else {
        long double ang = -PI * (long double)j / (long double)half;
        c = (double)cosl(ang); s = (double)sinl(ang);
      }
      inv[2 * (half + j)] = c;
      inv[2 * (half + j) + 1] = s;
    }
  /* untwist: conj(exp(i*pi*j/N))/n for j <= n/2, mirrored (cos <-> sin) above */
  for (uint32_t j = 0; j < n; ++j) {
    if (j <= n / 2) {
      long double ang = PI * (long double)j / (long double)N;
      untw[2 * j] = (double)cosl(ang) / (double)n;     /* power-of-two scaling: exact */
      untw[2 * j + 1] = -((double)sinl(ang)) / (double)n;
    } else {
      untw[2 * j] = -untw[2 * (n - j) + 1];
      untw[2 * j + 1] = -untw[2 * (n - j)];
    }
  }
}

void orc_fft_tables(uint32_t N, double *fwd, double *inv, double *untw) { fft_fill_tables(N, fwd, inv, untw); }

static const fft_plan *fft_get_plan(uint32_t N) {
  const fft_plan *res = NULL;
#pragma omp critical(orc_fft_plan)
  {
    for (int i = 0; i < g_nfplans; ++i) if (g_fplans[i].N == N) res = &g_fplans[i];
    if (!res) {
      fft_plan *p = &g_fplans[g_nfplans];
      p->N = N;
      p->fwd = (double *)malloc(sizeof(double) * N);
      p->inv = (double *)malloc(sizeof(double) * N);
      p->untw = (double *)malloc(sizeof(double) * N);
      fft_fill_tables(N, p->fwd, p->inv, p->untw);
      ++g_nfplans;
      res = p;
    }
  }
  return res;
}

/* (a,b) -> (a + s*b, 2a - (a + s*b)) */
static inline void bfly(double *a, double *b, double sr, double si) {
  double ar = a[0], ai = a[1], br = b[0], bi = b[1];
  double o1r = fma(-bi, si, fma(br, sr, ar));
  double o1i = fma(bi, sr, fma(br, si, ai));
  a[0] = o1r; a[1] = o1i;
  b[0] = fma(2.0, ar, -o1r);
  b[1] = fma(2.0, ai, -o1i);
}

/* Two butterflies at a time on AVX2 (two complex points per register): lane by lane the SAME operations as bfly()
 * in the same order — inner fma(br, s?, a?), outer fma(-+bi, s?, inner), then fma(2, a, -o1) — so the results are
 * the scalar ones bit for bit (tests/test_oracle_pins.py pins them to the reference's golden vectors). */
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
/* a = [a0r a0i a1r a1i], b likewise; s_in = [s0r s0i s1r s1i] (inner multipliers), s_out = [-s0i s0r -s1i s1r] */
static inline void bfly2(double *pa, double *pb, __m256d s_in, __m256d s_out) {
  const __m256d a = _mm256_loadu_pd(pa), b = _mm256_loadu_pd(pb);
  const __m256d brr = _mm256_movedup_pd(b);        /* [br br br' br'] */
  const __m256d bii = _mm256_permute_pd(b, 0xF);   /* [bi bi bi' bi'] */
  const __m256d inner = _mm256_fmadd_pd(brr, s_in, a);
  const __m256d o1 = _mm256_fmadd_pd(bii, s_out, inner);
  _mm256_storeu_pd(pa, o1);
  _mm256_storeu_pd(pb, _mm256_fmsub_pd(_mm256_set1_pd(2.0), a, o1));
}
#define ORC_HAVE_AVX2 1
#else
#define ORC_HAVE_AVX2 0
#endif

static void fft_forward_inplace(double *v, uint32_t N) {
  const fft_plan *pl = fft_get_plan(N);
  uint32_t n = N / 2;
  for (uint32_t m = n, cnt = 1; m >= 2; m /= 2, cnt *= 2) {
    uint32_t half = m / 2;
#if ORC_HAVE_AVX2
    if (half >= 2) {  /* one twiddle per group, two consecutive j per step */
      for (uint32_t g = 0; g < cnt; ++g) {
        const double sr = pl->fwd[2 * (cnt + g)], si = pl->fwd[2 * (cnt + g) + 1];
        const __m256d s_in = _mm256_setr_pd(sr, si, sr, si), s_out = _mm256_setr_pd(-si, sr, -si, sr);
        double *base = v + 2 * (size_t)g * m;
        for (uint32_t j = 0; j < half; j += 2) bfly2(base + 2 * j, base + 2 * (j + half), s_in, s_out);
      }
      continue;
    }
    if (cnt >= 2) {  /* half == 1: groups g, g+1 = points [a b a' b'] in memory, one twiddle each */
      for (uint32_t g = 0; g < cnt; g += 2) {
        double *base = v + 4 * (size_t)g;
        const __m256d x = _mm256_loadu_pd(base), y = _mm256_loadu_pd(base + 4);  /* [a b], [a' b'] */
        const __m256d a = _mm256_permute2f128_pd(x, y, 0x20), b = _mm256_permute2f128_pd(x, y, 0x31);
        const double *t = pl->fwd + 2 * (cnt + g);  /* s_g, s_{g+1} */
        const __m256d s_in = _mm256_loadu_pd(t);
        const __m256d s_out = _mm256_xor_pd(_mm256_permute_pd(s_in, 0x5), _mm256_setr_pd(-0.0, 0.0, -0.0, 0.0));
        const __m256d brr = _mm256_movedup_pd(b), bii = _mm256_permute_pd(b, 0xF);
        const __m256d inner = _mm256_fmadd_pd(brr, s_in, a);
        const __m256d o1 = _mm256_fmadd_pd(bii, s_out, inner);
        const __m256d o2 = _mm256_fmsub_pd(_mm256_set1_pd(2.0), a, o1);
        _mm256_storeu_pd(base, _mm256_permute2f128_pd(o1, o2, 0x20));
        _mm256_storeu_pd(base + 4, _mm256_permute2f128_pd(o1, o2, 0x31));
      }
      continue;
    }
#endif
    for (uint32_t g = 0; g < cnt; ++g) {
      double sr = pl->fwd[2 * (cnt + g)], si = pl->fwd[2 * (cnt + g) + 1];
      double *base = v + 2 * (size_t)g * m;
      for (uint32_t j = 0; j < half; ++j) bfly(base + 2 * j, base + 2 * (j + half), sr, si);
    }
  }
}

static void fft_inverse_inplace(double *v, uint32_t N) {
  const fft_plan *pl = fft_get_plan(N);
  uint32_t n = N / 2;
  for (uint32_t half = 1; half < n; half *= 2) {
    uint32_t m = 2 * half;
#if ORC_HAVE_AVX2
    if (half >= 4) {  /* twiddle inv[half + j] per j: two consecutive j per step */
      for (uint32_t q = 0; q < n / m; ++q) {
        double *base = v + 2 * (size_t)q * m;
        for (uint32_t j = 0; j < half; j += 2) {
          const __m256d s_in = _mm256_loadu_pd(pl->inv + 2 * (half + j));
          const __m256d s_out = _mm256_xor_pd(_mm256_permute_pd(s_in, 0x5), _mm256_setr_pd(-0.0, 0.0, -0.0, 0.0));
          bfly2(base + 2 * j, base + 2 * (j + half), s_in, s_out);
        }
      }
      continue;
    }
#endif
    for (uint32_t q = 0; q < n / m; ++q) {
      double *base = v + 2 * (size_t)q * m;
      for (uint32_t j = 0; j < half; ++j) {
        double *a = base + 2 * j, *b = base + 2 * (j + half);
        if (half == 1) {
          double ar = a[0], ai = a[1], br = b[0], bi = b[1];
          a[0] = ar + br; a[1] = ai + bi; b[0] = ar - br; b[1] = ai - bi;
        } else if (half == 2) {
          double ar = a[0], ai = a[1], br = b[0], bi = b[1];
          if (j == 0) { a[0] = ar + br; a[1] = ai + bi; b[0] = ar - br; b[1] = ai - bi; }
          else { /* w = -i : w*b = (bi, -br) */
            a[0] = ar + bi; a[1] = ai - br; b[0] = ar - bi; b[1] = ai + br;
          }
        } else {
          bfly(a, b, pl->inv[2 * (half + j)], pl->inv[2 * (half + j) + 1]);
        }
      }
    }
  }
}

/* f64 -> i64 of an integer-valued double in [-2^63, 2^63]: +2^63 folds onto -2^63, the
 * two's-complement torus value (what the reference's SIMD conversions do and its test
 * fft/x86.rs:1030-1112 asserts; the scalar `as i64` would saturate to 2^63-1 instead). */
int64_t orc_f64_to_i64_sat(double x) {
  if (x >= 9223372036854775808.0) return INT64_MIN;
  if (x <= -9223372036854775808.0) return INT64_MIN;
  return (int64_t)x;
}

/* cc/commons/math/torus/mod.rs:73-79 with round := nearest-even (the reference's
 * AVX-512 path, fft/x86.rs:543-553; its scalar path rounds half away — differs
 * only on exact .5 ties) */
uint64_t orc_from_torus(double t) {
  double f = t - rint(t);
  f = f * 18446744073709551616.0;
  f = rint(f);
  return (uint64_t)orc_f64_to_i64_sat(f);
}

void orc_fft_forward_int(double *out, const int64_t *digits, uint32_t N) {
  uint32_t n = N / 2;
  for (uint32_t j = 0; j < n; ++j) { out[2 * j] = (double)digits[j]; out[2 * j + 1] = (double)digits[j + n]; }
  fft_forward_inplace(out, N);
}

/* fft/mod.rs:201-222 (convert_forward_torus): signed value * 2^-64 */
void orc_fft_forward_torus(double *out, const uint64_t *poly, uint32_t N) {
  uint32_t n = N / 2;
  const double norm = 5.421010862427522e-20; /* 2^-64 */
  for (uint32_t j = 0; j < n; ++j) {
    out[2 * j] = (double)(int64_t)poly[j] * norm;
    out[2 * j + 1] = (double)(int64_t)poly[j + n] * norm;
  }
  fft_forward_inplace(out, N);
}

/* The transform on plain f64 data, "compressed" polynomials (complex[i] = (p[i], p[i + N/2])): what the reference's backend
   tests drive through cuda_forward_fft_classic_async / cuda_fourier_polynomial_mul_async
   (backends/tfhe-cuda-backend/cuda/include/pbs/programmable_bootstrap.h:8-24; src/fft/bnsmfft.cuh:520-545, :695-768).
   The spectrum comes out in the tree order of DESIGN.md's transform spec = the native order of the reference's NSMFFT_direct:
   natural frequency f at index bitreverse((n - f) mod n) (tests/test_fourier_entry_points.py pins it on the reference's
   golden spectrum). */
void orc_fft_forward_f64(double *out, const double *in, uint32_t N) {
  memcpy(out, in, sizeof(double) * N);
  fft_forward_inplace(out, N);
}
/* negacyclic product of two compressed polynomials: pointwise product of the spectra, backward transform, untwist (1/n in
   the untwist table) — batch_polynomial_mul's data flow */
void orc_fft_polynomial_mul_f64(double *out, const double *a, const double *b, uint32_t N) {
  const fft_plan *pl = fft_get_plan(N);
  uint32_t n = N / 2;
  double *fa = (double *)malloc(sizeof(double) * N), *fb = (double *)malloc(sizeof(double) * N);
  orc_fft_forward_f64(fa, a, N);
  orc_fft_forward_f64(fb, b, N);
  for (uint32_t j = 0; j < n; ++j) {  /* fb * fa, the product order of the device's cmul_first(x, y) */
    double xr = fb[2 * j], xi = fb[2 * j + 1], yr = fa[2 * j], yi = fa[2 * j + 1];
    fb[2 * j] = fma(-xi, yi, xr * yr);
    fb[2 * j + 1] = fma(xi, yr, xr * yi);
  }
  fft_inverse_inplace(fb, N);
  for (uint32_t j = 0; j < n; ++j) {
    double yr = fb[2 * j], yi = fb[2 * j + 1], ur = pl->untw[2 * j], ui = pl->untw[2 * j + 1];
    out[2 * j] = fma(-yi, ui, yr * ur);
    out[2 * j + 1] = fma(yi, ur, yr * ui);
  }
  free(fa);
  free(fb);
}

/* fft/mod.rs:311-330 (convert_add_backward_torus) */
void orc_fft_backward_add(uint64_t *poly, double *fourier, uint32_t N) {
  const fft_plan *pl = fft_get_plan(N);
  uint32_t n = N / 2;
  fft_inverse_inplace(fourier, N);
#if ORC_HAVE_AVX2
  /* two points per step; untwist and orc_from_torus lane by lane as below (rint = round to nearest even) */
  const __m256d sgn = _mm256_setr_pd(-0.0, 0.0, -0.0, 0.0), two64 = _mm256_set1_pd(18446744073709551616.0);
  for (uint32_t j = 0; j < n; j += 2) {
    const __m256d y = _mm256_loadu_pd(fourier + 2 * j), u = _mm256_loadu_pd(pl->untw + 2 * j);
    const __m256d yrr = _mm256_movedup_pd(y), yii = _mm256_permute_pd(y, 0xF);
    const __m256d us = _mm256_xor_pd(_mm256_permute_pd(u, 0x5), sgn);           /* [-ui ur] */
    const __m256d t = _mm256_fmadd_pd(yii, us, _mm256_mul_pd(yrr, u));          /* [tr ti tr' ti'] */
    __m256d f = _mm256_sub_pd(t, _mm256_round_pd(t, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
    f = _mm256_round_pd(_mm256_mul_pd(f, two64), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
    double w[4];
    _mm256_storeu_pd(w, f);
    poly[j] += (uint64_t)orc_f64_to_i64_sat(w[0]);
    poly[j + n] += (uint64_t)orc_f64_to_i64_sat(w[1]);
    poly[j + 1] += (uint64_t)orc_f64_to_i64_sat(w[2]);
    poly[j + 1 + n] += (uint64_t)orc_f64_to_i64_sat(w[3]);
  }
#else
  for (uint32_t j = 0; j < n; ++j) {
    double yr = fourier[2 * j], yi = fourier[2 * j + 1];
    double ur = pl->untw[2 * j], ui = pl->untw[2 * j + 1];
    double tr = fma(-yi, ui, yr * ur);
    double ti = fma(yi, ur, yr * ui);
    poly[j] += orc_from_torus(tr);
    poly[j + n] += orc_from_torus(ti);
  }
#endif
}

/* cc/algorithms/lwe_bootstrap_key_conversion.rs:20-150 ; layout stays
 * [n][level][row][poly] with each polynomial now N/2 complex (N doubles) */
void orc_convert_bsk_fft(double *bsk_f, const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N,
                         uint32_t level) {
  size_t polys = (size_t)n * level * (k + 1) * (k + 1);
  fft_get_plan(N);
#pragma omp parallel for schedule(static)
  for (size_t p = 0; p < polys; ++p) orc_fft_forward_torus(bsk_f + p * N, bsk_std + p * N, N);
}

/* cc/fft_impl/fft64/crypto/ggsw.rs:483-602 + :616-697 */
void orc_ext_product_fft(uint64_t *acc, const uint64_t *ct1, const double *ggsw_f, uint32_t k,
                            uint32_t N, uint32_t base_log, uint32_t level, uint64_t *states,
                            int64_t *digits, double *fbuf, double *outbuf) {
  size_t gl = (size_t)(k + 1) * N;
  uint32_t n = N / 2;
  for (size_t j = 0; j < gl; ++j) states[j] = orc_decomp_init_state(ct1[j], base_log, level);
  int first = 1;
  for (uint32_t idx = 0; idx < level; ++idx) {
    for (uint32_t row = 0; row <= k; ++row) {
      for (uint32_t j = 0; j < N; ++j)
        digits[j] = (int64_t)orc_decompose_one_level(base_log, &states[(size_t)row * N + j]);
      orc_fft_forward_int(fbuf, digits, N);
      const double *grow = ggsw_f + ((size_t)idx * (k + 1) + row) * gl;
      for (uint32_t c = 0; c <= k; ++c) {
        double *o = outbuf + (size_t)c * N;
        const double *g = grow + (size_t)c * N;
#if ORC_HAVE_AVX2
        /* two points per step, lane by lane the scalar operations below in the same order */
        const __m256d sgn = _mm256_setr_pd(-0.0, 0.0, -0.0, 0.0);
        for (uint32_t j = 0; j < n; j += 2) {
          const __m256d x = _mm256_loadu_pd(fbuf + 2 * j), y = _mm256_loadu_pd(g + 2 * j);
          const __m256d xrr = _mm256_movedup_pd(x), xii = _mm256_permute_pd(x, 0xF);
          const __m256d ys = _mm256_xor_pd(_mm256_permute_pd(y, 0x5), sgn);  /* [-yi yr] */
          const __m256d inner = first ? _mm256_mul_pd(xrr, y) : _mm256_fmadd_pd(xrr, y, _mm256_loadu_pd(o + 2 * j));
          _mm256_storeu_pd(o + 2 * j, _mm256_fmadd_pd(xii, ys, inner));
        }
#else
        for (uint32_t j = 0; j < n; ++j) {
          double xr = fbuf[2 * j], xi = fbuf[2 * j + 1], yr = g[2 * j], yi = g[2 * j + 1];
          if (first) {
            o[2 * j] = fma(-xi, yi, xr * yr);
            o[2 * j + 1] = fma(xi, yr, xr * yi);
          } else {
            o[2 * j] = fma(-xi, yi, fma(xr, yr, o[2 * j]));
            o[2 * j + 1] = fma(xi, yr, fma(xr, yi, o[2 * j + 1]));
          }
        }
#endif
      }
      first = 0;
    }
  }
  for (uint32_t c = 0; c <= k; ++c) orc_fft_backward_add(acc + (size_t)c * N, outbuf + (size_t)c * N, N);
}

/* cc/fft_impl/fft64/crypto/bootstrap.rs:294-380,480-520 */
void orc_pbs_fft(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut, const double *bsk_f,
                 uint32_t n, uint32_t k, uint32_t N, uint32_t base_log, uint32_t level,
                 uint32_t ms_type) {
  size_t gl = (size_t)(k + 1) * N;
  size_t ggsw_sz = (size_t)level * (k + 1) * gl;
  uint64_t *buf = (uint64_t *)malloc(sizeof(uint64_t) * (gl * 3 + n + 1));
  uint64_t *acc = buf, *ct1 = buf + gl, *states = buf + 2 * gl, *msed = buf + 3 * gl;
  int64_t *digits = (int64_t *)malloc(sizeof(int64_t) * N);
  double *fbuf = (double *)malloc(sizeof(double) * (N + gl));
  double *outbuf = fbuf + N;
  memcpy(acc, lut, sizeof(uint64_t) * gl);
  orc_lwe_modulus_switch(lwe_in, n, orc_log2_u32(2 * N), ms_type, msed);
  for (uint32_t p = 0; p <= k; ++p) {
    memcpy(ct1, acc + (size_t)p * N, sizeof(uint64_t) * N);
    orc_monomial_div(acc + (size_t)p * N, ct1, N, msed[n]);
  }
  for (uint32_t i = 0; i < n; ++i) {
    uint64_t a = msed[i];
    if (a == 0) continue;
    for (uint32_t p = 0; p <= k; ++p)
      orc_monomial_mul_and_sub(ct1 + (size_t)p * N, acc + (size_t)p * N, N, a);
    orc_ext_product_fft(acc, ct1, bsk_f + (size_t)i * ggsw_sz, k, N, base_log, level, states, digits,
                    fbuf, outbuf);
  }
  orc_sample_extract(lwe_out, acc, k, N, 0);
  free(buf); free(digits); free(fbuf);
}

/* ---------------------------------------------------------------- batches */
uint32_t orc_max_threads(void) {
#ifdef _OPENMP
  return (uint32_t)omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_pbs_batch(uint32_t engine, uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                   const void *bsk, uint32_t n, uint32_t k, uint32_t N, uint32_t base_log,
                   uint32_t level, uint32_t ms_type, uint32_t count, uint32_t threads) {
  size_t out_sz = (size_t)k * N + 1;
  if (engine == 1) ntt_get_plan(N);
  if (engine == 2) fft_get_plan(N);
#ifdef _OPENMP
  if (threads == 0) threads = (uint32_t)omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic) num_threads(threads)
  for (uint32_t i = 0; i < count; ++i) {
    const uint64_t *in = lwe_in + (size_t)i * (n + 1);
    uint64_t *out = lwe_out + (size_t)i * out_sz;
    if (engine == 0) orc_pbs_exact(out, in, lut, (const uint64_t *)bsk, n, k, N, base_log, level, ms_type);
    else if (engine == 1) orc_pbs_ntt_bnf(out, in, lut, (const uint64_t *)bsk, n, k, N, base_log, level, ms_type);
    else orc_pbs_fft(out, in, lut, (const double *)bsk, n, k, N, base_log, level, ms_type);
  }
}

void orc_keyswitch_batch(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *ksk,
                         uint32_t n_in, uint32_t n_out, uint32_t base_log, uint32_t level,
                         uint32_t count, uint32_t threads) {
#ifdef _OPENMP
  if (threads == 0) threads = (uint32_t)omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic) num_threads(threads)
  for (uint32_t i = 0; i < count; ++i)
    orc_keyswitch(lwe_out + (size_t)i * (n_out + 1), lwe_in + (size_t)i * (n_in + 1), ksk, n_in,
                  n_out, base_log, level);
}
