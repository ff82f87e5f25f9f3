/*
 * tfhe_oracle_multibit.c — multi-bit PBS restatement.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows cc/algorithms/lwe_multi_bit_programmable_bootstrapping.rs (deterministic
 * variant, :647-880) and cc/algorithms/lwe_multi_bit_bootstrap_key_generation.rs.
 *
 * The CPU reference combines the group's GGSWs in the FOURIER domain
 * (prepare_multi_bit_ggsw_mem_optimized :116-156); the reference's GPU backend
 * combines them in the STANDARD (u64) domain with exact monomial products and
 * transforms afterwards (backends/tfhe-cuda-backend/cuda/src/pbs/
 * programmable_bootstrap_multibit.cuh:40-330).  We restate the latter (it is the
 * interface we replace, and it is exact up to the final transform); SURVEY D7
 * explains why raw bits cannot be compared with the CPU Fourier-combine anyway.
 */
#include "tfhe_oracle.h"
#include "tfhe_oracle_internal.h"
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

static void multi_bit_core(int use_fft, uint64_t *lwe_out, const uint64_t *lwe_in,
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
  double *kb_f = NULL, *fbuf = NULL, *outbuf = NULL;
  if (use_fft) {
    kb_f = (double *)malloc(sizeof(double) * ggsw_sz);
    fbuf = (double *)malloc(sizeof(double) * (N + gl));
    outbuf = fbuf + N;
  }
  /* acc <- LUT * X^{-b_hat}  (:700-720 of the reference's blind rotate) */
  for (uint32_t p = 0; p <= k; ++p) orc_monomial_div(ct0 + (size_t)p * N, lut + (size_t)p * N, N, body_hat);

  uint64_t *src = ct0, *dst = ct1;
  for (uint32_t grp = 0; grp < groups; ++grp) {
    build_keybundle(kb, bsk_std + (size_t)grp * per * ggsw_sz, deg + (size_t)grp * per, k, N, level, g, tmp);
    memset(dst, 0, sizeof(uint64_t) * gl);
    if (use_fft) {
      for (size_t p = 0; p < ggsw_sz / N; ++p) orc_fft_forward_torus(kb_f + p * N, kb + p * N, N);
      orc_ext_product_fft(dst, src, kb_f, k, N, base_log, level, states, digits, fbuf, outbuf);
    } else {
      orc_ext_product_exact(dst, src, kb, k, N, base_log, level, digits, states);
    }
    uint64_t *t = src; src = dst; dst = t;
  }
  orc_sample_extract(lwe_out, src, k, N, 0);
  free(deg); free(buf); free(digits);
  if (use_fft) { free(kb_f); free(fbuf); }
}

void orc_pbs_multi_bit_exact(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                             const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N,
                             uint32_t base_log, uint32_t level, uint32_t g) {
  multi_bit_core(0, lwe_out, lwe_in, lut, bsk_std, n, k, N, base_log, level, g);
}

void orc_pbs_multi_bit_fft(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                           const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N,
                           uint32_t base_log, uint32_t level, uint32_t g) {
  multi_bit_core(1, lwe_out, lwe_in, lut, bsk_std, n, k, N, base_log, level, g);
}
