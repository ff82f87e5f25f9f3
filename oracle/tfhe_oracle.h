/*
 * tfhe_oracle.h — CPU restatement of the tfhe-rs core_crypto PBS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (tfhe_rs_amd/) never links,
 * imports or falls back to it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the tfhe-rs tree).
 *
 * PARITY STATUS:
 *   - PINNED to bytes the reference produced: tests/test_reference_kat.py regenerates the
 *     reference's golden PBS test vectors (apps/test-vectors, BOTH parameter sets — toy and the
 *     production-size valid_params_128 with Gaussian noise: keys, KSK, BSK,
 *     keyswitch, modulus switch, blind rotation with exact products, sample extraction, id and
 *     2x LUTs) through THIS oracle and matches all 14 integer-path SHA-256 digests per set of
 *     apps/test-vectors/checksums.sha256 (the .cbor payloads are Git-LFS stubs, the digests are
 *     not).  This pins modulus switch, decomposer, monomial ops, keyswitch, GGSW/GLWE encryption
 *     layout, blind rotation order and sample extraction.
 *   - PINNED IN PHASE to bytes the reference's GPU backend produced at the BASELINE parameter sets:
 *     tests/test_pbs_golden.py regenerates keys, bootstrap keys and inputs of the reference's GPU golden-value
 *     test (tfhe/src/core_crypto/gpu/algorithms/test/pbs_golden: PARAM_MESSAGE_2_CARRY_2 and multi-bit g = 4,
 *     captured on an H100) from its fixed seed; the golden ciphertexts decrypt under the regenerated key and
 *     the exact and f64 engines of this oracle (classic AND multi-bit) land within 2^51 of them in phase.
 *   - integer pieces are additionally pinned against the value tables / doc-test vectors of the
 *     reference's unit tests (tests/golden/reference_kats.json, tests/golden/make_golden.py).
 *   - NTT path: exact ring arithmetic over the pinned pieces + the reference's fixed Goldilocks
 *     roots; no reference output bytes exist for it ("unpinned" beyond its pieces).
 *   - f64 FFT path: the reference's own bits are not reproducible (runtime-planned FFT order,
 *     SURVEY D3); pinned only to this file's fixed operation order plus the phase tolerance
 *     against the pinned exact path.
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- seeded PRNG (xoshiro256**); parity is on identical inputs, not on the
 *      reference's AES-CTR csprng ---- */
typedef struct { uint64_t s[4]; } orc_rng;
void     orc_rng_seed(orc_rng *r, uint64_t seed);
uint64_t orc_rng_next(orc_rng *r);
int64_t  orc_rng_tuniform(orc_rng *r, uint32_t bound_log2);

/* ---- A.2/A.3 modulus switch ---- */
uint64_t orc_modulus_switch(uint64_t x, uint32_t log_modulus);
uint64_t orc_centered_ms_body_correction(const uint64_t *lwe, uint32_t n, uint32_t log_modulus);
/* out[0..n) = ms(mask), out[n] = ms(body + corr); ms_type 0 = standard, 1 = centered */
void     orc_lwe_modulus_switch(const uint64_t *lwe, uint32_t n, uint32_t log_modulus,
                                uint32_t ms_type, uint64_t *out);

/* ---- A.5 signed decomposer ---- */
uint64_t orc_closest_representable(uint64_t x, uint32_t base_log, uint32_t level);
uint64_t orc_decomp_init_state(uint64_t x, uint32_t base_log, uint32_t level);
uint64_t orc_decompose_one_level(uint32_t base_log, uint64_t *state);
/* digits[0] pairs with GGSW level-matrix 0 (= level `level`), digits[level-1] with level 1 */
void     orc_decompose(uint64_t x, uint32_t base_log, uint32_t level, int64_t *digits);

/* ---- A.4 monomial ops on one polynomial of N u64 ---- */
void orc_monomial_div(uint64_t *out, const uint64_t *in, uint32_t N, uint64_t degree);
void orc_monomial_mul(uint64_t *out, const uint64_t *in, uint32_t N, uint64_t degree);
void orc_monomial_mul_and_sub(uint64_t *out, const uint64_t *in, uint32_t N, uint64_t degree);

/* ---- A.8 sample extract / keyswitch / LUT ---- */
void orc_sample_extract(uint64_t *lwe_out, const uint64_t *glwe, uint32_t k, uint32_t N, uint32_t nth);
void orc_keyswitch(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *ksk,
                   uint32_t n_in, uint32_t n_out, uint32_t base_log, uint32_t level);
/* u64 ciphertext, u32 key -> u32 ciphertext (keyswitch_lwe_ciphertext_with_scalar_change: the KS32 pattern) */
void orc_keyswitch_64_32(uint32_t *lwe_out, const uint64_t *lwe_in, const uint32_t *ksk,
                         uint32_t n_in, uint32_t n_out, uint32_t base_log, uint32_t level);
void orc_generate_lut(uint64_t *glwe_out, uint32_t k, uint32_t N, uint32_t message_modulus,
                      uint64_t delta, const uint64_t *f_table);

/* ---- exact negacyclic product (Karatsuba, mod 2^64) ---- */
void orc_negacyclic_mul_add(uint64_t *out, const int64_t *small, const uint64_t *big, uint32_t N);
void orc_negacyclic_mul_add_naive(uint64_t *out, const int64_t *small, const uint64_t *big, uint32_t N);

/* ---- key generation / encryption (test inputs) ---- */
void     orc_gen_binary_key(orc_rng *r, uint64_t *sk, uint32_t len);
void     orc_lwe_encrypt(orc_rng *r, uint64_t *ct, const uint64_t *sk, uint32_t n,
                         uint64_t plaintext, uint32_t noise_bound_log2);
uint64_t orc_lwe_decrypt(const uint64_t *ct, const uint64_t *sk, uint32_t n);
void     orc_glwe_encrypt_assign(orc_rng *r, uint64_t *glwe /* body holds the plaintext */,
                                 const uint64_t *glwe_sk, uint32_t k, uint32_t N,
                                 uint32_t noise_bound_log2);
void     orc_gen_bsk(uint64_t seed, uint64_t *bsk, const uint64_t *lwe_sk, uint32_t n,
                     const uint64_t *glwe_sk, uint32_t k, uint32_t N,
                     uint32_t base_log, uint32_t level, uint32_t noise_bound_log2);
void     orc_gen_ksk(uint64_t seed, uint64_t *ksk, const uint64_t *sk_in, uint32_t n_in,
                     const uint64_t *sk_out, uint32_t n_out,
                     uint32_t base_log, uint32_t level, uint32_t noise_bound_log2);
void     orc_gen_multi_bit_bsk(uint64_t seed, uint64_t *bsk, const uint64_t *lwe_sk, uint32_t n,
                               const uint64_t *glwe_sk, uint32_t k, uint32_t N,
                               uint32_t base_log, uint32_t level, uint32_t grouping_factor,
                               uint32_t noise_bound_log2);

/* ---- PBS, exact integer semantics (karatsuba_pbs.rs) ---- */
void orc_pbs_exact(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                   const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N,
                   uint32_t base_log, uint32_t level, uint32_t ms_type);
/* blind rotation only, leaves the rotated accumulator in acc ((k+1)*N) */
void orc_blind_rotate_exact(uint64_t *acc, const uint64_t *msed /* n+1 */, const uint64_t *bsk_std,
                            uint32_t n, uint32_t k, uint32_t N, uint32_t base_log, uint32_t level);

/* ---- Goldilocks NTT (tfhe-ntt prime64 Solinas) ---- */
#define ORC_GOLDILOCKS_P 0xFFFFFFFF00000001ull
uint64_t orc_gl_add(uint64_t a, uint64_t b);
uint64_t orc_gl_sub(uint64_t a, uint64_t b);
uint64_t orc_gl_mul(uint64_t a, uint64_t b);
uint64_t orc_gl_pow(uint64_t a, uint64_t e);
uint64_t orc_gl_primitive_root_2N(uint32_t N);         /* psi: psi^N = -1 */
void     orc_ntt_forward(uint64_t *data, uint32_t N);  /* negacyclic, output bit-reversed */
void     orc_ntt_inverse(uint64_t *data, uint32_t N);  /* inverse of the above, UNnormalised */
void     orc_ntt_normalize(uint64_t *data, uint32_t N);
uint64_t orc_modswitch_pow2_to_prime(uint64_t x);      /* width 64 */
uint64_t orc_modswitch_prime_to_pow2(uint64_t v);      /* width 64 */
void     orc_convert_bsk_ntt(uint64_t *bsk_ntt, const uint64_t *bsk_std, uint32_t n, uint32_t k,
                             uint32_t N, uint32_t level);
void     orc_pbs_ntt_bnf(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                         const uint64_t *bsk_ntt, uint32_t n, uint32_t k, uint32_t N,
                         uint32_t base_log, uint32_t level, uint32_t ms_type);

/* ---- fixed-order f64 negacyclic FFT (DESIGN.md §4 "transform spec") ---- */
/* tables: fwd[2*idx], idx = (1<<d)+g ; inv[2*idx], idx = half+j ; untw[2*j] */
void orc_fft_tables(uint32_t N, double *fwd /* 2*(N/2) */, double *inv /* 2*(N/2) */,
                    double *untwist /* 2*(N/2) */);
void orc_fft_forward_int(double *out /* 2*(N/2) */, const int64_t *digits, uint32_t N);
void orc_fft_forward_torus(double *out, const uint64_t *poly, uint32_t N);
void orc_fft_backward_add(uint64_t *poly, double *fourier /* clobbered */, uint32_t N);
void orc_fft_forward_f64(double *out /* 2*(N/2) */, const double *in /* compressed polynomial */, uint32_t N);
void orc_fft_polynomial_mul_f64(double *out, const double *a, const double *b, uint32_t N);
int64_t orc_f64_to_i64_sat(double x);
uint64_t orc_from_torus(double t);
void orc_convert_bsk_fft(double *bsk_f, const uint64_t *bsk_std, uint32_t n, uint32_t k,
                         uint32_t N, uint32_t level);
void orc_pbs_fft(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                 const double *bsk_f, uint32_t n, uint32_t k, uint32_t N,
                 uint32_t base_log, uint32_t level, uint32_t ms_type);

/* ---- multi-bit PBS, deterministic semantics
 *      (lwe_multi_bit_programmable_bootstrapping.rs:647-880).  exact engine: keybundle combined in the
 *      integer domain (standard-domain key); f64 engine: combined in the Fourier domain like the CPU
 *      reference (:116-156), key converted once by orc_convert_multi_bit_bsk_fft ---- */
void orc_multi_bit_modulus_switch(const uint64_t *lwe, uint32_t n, uint32_t log_modulus,
                                  uint32_t grouping_factor, uint64_t *degrees /* (n/g)*(2^g) */,
                                  uint64_t *body_hat);
void orc_pbs_multi_bit_exact(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                             const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N,
                             uint32_t base_log, uint32_t level, uint32_t grouping_factor);
void orc_monomial_table(uint32_t N, double *z /* 2N complex: e^{i pi j / N} */);
void orc_monomial_fourier(uint32_t N, uint64_t degree, const double *z, double *m /* N/2 complex */);
void orc_convert_multi_bit_bsk_fft(double *bsk_f, const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N,
                                   uint32_t level, uint32_t grouping_factor);
void orc_pbs_multi_bit_fft(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                           const double *bsk_f, uint32_t n, uint32_t k, uint32_t N,
                           uint32_t base_log, uint32_t level, uint32_t grouping_factor);
void orc_pbs_multi_bit_fft_batch(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut,
                                 const double *bsk_f, uint32_t n, uint32_t k, uint32_t N, uint32_t base_log,
                                 uint32_t level, uint32_t grouping_factor, uint32_t count, uint32_t threads);

/* ---- the reference's own f64 transform in its golden-vector configuration (tfhe-fft radix-4 DIF plan,
 *      x86 conversion paths): tfhe_oracle_dif4.c ---- */
void orc_dif4_fft(double *buf /* N/2 complex, in place */, uint32_t N, int fwd);
void orc_dif4_convert_bsk(double *bsk_f, const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N, uint32_t level);
void orc_dif4_blind_rotate(uint64_t *acc, const uint64_t *lut, const uint64_t *msed, const double *bsk_f, uint32_t n,
                           uint32_t k, uint32_t N, uint32_t base_log, uint32_t level);

/* ---- batches (OpenMP over independent LWEs, like the reference's rayon bench,
 *      tfhe-benchmark/benches/core_crypto/pbs_bench.rs:176-196) ---- */
/* engine: 0 = exact, 1 = ntt_bnf, 2 = fft ; bsk in the engine's own format */
void orc_pbs_batch(uint32_t engine, uint64_t *lwe_out, const uint64_t *lwe_in,
                   const uint64_t *lut, const void *bsk, uint32_t n, uint32_t k, uint32_t N,
                   uint32_t base_log, uint32_t level, uint32_t ms_type, uint32_t count,
                   uint32_t threads);
void orc_keyswitch_batch(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *ksk,
                         uint32_t n_in, uint32_t n_out, uint32_t base_log, uint32_t level,
                         uint32_t count, uint32_t threads);
uint32_t orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
