// reference_gpu_tests.cpp — the reference's own GPU tests for the PBS hot path, restated in C++ on top of the compiled
// host mirror tfhe_rs_amd/host/core_crypto_gpu.hpp (which is written over the C ABI only).  The Rust originals cannot
// be compiled here (no cargo / rustc: SURVEY D5, row N3); each test below follows one of them step for step — same
// parameter sets, same number of encryptions per message, same index patterns, same determinism and decryption
// assertions — so a maintainer can read the two side by side:
//
//   lwe_encrypt_pbs_decrypt                        tfhe/src/core_crypto/gpu/algorithms/test/lwe_programmable_bootstrapping.rs:12-200
//   lwe_encrypt_centered_ms_pbs_decrypt            …/lwe_programmable_bootstrapping.rs:202-380
//   lwe_encrypt_multi_bit_pbs_decrypt_custom_mod   …/lwe_multi_bit_programmable_bootstrapping.rs:11-211
//   lwe_encrypt_ks_decrypt_custom_mod(_mb)         …/lwe_keyswitch.rs:14-312
//   lwe_encrypt_ks_decrypt_custom_mod_ks32         …/lwe_keyswitch.rs:314-523
//   test_(round_to_)closest_representable_gpu      …/lwe_keyswitch.rs:525-609
//   glwe_encrypt_sample_extract_decrypt_custom_mod …/glwe_sample_extraction.rs:14-149
//   compare_cpu_and_gpu_centered_modulus_switch    …/modulus_switch.rs:276-361 (with the four cooperative dimensions)
//   compare_cpu_and_gpu_cooperative_centered_modulus_switch_{throughput_pbs,generic}_block   …/modulus_switch.rs:380-488
//   cuda_fft_mult, forward_matches_classic_fft     backends/tfhe-cuda-backend/cuda/tests_and_benchmarks/tests/{test_fft,test_forward_fft16x4x16}.cpp
//   test_regression_fft16x4x16                     …/test/fft/mod.rs:268-294 (golden spectrum)
//   assert_gpu_determinism / should_check_determinism   …/test/mod.rs:34-84
//   mismatched_dimensions_panic                    the `assert_eq!`s of gpu/algorithms/*.rs (own test: the reference has none)
//
// Test infrastructure.  CPU-side key generation, encryption, decryption, LUT generation and the CPU modulus switch come
// from the oracle (oracle/tfhe_oracle.h) — the checker; the operator under test is reached through the host mirror only.
// Noise: the oracle's test generator has TUniform noise only, so Gaussian key noise of the reference's sets is restated
// as the TUniform bound of at least the same variance (2^b / sqrt 3 >= std * 2^64); the input ciphertexts get real Gaussian
// noise (Box–Muller on the oracle's generator).  Randomness is seeded (the reference's TestResources is not), so a failure
// reproduces.
//
//   usage: reference_gpu_tests <toy|reference> [test-name-substring]
//   "reference" = the reference's parameter sets (GPU tier); "toy" = small sets with the same code paths for the host
//   emulation tier (tests/emu), where one N = 2048 bootstrap takes seconds.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <vector>

#include "../../oracle/tfhe_oracle.h"
#include "../../tfhe_rs_amd/host/core_crypto_gpu.hpp"

using namespace tfhe::core_crypto::gpu;
using u64 = uint64_t;

#define CHECK(c)                                                                          \
  do {                                                                                    \
    if (!(c)) throw std::runtime_error(std::string("assertion failed: ") + #c + " at line " + std::to_string(__LINE__)); \
  } while (0)
#define CHECK_EQ(a, b)                                                                                                 \
  do {                                                                                                                 \
    auto va = (a);                                                                                                     \
    auto vb = (b);                                                                                                     \
    if (!(va == vb))                                                                                                   \
      throw std::runtime_error(std::string("assertion `left == right` failed: ") + #a + " = " + std::to_string(va) + ", " + #b + \
                               " = " + std::to_string(vb) + " at line " + std::to_string(__LINE__));                   \
  } while (0)

// ---- parameter sets: core_crypto/algorithms/test/mod.rs:57-80 (classic), :196-318 (multi-bit); noise as a standard deviation
// on the torus (Gaussian) or a TUniform bound (negative std = TUniform(-std))
struct ClassicTestParams {
  const char *name;
  size_t lwe_dimension, glwe_dimension, polynomial_size;
  double lwe_noise_std, glwe_noise_std;
  size_t pbs_base_log, pbs_level, ks_base_log, ks_level, message_modulus_log;
};
struct MultiBitTestParams {
  const char *name;
  size_t input_lwe_dimension;
  double lwe_noise_std;
  size_t decomp_base_log, decomp_level_count, glwe_dimension, polynomial_size;
  double glwe_noise_std;
  size_t message_modulus_log, grouping_factor;
};
static double tuniform(int b) { return -(double)b; }

static const ClassicTestParams TEST_PARAMS_4_BITS_NATIVE_U64 = {"test_params_4_bits_native_u64", 742, 1, 2048, 0.000007069849454709433,
                                                               0.00000000000000029403601535432533, 23, 1, 3, 5, 4};
static const MultiBitTestParams MULTI_BIT_2_2_2_PARAMS = {"multi_bit_2_2_2_params", 818, 0.000002226459789930014, 22, 1, 1, 2048,
                                                         0.0000000000000003152931493498455, 4, 2};
static const MultiBitTestParams MULTI_BIT_2_2_3_PARAMS = {"multi_bit_2_2_3_params", 888, 0.0000006125031601933181, 21, 1, 1, 2048,
                                                         0.0000000000000003152931493498455, 4, 3};
static const MultiBitTestParams MULTI_BIT_2_2_4_PARAMS = {"multi_bit_2_2_4_params", 920, tuniform(45), 22, 1, 1, 2048, tuniform(17), 4, 4};
static const MultiBitTestParams MULTI_BIT_3_3_2_PARAMS = {"multi_bit_3_3_2_params", 922, 0.0000003272369292345697, 14, 2, 1, 8192,
                                                         0.0000000000000000002168404344971009, 6, 2};
// small sets for the host-emulation tier: every code path of the reference sets (one level / two levels, k = 1 / 2, N = 2048
// throughput and latency kernels, g = 2, 3, 4)
static const ClassicTestParams TOY_4_BITS_N2048 = {"toy_4_bits_n2048", 10, 1, 2048, tuniform(40), tuniform(17), 23, 1, 4, 4, 4};
static const ClassicTestParams TOY_2_BITS_K2_N256 = {"toy_2_bits_k2_n256", 20, 2, 256, tuniform(40), tuniform(20), 12, 3, 3, 6, 2};
static const MultiBitTestParams TOY_MB_2 = {"toy_multi_bit_g2", 16, tuniform(40), 15, 2, 1, 512, tuniform(20), 2, 2};
static const MultiBitTestParams TOY_MB_3 = {"toy_multi_bit_g3", 18, tuniform(40), 15, 2, 1, 256, tuniform(20), 2, 3};
static const MultiBitTestParams TOY_MB_4 = {"toy_multi_bit_g4_n2048", 8, tuniform(45), 22, 1, 1, 2048, tuniform(17), 4, 4};

static bool g_toy = false;
static size_t nb_tests() { return g_toy ? 1 : 10; }  // const NB_TESTS: usize = 10 in every reference test

// ---- TestResources (core_crypto/algorithms/test/mod.rs:560-600): one seeded generator for secrets, one for encryption
struct TestResources {
  orc_rng secret_random_generator, encryption_random_generator;
  uint64_t key_seed;
  explicit TestResources(uint64_t seed) : key_seed(seed * 1000003u + 17) {
    orc_rng_seed(&secret_random_generator, seed);
    orc_rng_seed(&encryption_random_generator, seed ^ 0x9e3779b97f4a7c15ull);
  }
};

static uint32_t key_noise_bound(double std) {  // TUniform bound with at least the Gaussian's variance
  if (std <= 0) return std < 0 ? (uint32_t)(-std) : 0;
  return (uint32_t)std::ceil(std::log2(std * 18446744073709551616.0 * std::sqrt(3.0)));
}
static int64_t sample_noise(orc_rng *r, double std) {
  if (std < 0) return orc_rng_tuniform(r, (uint32_t)(-std));
  if (std == 0) return 0;
  const double u1 = ((orc_rng_next(r) >> 11) + 1) * (1.0 / 9007199254740993.0), u2 = (orc_rng_next(r) >> 11) * (1.0 / 9007199254740992.0);
  return (int64_t)std::llround(std * 18446744073709551616.0 * std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
}
// allocate_and_encrypt_new_lwe_ciphertext (algorithms/lwe_encryption.rs)
static std::vector<u64> encrypt_lwe(TestResources &rsc, const std::vector<u64> &sk, u64 plaintext, double noise_std) {
  std::vector<u64> ct(sk.size() + 1);
  orc_rng *r = &rsc.encryption_random_generator;
  u64 b = plaintext + (u64)sample_noise(r, noise_std);
  for (size_t i = 0; i < sk.size(); ++i) {
    ct[i] = orc_rng_next(r);
    if (sk[i]) b += ct[i];
  }
  ct[sk.size()] = b;
  return ct;
}
static std::vector<u64> binary_key(TestResources &rsc, size_t len) {
  std::vector<u64> sk(len);
  orc_gen_binary_key(&rsc.secret_random_generator, sk.data(), (uint32_t)len);
  return sk;
}
static u64 round_decode(u64 decrypted, u64 delta) {  // algorithms/test/mod.rs:489-491 -> divide_round (algorithms/misc.rs:6-18)
  return decrypted / delta + (decrypted % delta >= (delta >> 1));
}
static std::vector<u64> generate_lut(size_t N, size_t k, u64 msg_modulus, u64 delta, const std::function<u64(u64)> &f) {
  std::vector<u64> table(msg_modulus), acc((k + 1) * N);
  for (u64 m = 0; m < msg_modulus; ++m) table[m] = f(m);
  orc_generate_lut(acc.data(), (uint32_t)k, (uint32_t)N, (uint32_t)msg_modulus, delta, table.data());
  return acc;
}

// test/mod.rs:34-73
static void assert_gpu_determinism(const std::vector<u64> &first_run, const std::vector<u64> &second_run, const char *op) {
  if (first_run.size() != second_run.size())
    throw std::runtime_error(std::string("Failed determinism check for ") + op + ": output lengths differ");
  size_t diff = 0, first = 0;
  for (size_t i = 0; i < first_run.size(); ++i)
    if (first_run[i] != second_run[i] && !diff++) first = i;
  if (diff)
    throw std::runtime_error(std::string("Failed determinism check for ") + op + ": running it twice on the same input gave different outputs, " +
                             std::to_string(diff) + "/" + std::to_string(first_run.size()) + " coefficients differ, first at index " +
                             std::to_string(first));
}
// test/mod.rs:75-84: on the emulation tier every set runs it (the toy sets have fewer message bits)
static bool should_check_determinism(size_t message_modulus_log) { return g_toy || message_modulus_log == 4; }

static CudaVec<u64> device_indexes(const std::vector<u64> &h, const CudaStreams &stream) {
  CudaVec<u64> d = CudaVec<u64>::new_async(h.size(), stream, 0);
  d.copy_from_cpu_async(h, stream, 0);
  stream.synchronize();
  return d;
}

// ---------------------------------------------------------------------------------------------------------------------
// lwe_programmable_bootstrapping.rs:12-200 and :202-380 (centered = the second test: "We force the centered modulus switch")
static void lwe_encrypt_pbs_decrypt_impl(const ClassicTestParams &params, bool centered) {
  const size_t input_lwe_dimension = params.lwe_dimension, glwe_dimension = params.glwe_dimension, polynomial_size = params.polynomial_size;
  const u64 msg_modulus = u64(1) << params.message_modulus_log;
  const CiphertextModulus ciphertext_modulus = CiphertextModulus::new_native();
  const u64 delta = (u64(1) << 63) / msg_modulus;  // encoding_with_padding / msg_modulus
  CudaStreams stream = CudaStreams::new_single_gpu(GpuIndex(0));
  TestResources rsc(centered ? 11 : 7);
  auto f = [&](u64 x) { return (x * 2 - 1) % msg_modulus; };  // wrapping_mul(2).wrapping_sub(1).wrapping_rem(msg_modulus)
  u64 msg = msg_modulus;
  const size_t number_of_messages = 1;

  const std::vector<u64> accumulator = generate_lut(polynomial_size, glwe_dimension, msg_modulus, delta, f);
  const std::vector<u64> input_lwe_secret_key = binary_key(rsc, input_lwe_dimension);
  const std::vector<u64> output_glwe_secret_key = binary_key(rsc, glwe_dimension * polynomial_size);
  const std::vector<u64> &output_lwe_secret_key = output_glwe_secret_key;  // into_lwe_secret_key
  const size_t output_lwe_dimension = output_lwe_secret_key.size();

  std::vector<u64> bsk(input_lwe_dimension * (glwe_dimension + 1) * (glwe_dimension + 1) * params.pbs_level * polynomial_size);
  orc_gen_bsk(rsc.key_seed, bsk.data(), input_lwe_secret_key.data(), (uint32_t)input_lwe_dimension, output_glwe_secret_key.data(),
              (uint32_t)glwe_dimension, (uint32_t)polynomial_size, (uint32_t)params.pbs_base_log, (uint32_t)params.pbs_level,
              key_noise_bound(params.glwe_noise_std));
  const CudaLweBootstrapKey d_bsk = CudaLweBootstrapKey::from_lwe_bootstrap_key(
      bsk, input_lwe_dimension, glwe_dimension, polynomial_size, params.pbs_base_log, params.pbs_level, centered, stream);

  while (msg != 0) {
    msg -= 1;
    for (size_t t = 0; t < nb_tests(); ++t) {
      const u64 plaintext = msg * delta;
      const std::vector<u64> lwe_ciphertext_in = encrypt_lwe(rsc, input_lwe_secret_key, plaintext, params.lwe_noise_std);
      const auto d_lwe_ciphertext_in = CudaLweCiphertextList<u64>::from_lwe_ciphertext(lwe_ciphertext_in, ciphertext_modulus, stream);
      CudaLweCiphertextList<u64> d_out_pbs_ct(output_lwe_dimension, 1, ciphertext_modulus, stream);
      const auto d_accumulator =
          CudaGlweCiphertextList<u64>::from_glwe_ciphertext(accumulator, glwe_dimension, polynomial_size, ciphertext_modulus, stream);
      std::vector<u64> test_vector_indexes(number_of_messages);
      std::iota(test_vector_indexes.begin(), test_vector_indexes.end(), 0);
      const CudaVec<u64> d_test_vector_indexes = device_indexes(test_vector_indexes, stream);
      const size_t num_blocks = d_lwe_ciphertext_in.lwe_ciphertext_count();
      std::vector<u64> lwe_indexes(num_blocks);
      std::iota(lwe_indexes.begin(), lwe_indexes.end(), 0);
      const CudaVec<u64> d_output_indexes = device_indexes(lwe_indexes, stream), d_input_indexes = device_indexes(lwe_indexes, stream);

      cuda_programmable_bootstrap_lwe_ciphertext(d_lwe_ciphertext_in, d_out_pbs_ct, d_accumulator, d_test_vector_indexes, d_output_indexes,
                                                 d_input_indexes, d_bsk, stream);
      const std::vector<u64> out_pbs_ct = d_out_pbs_ct.into_lwe_ciphertext(stream);

      if (should_check_determinism(params.message_modulus_log)) {
        CudaLweCiphertextList<u64> d_out_pbs_ct_bis(output_lwe_dimension, 1, ciphertext_modulus, stream);
        cuda_programmable_bootstrap_lwe_ciphertext(d_lwe_ciphertext_in, d_out_pbs_ct_bis, d_accumulator, d_test_vector_indexes,
                                                   d_output_indexes, d_input_indexes, d_bsk, stream);
        assert_gpu_determinism(out_pbs_ct, d_out_pbs_ct_bis.into_lwe_ciphertext(stream),
                               centered ? "cuda_programmable_bootstrap_lwe_ciphertext (centered modulus switch)"
                                        : "cuda_programmable_bootstrap_lwe_ciphertext");
      }
      const u64 decrypted = orc_lwe_decrypt(out_pbs_ct.data(), output_lwe_secret_key.data(), (uint32_t)output_lwe_dimension);
      const u64 decoded = round_decode(decrypted, delta) % msg_modulus;
      CHECK_EQ(decoded, f(msg));
    }
  }
}

// lwe_multi_bit_programmable_bootstrapping.rs:11-211
static void lwe_encrypt_multi_bit_pbs_decrypt_custom_mod(const MultiBitTestParams &params) {
  const size_t input_lwe_dimension = params.input_lwe_dimension, glwe_dimension = params.glwe_dimension,
               polynomial_size = params.polynomial_size, grouping_factor = params.grouping_factor;
  const u64 msg_modulus = u64(1) << params.message_modulus_log;
  const CiphertextModulus ciphertext_modulus = CiphertextModulus::new_native();
  const u64 delta = (u64(1) << 63) / msg_modulus;
  CudaStreams stream = CudaStreams::new_single_gpu(GpuIndex(0));
  TestResources rsc(13 + grouping_factor);
  auto f = [&](u64 x) { return (x * 2 - 1) % msg_modulus; };
  u64 msg = msg_modulus;

  const std::vector<u64> accumulator = generate_lut(polynomial_size, glwe_dimension, msg_modulus, delta, f);
  const std::vector<u64> input_lwe_secret_key = binary_key(rsc, input_lwe_dimension);
  const std::vector<u64> output_glwe_secret_key = binary_key(rsc, glwe_dimension * polynomial_size);
  const size_t output_lwe_dimension = output_glwe_secret_key.size();

  std::vector<u64> bsk((input_lwe_dimension / grouping_factor) * (size_t(1) << grouping_factor) * (glwe_dimension + 1) * (glwe_dimension + 1) *
                       params.decomp_level_count * polynomial_size);
  orc_gen_multi_bit_bsk(rsc.key_seed, bsk.data(), input_lwe_secret_key.data(), (uint32_t)input_lwe_dimension, output_glwe_secret_key.data(),
                        (uint32_t)glwe_dimension, (uint32_t)polynomial_size, (uint32_t)params.decomp_base_log,
                        (uint32_t)params.decomp_level_count, (uint32_t)grouping_factor, key_noise_bound(params.glwe_noise_std));
  const CudaLweMultiBitBootstrapKey d_bsk = CudaLweMultiBitBootstrapKey::from_lwe_multi_bit_bootstrap_key(
      bsk, input_lwe_dimension, glwe_dimension, polynomial_size, params.decomp_base_log, params.decomp_level_count, grouping_factor, stream);

  while (msg != 0) {
    msg -= 1;
    for (size_t t = 0; t < nb_tests(); ++t) {
      const std::vector<u64> lwe_ciphertext_in = encrypt_lwe(rsc, input_lwe_secret_key, msg * delta, params.lwe_noise_std);
      const auto d_lwe_ciphertext_in = CudaLweCiphertextList<u64>::from_lwe_ciphertext(lwe_ciphertext_in, ciphertext_modulus, stream);
      CudaLweCiphertextList<u64> d_out_pbs_ct(output_lwe_dimension, 1, ciphertext_modulus, stream);
      const auto d_accumulator =
          CudaGlweCiphertextList<u64>::from_glwe_ciphertext(accumulator, glwe_dimension, polynomial_size, ciphertext_modulus, stream);
      const CudaVec<u64> d_test_vector_indexes = device_indexes({0}, stream), d_output_indexes = device_indexes({0}, stream),
                         d_input_indexes = device_indexes({0}, stream);
      cuda_multi_bit_programmable_bootstrap_lwe_ciphertext(d_lwe_ciphertext_in, d_out_pbs_ct, d_accumulator, d_test_vector_indexes,
                                                           d_output_indexes, d_input_indexes, d_bsk, stream);
      const std::vector<u64> out_pbs_ct = d_out_pbs_ct.into_lwe_ciphertext(stream);
      if (should_check_determinism(params.message_modulus_log)) {
        CudaLweCiphertextList<u64> d_out_pbs_ct_bis(output_lwe_dimension, 1, ciphertext_modulus, stream);
        cuda_multi_bit_programmable_bootstrap_lwe_ciphertext(d_lwe_ciphertext_in, d_out_pbs_ct_bis, d_accumulator, d_test_vector_indexes,
                                                             d_output_indexes, d_input_indexes, d_bsk, stream);
        assert_gpu_determinism(out_pbs_ct, d_out_pbs_ct_bis.into_lwe_ciphertext(stream), "cuda_multi_bit_programmable_bootstrap_lwe_ciphertext");
      }
      const u64 decrypted = orc_lwe_decrypt(out_pbs_ct.data(), output_glwe_secret_key.data(), (uint32_t)output_lwe_dimension);
      CHECK_EQ(round_decode(decrypted, delta) % msg_modulus, f(msg));
    }
  }
}

// integer/gpu/server_key/radix/tests_noise_distribution/utils/noise_simulation.rs:1186-1362 `multi_bit_mod_switch` followed by
// `apply_generic_blind_rotation` (what the reference's noise tests do around the multi-bit blind rotation): the output buffer
// takes the input ciphertext first, cuda_modulus_switch_multi_bit_64_async writes behind it, and
// programmable_bootstrap_multi_bit_noise_tests bootstraps from that buffer.  Checked here: the result IS the standard multi-bit
// bootstrap of the same ciphertext (same bits) and decrypts to f(msg).  N = 2048 only, as in the reference.
static void multi_bit_mod_switch_then_blind_rotation(const MultiBitTestParams &params) {
  const size_t input_lwe_dimension = params.input_lwe_dimension, glwe_dimension = params.glwe_dimension,
               polynomial_size = params.polynomial_size, grouping_factor = params.grouping_factor;
  const u64 msg_modulus = u64(1) << params.message_modulus_log;
  const CiphertextModulus ciphertext_modulus = CiphertextModulus::new_native();
  const u64 delta = (u64(1) << 63) / msg_modulus;
  CudaStreams stream = CudaStreams::new_single_gpu(GpuIndex(0));
  TestResources rsc(29 + grouping_factor);
  auto f = [&](u64 x) { return (x * 3 + 1) % msg_modulus; };
  const std::vector<u64> accumulator = generate_lut(polynomial_size, glwe_dimension, msg_modulus, delta, f);
  const std::vector<u64> input_lwe_secret_key = binary_key(rsc, input_lwe_dimension);
  const std::vector<u64> output_glwe_secret_key = binary_key(rsc, glwe_dimension * polynomial_size);
  const size_t output_lwe_dimension = output_glwe_secret_key.size(), lwe_size = input_lwe_dimension + 1;
  std::vector<u64> bsk((input_lwe_dimension / grouping_factor) * (size_t(1) << grouping_factor) * (glwe_dimension + 1) * (glwe_dimension + 1) *
                       params.decomp_level_count * polynomial_size);
  orc_gen_multi_bit_bsk(rsc.key_seed, bsk.data(), input_lwe_secret_key.data(), (uint32_t)input_lwe_dimension, output_glwe_secret_key.data(),
                        (uint32_t)glwe_dimension, (uint32_t)polynomial_size, (uint32_t)params.decomp_base_log,
                        (uint32_t)params.decomp_level_count, (uint32_t)grouping_factor, key_noise_bound(params.glwe_noise_std));
  const CudaLweMultiBitBootstrapKey d_bsk = CudaLweMultiBitBootstrapKey::from_lwe_multi_bit_bootstrap_key(
      bsk, input_lwe_dimension, glwe_dimension, polynomial_size, params.decomp_base_log, params.decomp_level_count, grouping_factor, stream);
  const auto d_accumulator =
      CudaGlweCiphertextList<u64>::from_glwe_ciphertext(accumulator, glwe_dimension, polynomial_size, ciphertext_modulus, stream);
  const CudaVec<u64> zero_index = device_indexes({0}, stream);
  for (u64 msg = 0; msg < msg_modulus; msg += (g_toy ? 1 : 5)) {
    const std::vector<u64> lwe_ciphertext_in = encrypt_lwe(rsc, input_lwe_secret_key, msg * delta, params.lwe_noise_std);
    const auto d_input = CudaLweCiphertextList<u64>::from_lwe_ciphertext(lwe_ciphertext_in, ciphertext_modulus, stream);
    // allocate_multi_bit_mod_switch_result: (1 << grouping_factor) + 1 ciphertexts of the input's size
    CudaLweCiphertextList<u64> d_switched(input_lwe_dimension, (size_t(1) << grouping_factor) + 1, ciphertext_modulus, stream);
    // multi_bit_mod_switch: the input into the first slot, the switch behind it
    d_switched.d_vec.copy_from_cpu_async(lwe_ciphertext_in, stream, 0);
    stream.synchronize();
    CudaVec<u64> input_copy = CudaVec<u64>::from_cpu_async(lwe_ciphertext_in, stream, 0);
    cuda_modulus_switch_multi_bit_ciphertext(stream, d_switched.d_vec, input_copy, 12, (uint32_t)polynomial_size, (uint32_t)grouping_factor,
                                             lwe_size);
    // the switch's words are what the CPU switch gives (lwe_multi_bit_programmable_bootstrapping.rs:30-65)
    {
      std::vector<u64> deg((input_lwe_dimension / grouping_factor) << grouping_factor);
      const std::vector<u64> host = d_switched.d_vec.to_cpu(stream, 0);
      u64 body = 0;
      orc_multi_bit_modulus_switch(lwe_ciphertext_in.data(), (uint32_t)input_lwe_dimension, 12, (uint32_t)grouping_factor, deg.data(), &body);
      for (size_t i = 0; i < deg.size(); ++i) CHECK_EQ(host[lwe_size + i], deg[i]);
    }
    // apply_generic_blind_rotation
    CudaLweCiphertextList<u64> d_out(output_lwe_dimension, 1, ciphertext_modulus, stream), d_std(output_lwe_dimension, 1, ciphertext_modulus, stream);
    programmable_bootstrap_multi_bit_noise_tests(stream, d_out.d_vec, zero_index, d_accumulator.d_vec, zero_index, d_switched.d_vec, zero_index,
                                                 d_bsk.d_vec, input_lwe_dimension, glwe_dimension, polynomial_size, params.decomp_base_log,
                                                 params.decomp_level_count, grouping_factor, 1);
    cuda_multi_bit_programmable_bootstrap_lwe_ciphertext(d_input, d_std, d_accumulator, zero_index, zero_index, zero_index, d_bsk, stream);
    const std::vector<u64> out = d_out.to_lwe_ciphertext_list(stream), std_out = d_std.to_lwe_ciphertext_list(stream);
    assert_gpu_determinism(std_out, out, "programmable_bootstrap_multi_bit_noise_tests vs the standard multi-bit bootstrap");
    const u64 decrypted = orc_lwe_decrypt(out.data(), output_glwe_secret_key.data(), (uint32_t)output_lwe_dimension);
    CHECK_EQ(round_decode(decrypted, delta) % msg_modulus, f(msg));
  }
}

// lwe_keyswitch.rs:72-312 `base_lwe_encrypt_ks_decrypt_custom_mod`: GEMM and classic keyswitch decrypt correctly, are
// bit-wise equal, and only a subset of the LWEs can be keyswitched (the others stay zero)
static void base_lwe_encrypt_ks_decrypt_custom_mod(size_t lwe_dimension, double lwe_noise_std, size_t message_modulus_log, size_t glwe_dimension,
                                                   size_t polynomial_size, size_t ks_decomp_base_log, size_t ks_decomp_level_count) {
  CudaStreams stream = CudaStreams::new_single_gpu(GpuIndex(0));
  TestResources rsc(23);
  const CiphertextModulus ciphertext_modulus = CiphertextModulus::new_native();
  const u64 msg_modulus = u64(1) << message_modulus_log;
  u64 msg = msg_modulus;
  const u64 delta = (u64(1) << 63) / msg_modulus;

  const std::vector<u64> lwe_sk = binary_key(rsc, lwe_dimension);
  const std::vector<u64> big_lwe_sk = binary_key(rsc, glwe_dimension * polynomial_size);  // glwe_sk.into_lwe_secret_key()
  std::vector<u64> ksk_big_to_small(big_lwe_sk.size() * ks_decomp_level_count * (lwe_dimension + 1));
  orc_gen_ksk(rsc.key_seed, ksk_big_to_small.data(), big_lwe_sk.data(), (uint32_t)big_lwe_sk.size(), lwe_sk.data(), (uint32_t)lwe_dimension,
              (uint32_t)ks_decomp_base_log, (uint32_t)ks_decomp_level_count, key_noise_bound(lwe_noise_std));
  const auto d_ksk_big_to_small = CudaLweKeyswitchKey<u64>::from_lwe_keyswitch_key(ksk_big_to_small, big_lwe_sk.size(), lwe_dimension,
                                                                                   ks_decomp_base_log, ks_decomp_level_count, stream);
  // the reference walks every message with the same block pattern; the emulation tier takes the first two
  size_t msgs_left = g_toy ? 2 : msg_modulus;
  while (msg != 0 && msgs_left--) {
    msg -= 1;
    for (size_t test_idx = 0; test_idx < (g_toy ? 4 : 10); ++test_idx) {
      const size_t num_blocks = test_idx * test_idx * 3 + 1;
      std::vector<u64> plaintext_list(num_blocks);
      for (size_t i = 0; i < num_blocks; ++i) plaintext_list[i] = (i % msg_modulus) * delta;
      std::vector<u64> input_ct_list;
      input_ct_list.reserve(num_blocks * (big_lwe_sk.size() + 1));
      for (size_t i = 0; i < num_blocks; ++i) {
        const std::vector<u64> ct = encrypt_lwe(rsc, big_lwe_sk, plaintext_list[i], lwe_noise_std);
        input_ct_list.insert(input_ct_list.end(), ct.begin(), ct.end());
      }
      const auto input_ct_list_gpu = CudaLweCiphertextList<u64>::from_lwe_ciphertext_list(input_ct_list, big_lwe_sk.size(), ciphertext_modulus, stream);
      const std::vector<u64> output_ct_list(num_blocks * (lwe_dimension + 1), 0);
      auto zero_out = [&] { return CudaLweCiphertextList<u64>::from_lwe_ciphertext_list(output_ct_list, lwe_dimension, ciphertext_modulus, stream); };
      CudaLweCiphertextList<u64> output_ct_list_gpu = zero_out(), output_ct_list_gpu_gemm = zero_out();

      const bool use_trivial_indexes = test_idx % 2 == 0;
      const size_t num_blocks_to_ks = use_trivial_indexes ? (test_idx % 4 == 0 ? num_blocks : num_blocks / 2) : num_blocks;
      std::vector<size_t> lwe_indexes(num_blocks), lwe_indexes_out;
      std::iota(lwe_indexes.begin(), lwe_indexes.end(), 0);
      lwe_indexes_out = lwe_indexes;
      if (!use_trivial_indexes) {  // shuffle(&mut thread_rng()): Fisher–Yates on the seeded generator
        for (auto *v : {&lwe_indexes, &lwe_indexes_out})
          for (size_t i = v->size(); i > 1; --i) std::swap((*v)[i - 1], (*v)[orc_rng_next(&rsc.encryption_random_generator) % i]);
      }
      std::vector<u64> h_lwe_indexes(lwe_indexes.begin(), lwe_indexes.begin() + num_blocks_to_ks),
          h_lwe_indexes_out(lwe_indexes_out.begin(), lwe_indexes_out.begin() + num_blocks_to_ks);
      const CudaVec<u64> d_input_indexes = device_indexes(h_lwe_indexes, stream), d_output_indexes = device_indexes(h_lwe_indexes_out, stream);

      cuda_keyswitch_lwe_ciphertext(d_ksk_big_to_small, input_ct_list_gpu, output_ct_list_gpu, d_input_indexes, d_output_indexes,
                                    use_trivial_indexes, stream, false);
      cuda_keyswitch_lwe_ciphertext(d_ksk_big_to_small, input_ct_list_gpu, output_ct_list_gpu_gemm, d_input_indexes, d_output_indexes,
                                    use_trivial_indexes, stream, true);
      for (bool use_gemm : {false, true}) {  // determinism, on both the classical and the GEMM variant
        CudaLweCiphertextList<u64> output_ct_list_gpu_bis = zero_out();
        cuda_keyswitch_lwe_ciphertext(d_ksk_big_to_small, input_ct_list_gpu, output_ct_list_gpu_bis, d_input_indexes, d_output_indexes,
                                      use_trivial_indexes, stream, use_gemm);
        const std::vector<u64> first = (use_gemm ? output_ct_list_gpu_gemm : output_ct_list_gpu).to_lwe_ciphertext_list(stream),
                               second = output_ct_list_gpu_bis.to_lwe_ciphertext_list(stream);
        if (first != second) {  // which of the two runs left the checker's keyswitch, and where (sample positions s = tile * 32 + row)
          for (const std::vector<u64> *run : {&first, &second}) {
            std::string where;
            size_t bad = 0;
            for (size_t i = 0; i < num_blocks_to_ks; ++i) {
              std::vector<u64> want(lwe_dimension + 1);
              orc_keyswitch(want.data(), &input_ct_list[lwe_indexes[i] * (big_lwe_sk.size() + 1)], ksk_big_to_small.data(),
                            (uint32_t)big_lwe_sk.size(), (uint32_t)lwe_dimension, (uint32_t)ks_decomp_base_log, (uint32_t)ks_decomp_level_count);
              if (!std::equal(want.begin(), want.end(), &(*run)[lwe_indexes_out[i] * (lwe_dimension + 1)])) {
                if (bad++ < 12) where += " " + std::to_string(i);
              }
            }
            std::printf("  keyswitch of %zu of %zu blocks, %s run: %zu samples differ from the oracle, s =%s\n", num_blocks_to_ks, num_blocks,
                        run == &first ? "first" : "second", bad, where.c_str());
          }
        }
        assert_gpu_determinism(first, second, use_gemm ? "cuda_keyswitch_lwe_ciphertext (GEMM)" : "cuda_keyswitch_lwe_ciphertext");
      }
      // only the LWEs at the output indices are set; the test checks that the others remain 0
      std::vector<u64> ref_vec(num_blocks, 0);
      for (size_t i = 0; i < num_blocks_to_ks; ++i) ref_vec[lwe_indexes_out[i]] = round_decode(plaintext_list[lwe_indexes[i]], delta);
      CHECK_EQ(output_ct_list_gpu.lwe_ciphertext_count(), num_blocks);
      const std::vector<u64> cpu = output_ct_list_gpu.to_lwe_ciphertext_list(stream), cpu_gemm = output_ct_list_gpu_gemm.to_lwe_ciphertext_list(stream);
      for (size_t i = 0; i < num_blocks; ++i) {
        const u64 *a = &cpu[i * (lwe_dimension + 1)], *b = &cpu_gemm[i * (lwe_dimension + 1)];
        CHECK(std::equal(a, a + lwe_dimension + 1, b));  // GEMM vs classical: bit-wise equal
        CHECK_EQ(round_decode(orc_lwe_decrypt(a, lwe_sk.data(), (uint32_t)lwe_dimension), delta) % msg_modulus, ref_vec[i]);
        CHECK_EQ(round_decode(orc_lwe_decrypt(b, lwe_sk.data(), (uint32_t)lwe_dimension), delta) % msg_modulus, ref_vec[i]);
      }
    }
  }
}
static void lwe_encrypt_ks_decrypt_custom_mod(const ClassicTestParams &p) {  // lwe_keyswitch.rs:14-38
  base_lwe_encrypt_ks_decrypt_custom_mod(p.lwe_dimension, p.lwe_noise_std, p.message_modulus_log, p.glwe_dimension, p.polynomial_size,
                                         p.ks_base_log, p.ks_level);
}
static void lwe_encrypt_ks_decrypt_custom_mod_mb(const MultiBitTestParams &p) {  // lwe_keyswitch.rs:40-64: zero noise, the PBS decomposition
  base_lwe_encrypt_ks_decrypt_custom_mod(p.input_lwe_dimension, 0.0, p.message_modulus_log, p.glwe_dimension, p.polynomial_size, p.decomp_base_log,
                                         p.decomp_level_count);
}

// lwe_keyswitch.rs:314-523 `lwe_encrypt_ks_decrypt_ks32_common` on MULTI_BIT_2_2_2_KS32_PARAMS (test/mod.rs:92-108): u64
// ciphertexts under the big key, a u32 keyswitch key, u32 outputs; classic and GEMM keyswitch equal the CPU's
// keyswitch_lwe_ciphertext_with_scalar_change word for word and decrypt under the small key on 32 bits
static void lwe_encrypt_ks_decrypt_custom_mod_ks32() {
  const size_t lwe_dimension = g_toy ? 24 : 920, glwe_dimension = 1, polynomial_size = g_toy ? 256 : 2048, ks_decomp_base_log = 3,
               ks_decomp_level_count = 5, message_modulus_log = 2;
  const uint32_t noise_bound_log2 = 13;
  const u64 input_msg_modulus = u64(1) << message_modulus_log, input_delta = (u64(1) << 63) / input_msg_modulus;
  const uint32_t output_delta = (uint32_t(1) << 31) >> message_modulus_log;
  CudaStreams stream = CudaStreams::new_single_gpu(GpuIndex(0));
  TestResources rsc(37);
  const std::vector<u64> lwe_sk = binary_key(rsc, lwe_dimension), big_lwe_sk = binary_key(rsc, glwe_dimension * polynomial_size);
  // allocate_and_generate_new_lwe_keyswitch_key on u32: block i, level l first, encrypts s_i * 2^(32 - base_log * level)
  std::vector<uint32_t> ksk(big_lwe_sk.size() * ks_decomp_level_count * (lwe_dimension + 1));
  orc_rng *r = &rsc.encryption_random_generator;
  for (size_t i = 0, row = 0; i < big_lwe_sk.size(); ++i)
    for (size_t lvl = ks_decomp_level_count; lvl >= 1; --lvl, ++row) {
      uint32_t *ct = &ksk[row * (lwe_dimension + 1)];
      uint32_t b = (uint32_t)(big_lwe_sk[i] << (32 - ks_decomp_base_log * lvl)) + (uint32_t)orc_rng_tuniform(r, noise_bound_log2);
      for (size_t j = 0; j < lwe_dimension; ++j) {
        ct[j] = (uint32_t)orc_rng_next(r);
        if (lwe_sk[j]) b += ct[j];
      }
      ct[lwe_dimension] = b;
    }
  const auto d_ksk = CudaLweKeyswitchKey<uint32_t>::from_lwe_keyswitch_key(ksk, big_lwe_sk.size(), lwe_dimension, ks_decomp_base_log,
                                                                             ks_decomp_level_count, stream);
  for (u64 msg = input_msg_modulus; msg-- != 0;)
    for (size_t t = 0; t < nb_tests(); ++t) {
      const std::vector<u64> ct = encrypt_lwe(rsc, big_lwe_sk, msg * input_delta, tuniform((int)noise_bound_log2));
      std::vector<uint32_t> output_ct_ref(lwe_dimension + 1);
      orc_keyswitch_64_32(output_ct_ref.data(), ct.data(), ksk.data(), (uint32_t)big_lwe_sk.size(), (uint32_t)lwe_dimension,
                          (uint32_t)ks_decomp_base_log, (uint32_t)ks_decomp_level_count);
      auto decode32 = [&](const std::vector<uint32_t> &c) {
        uint32_t ph = c[lwe_dimension];
        for (size_t j = 0; j < lwe_dimension; ++j)
          if (lwe_sk[j]) ph -= c[j];
        return (u64)((ph / output_delta + (ph % output_delta >= (output_delta >> 1))) % (uint32_t)input_msg_modulus);
      };
      CHECK_EQ(decode32(output_ct_ref), msg);
      const auto d_ct = CudaLweCiphertextList<u64>::from_lwe_ciphertext(ct, CiphertextModulus::new_native(), stream);
      CudaLweCiphertextList<uint32_t> d_output_ct(lwe_dimension, 1, CiphertextModulus{32}, stream), d_output_ct_gemm(lwe_dimension, 1, CiphertextModulus{32}, stream);
      const CudaVec<u64> d_input_indexes = device_indexes({0}, stream), d_output_indexes = device_indexes({0}, stream);
      cuda_keyswitch_lwe_ciphertext(d_ksk, d_ct, d_output_ct, d_input_indexes, d_output_indexes, true, stream, false);
      cuda_keyswitch_lwe_ciphertext(d_ksk, d_ct, d_output_ct_gemm, d_input_indexes, d_output_indexes, true, stream, true);
      const std::vector<uint32_t> output_ct = d_output_ct.into_lwe_ciphertext(stream), output_ct_gemm = d_output_ct_gemm.into_lwe_ciphertext(stream);
      CHECK(output_ct == output_ct_ref);
      CHECK(output_ct_gemm == output_ct_ref);
      CHECK_EQ(decode32(output_ct), msg);
    }
}

// lwe_keyswitch.rs:525-556.  The reference's helper makes a stream and two one-word vectors per call; here they are made
// once per test (26,000 calls of the helper took 78 s on the MI355X — 3 ms of stream and allocation set-up each,
// profiles/r04h_new_gpu_tests3_timeout.log — against 1 s with the set-up hoisted)
struct ClosestRepresentableOnGpu {
  CudaStreams stream = CudaStreams::new_single_gpu(GpuIndex(0));
  CudaVec<u64> d_input = CudaVec<u64>::new_async(1, stream, 0), d_output = CudaVec<u64>::new_async(1, stream, 0);
  u64 operator()(u64 value, uint32_t base_log, uint32_t level_count) {
    const std::vector<u64> h_input{value};
    d_input.copy_from_cpu_async(h_input, stream, 0);
    cuda_closest_representable(stream, d_input, d_output, base_log, level_count);
    std::vector<u64> h_output{0};
    d_output.copy_to_cpu_async(h_output.data(), 1, stream, 0);
    stream.synchronize();
    return h_output[0];
  }
};
// lwe_keyswitch.rs:558-577: a value whose decomposition state starts negative (a logical shift instead of an arithmetic one
// on the last level loses the sign when base_log * (level_count + 1) > 64)
static void test_closest_representable_gpu() {
  const uint32_t base_log = 17, level_count = 3;
  const u64 val = 0x800000e355b0c827ull;
  const u64 rounded = orc_closest_representable(val, base_log, level_count);
  std::vector<int64_t> digits(level_count);
  orc_decompose(val, base_log, level_count, digits.data());  // digits[0] <-> level `level_count`
  u64 recomp = 0;
  for (uint32_t i = 0; i < level_count; ++i) recomp += (u64)digits[i] << (64 - base_log * (level_count - i));
  CHECK_EQ(rounded, recomp);
  ClosestRepresentableOnGpu test_util_closest_representable_on_gpu;
  CHECK_EQ(test_util_closest_representable_on_gpu(val, base_log, level_count), rounded);
}
// lwe_keyswitch.rs:579-609: every valid decomposer (math/decomposition/tests.rs:15-30), random values: moving the GPU's
// result by less than half a representable step does not change its closest representable
static void test_round_to_closest_representable_gpu() {
  const size_t runs_per_decomposer = g_toy ? 3 : 100;
  TestResources rsc(41);
  ClosestRepresentableOnGpu test_util_closest_representable_on_gpu;
  for (uint32_t base_log = 1; base_log < 64; ++base_log)
    for (uint32_t level_count = 1; level_count < 64 && base_log * level_count < 64; ++level_count)
      for (size_t run = 0; run < runs_per_decomposer; ++run) {
        const u64 val = orc_rng_next(&rsc.encryption_random_generator);
        const u64 rounded = test_util_closest_representable_on_gpu(val, base_log, level_count);
        const u64 epsilon = (u64(1) << (64 - base_log * level_count - 1)) / 2;
        CHECK_EQ(rounded, orc_closest_representable(rounded + epsilon, base_log, level_count));
        CHECK_EQ(rounded, orc_closest_representable(rounded - epsilon, base_log, level_count));
        CHECK_EQ(rounded, orc_closest_representable(val, base_log, level_count));  // and it is the CPU's value (own addition)
      }
}

// glwe_sample_extraction.rs:14-149
static void glwe_encrypt_sample_extract_decrypt_custom_mod(const ClassicTestParams &params) {
  const size_t glwe_dimension = params.glwe_dimension, polynomial_size = params.polynomial_size;
  const CiphertextModulus ciphertext_modulus = CiphertextModulus::new_native();
  TestResources rsc(29);
  const u64 msg_modulus = u64(1) << params.message_modulus_log, delta = (u64(1) << 63) / msg_modulus;
  CudaStreams streams = CudaStreams::new_single_gpu(GpuIndex(0));
  std::vector<u64> msgs;
  for (u64 msg = msg_modulus - 1; msg != 0; --msg) msgs.push_back(msg);
  for (size_t t = 0; t < nb_tests(); ++t) {
    const std::vector<u64> glwe_sk = binary_key(rsc, glwe_dimension * polynomial_size);
    const std::vector<u64> &equivalent_lwe_sk = glwe_sk;
    const size_t glwe_words = (glwe_dimension + 1) * polynomial_size;
    std::vector<u64> glwe_list(msgs.size() * glwe_words, 0);
    for (size_t i = 0; i < msgs.size(); ++i) {  // encrypt_glwe_ciphertext_list: every coefficient holds msg * delta
      u64 *glwe = &glwe_list[i * glwe_words];
      for (size_t j = 0; j < polynomial_size; ++j) glwe[glwe_dimension * polynomial_size + j] = msgs[i] * delta;
      orc_glwe_encrypt_assign(&rsc.encryption_random_generator, glwe, glwe_sk.data(), (uint32_t)glwe_dimension, (uint32_t)polynomial_size,
                              key_noise_bound(params.glwe_noise_std));
    }
    const auto input_cuda_glwe_list =
        CudaGlweCiphertextList<u64>::from_glwe_ciphertext_list(glwe_list, glwe_dimension, polynomial_size, ciphertext_modulus, streams);
    const size_t lwe_per_glwe = 2;
    CudaLweCiphertextList<u64> output_cuda_lwe_ciphertext_list(equivalent_lwe_sk.size(), msgs.size() * lwe_per_glwe, ciphertext_modulus, streams);
    std::vector<uint32_t> nths(msgs.size() * lwe_per_glwe);
    for (size_t x = 0; x < nths.size(); ++x) nths[x] = (uint32_t)(x % polynomial_size);
    cuda_extract_lwe_samples_from_glwe_ciphertext_list(input_cuda_glwe_list, output_cuda_lwe_ciphertext_list, nths, (uint32_t)lwe_per_glwe, streams);
    const std::vector<u64> gpu_output = output_cuda_lwe_ciphertext_list.to_lwe_ciphertext_list(streams);
    CudaLweCiphertextList<u64> bis(equivalent_lwe_sk.size(), msgs.size() * lwe_per_glwe, ciphertext_modulus, streams);
    cuda_extract_lwe_samples_from_glwe_ciphertext_list(input_cuda_glwe_list, bis, nths, (uint32_t)lwe_per_glwe, streams);
    assert_gpu_determinism(gpu_output, bis.to_lwe_ciphertext_list(streams), "cuda_extract_lwe_samples_from_glwe_ciphertext_list");
    u64 count = msg_modulus - 1;
    for (size_t i = 0; i < msgs.size(); ++i, --count)
      for (size_t j = 0; j < lwe_per_glwe; ++j) {
        const u64 *ct = &gpu_output[(i * lwe_per_glwe + j) * (equivalent_lwe_sk.size() + 1)];
        CHECK_EQ(round_decode(orc_lwe_decrypt(ct, equivalent_lwe_sk.data(), (uint32_t)equivalent_lwe_sk.size()), delta) % msg_modulus, count);
      }
  }
}

// modulus_switch.rs:276-361 with the dimensions of :470-488 (COOPERATIVE_TEST_LWE_DIMENSIONS) and the first test's 800:
// the GPU's centered modulus switch of one ciphertext equals the CPU's word for word
static void compare_cpu_and_gpu_centered_modulus_switch() {
  const uint32_t log_modulus = 12;
  const CiphertextModulus ciphertext_modulus = CiphertextModulus::new_native();
  CudaStreams streams = CudaStreams::new_multi_gpu();
  TestResources rsc(31);
  for (size_t lwe_dimension : {size_t(100), size_t(512), size_t(742), size_t(800)}) {
    std::vector<u64> sk(lwe_dimension, 0);
    for (size_t i = 0; i < lwe_dimension; i += 2) sk[i] = 1;  // sk.iter_mut().step_by(2)
    const std::vector<u64> lwe = encrypt_lwe(rsc, sk, 0, 0.0);
    std::vector<u64> msed_container(lwe_dimension + 1);  // lwe_ciphertext_centered_binary_modulus_switch (algorithms/modulus_switch.rs:35-103)
    orc_lwe_modulus_switch(lwe.data(), (uint32_t)lwe_dimension, log_modulus, 1, msed_container.data());
    const auto d_lwe_input = CudaLweCiphertextList<u64>::from_lwe_ciphertext(lwe, ciphertext_modulus, streams);
    CudaLweCiphertextList<u64> d_lwe_output(lwe_dimension, 1, ciphertext_modulus, streams), d_lwe_output_bis(lwe_dimension, 1, ciphertext_modulus, streams);
    for (auto *out : {&d_lwe_output, &d_lwe_output_bis})
      cuda_centered_modulus_switch_64_async(streams.ptr[0], streams.gpu_indexes[0].get(), out->d_vec.as_mut_c_ptr(0), d_lwe_input.d_vec.as_c_ptr(0),
                                            (uint32_t)d_lwe_input.lwe_dimension(), log_modulus);
    const std::vector<u64> converted_gpu_ct = d_lwe_output.into_lwe_ciphertext(streams);
    assert_gpu_determinism(converted_gpu_ct, d_lwe_output_bis.into_lwe_ciphertext(streams), "cuda_centered_modulus_switch_64");
    CHECK(msed_container == converted_gpu_ct);
  }
}

// modulus_switch.rs:380-488: the cooperative correction (the one the bootstrap kernels run in their prologue, where a wrong
// correction is hidden by the decoding of the output) on its own, against the CPU and against the sequential GPU kernel,
// for the block of the throughput kernel (64, 2) and the generic one (512, 1); 10 ciphertexts per dimension, mask and
// body varied
static void check_cuda_cooperative_centered_modulus_switch(uint32_t dim_x, uint32_t dim_y, size_t lwe_dimension) {
  const size_t NB_TESTS = 10;
  const double lwe_noise_std = 0.000007069849454709433;
  const uint32_t log_modulus = 12;
  const CiphertextModulus ciphertext_modulus = CiphertextModulus::new_native();
  CudaStreams streams = CudaStreams::new_single_gpu(GpuIndex(0));
  TestResources rsc(37 + lwe_dimension + dim_x);
  const std::vector<u64> sk = binary_key(rsc, lwe_dimension);
  orc_rng random_generator;
  orc_rng_seed(&random_generator, 41 + lwe_dimension);
  enum Algorithm { Centered, CenteredCooperative };
  auto cuda_centered_modulus_switch = [&](Algorithm ms, const std::vector<u64> &lwe) {
    const auto d_lwe_input = CudaLweCiphertextList<u64>::from_lwe_ciphertext(lwe, ciphertext_modulus, streams);
    CudaLweCiphertextList<u64> d_lwe_output(lwe_dimension, 1, ciphertext_modulus, streams);
    if (ms == Centered)
      cuda_centered_modulus_switch_64_async(streams.ptr[0], streams.gpu_indexes[0].get(), d_lwe_output.d_vec.as_mut_c_ptr(0),
                                            d_lwe_input.d_vec.as_c_ptr(0), (uint32_t)lwe_dimension, log_modulus);
    else
      cuda_centered_modulus_switch_cooperative_64_async(streams.ptr[0], streams.gpu_indexes[0].get(), d_lwe_output.d_vec.as_mut_c_ptr(0),
                                                        d_lwe_input.d_vec.as_c_ptr(0), (uint32_t)lwe_dimension, log_modulus, dim_x, dim_y);
    return d_lwe_output.into_lwe_ciphertext(streams);
  };
  for (size_t t = 0; t < NB_TESTS; ++t) {
    const std::vector<u64> lwe = encrypt_lwe(rsc, sk, orc_rng_next(&random_generator), lwe_noise_std);
    std::vector<u64> cpu_container(lwe_dimension + 1);
    orc_lwe_modulus_switch(lwe.data(), (uint32_t)lwe_dimension, log_modulus, 1, cpu_container.data());
    const std::vector<u64> sequential_container = cuda_centered_modulus_switch(Centered, lwe);
    const std::vector<u64> cooperative_container = cuda_centered_modulus_switch(CenteredCooperative, lwe);
    const std::vector<u64> cooperative_container_bis = cuda_centered_modulus_switch(CenteredCooperative, lwe);
    assert_gpu_determinism(cooperative_container, cooperative_container_bis, "cuda_centered_modulus_switch_cooperative_64");
    CHECK(cpu_container == cooperative_container);
    CHECK(sequential_container == cooperative_container);
  }
}
static const size_t COOPERATIVE_TEST_LWE_DIMENSIONS[] = {100, 512, 742, 800};
static void compare_cpu_and_gpu_cooperative_centered_modulus_switch_throughput_pbs_block() {
  for (size_t lwe_dimension : COOPERATIVE_TEST_LWE_DIMENSIONS) check_cuda_cooperative_centered_modulus_switch(64, 2, lwe_dimension);
}
static void compare_cpu_and_gpu_cooperative_centered_modulus_switch_generic_block() {
  for (size_t lwe_dimension : COOPERATIVE_TEST_LWE_DIMENSIONS) check_cuda_cooperative_centered_modulus_switch(512, 1, lwe_dimension);
}

// ---- the transform's own entry points: the reference backend's C++ FFT tests and the Rust golden-spectrum regression
// compressed polynomial of the reference's tests: complex[i] = (p[i], p[i + N/2]) as [re, im, re, im, ...]
static std::vector<double> compress_poly(const std::vector<double> &p) {
  const size_t n = p.size();
  std::vector<double> c(n);
  for (size_t i = 0; i < n / 2; ++i) c[2 * i] = p[i], c[2 * i + 1] = p[i + n / 2];
  return c;
}
static std::vector<double> random_poly(orc_rng *r, size_t n) {  // fft_setup (setup_and_teardown.cpp:407-415): uniform in [-1, 1)
  std::vector<double> p(n);
  for (double &x : p) x = (double)(int64_t)orc_rng_next(r) * (1.0 / 9223372036854775808.0);
  return p;
}
// tests_and_benchmarks/tests/test_fft.cpp:82-146 `cuda_fft_mult`: the product of two random polynomials through
// cuda_fourier_polynomial_mul_async (output aliased onto input2, as there) against the schoolbook negacyclic product, 1e-9
static void cuda_fft_mult() {
  CudaStreams streams = CudaStreams::new_single_gpu(GpuIndex(0));
  orc_rng r;
  orc_rng_seed(&r, 97);
  struct P { size_t polynomial_size; int samples; };
  const std::vector<P> params = g_toy ? std::vector<P>{{256, 3}, {1024, 2}, {2048, 2}}
                                      : std::vector<P>{{256, 100}, {512, 100}, {1024, 100}, {2048, 100}, {4096, 100}, {8192, 50}, {16384, 10}};
  for (const P &pr : params) {
    const size_t N = pr.polynomial_size;
    std::vector<std::vector<double>> poly1, poly2;
    std::vector<double> h_cpoly1, h_cpoly2;
    for (int s = 0; s < pr.samples; ++s) {
      poly1.push_back(random_poly(&r, N));
      poly2.push_back(random_poly(&r, N));
      const auto c1 = compress_poly(poly1.back()), c2 = compress_poly(poly2.back());
      h_cpoly1.insert(h_cpoly1.end(), c1.begin(), c1.end());
      h_cpoly2.insert(h_cpoly2.end(), c2.begin(), c2.end());
    }
    auto d_cpoly1 = CudaVec<double>::from_cpu_async(h_cpoly1, streams, 0);
    auto d_cpoly2 = CudaVec<double>::from_cpu_async(h_cpoly2, streams, 0);
    cuda_fourier_polynomial_mul_async(streams.ptr[0], 0, d_cpoly1.as_mut_c_ptr(0), d_cpoly2.as_c_ptr(0), d_cpoly2.as_mut_c_ptr(0),
                                      (uint32_t)N, (uint32_t)pr.samples);
    const std::vector<double> res = d_cpoly2.to_cpu(streams);
    const int checked = N > 4096 ? 2 : pr.samples;  // the schoolbook product is quadratic
    for (int s = 0; s < checked; ++s) {
      std::vector<double> expected(2 * N, 0.0);
      for (size_t i = 0; i < N; ++i)
        for (size_t j = 0; j < N; ++j) expected[i + j] += poly1[s][i] * poly2[s][j];
      for (size_t i = 0; i < N; ++i) {
        const double want = expected[i] - expected[i + N];
        const double got = i < N / 2 ? res[(size_t)s * N + 2 * i] : res[(size_t)s * N + 2 * (i - N / 2) + 1];
        CHECK(std::fabs(got - want) < 1e-9);
      }
    }
  }
}
static size_t bitreverse_10(size_t x) {
  size_t r = 0;
  for (int i = 0; i < 10; i++) r = (r << 1) | ((x >> i) & 1u);
  return r;
}
// tests_and_benchmarks/tests/test_forward_fft16x4x16.cpp:118-152: natural order against the classic transform's native order
// through classic_index(f) = bitreverse_10((1024 - f) mod 1024), tolerance 2^-20
static void forward_matches_classic_fft() {
  CudaStreams streams = CudaStreams::new_single_gpu(GpuIndex(0));
  CHECK(cuda_fft16x4x16_is_supported_async(0));
  orc_rng r;
  orc_rng_seed(&r, 98);
  const size_t polynomial_size = 2048, half = 1024;
  const int samples = g_toy ? 3 : 100;
  std::vector<double> h_in;
  for (int s = 0; s < samples; ++s) {
    const auto c = compress_poly(random_poly(&r, polynomial_size));
    h_in.insert(h_in.end(), c.begin(), c.end());
  }
  const auto d_in = CudaVec<double>::from_cpu_async(h_in, streams, 0);
  CudaVec<double> d_out_fft16(polynomial_size * samples, streams, 0), d_out_classic(polynomial_size * samples, streams, 0);
  forward_fft16x4x16_async(streams, d_in, d_out_fft16, (uint32_t)polynomial_size, (uint32_t)samples);
  cuda_forward_fft_classic_async(streams.ptr[0], 0, d_in.as_c_ptr(0), d_out_classic.as_mut_c_ptr(0), (uint32_t)polynomial_size, (uint32_t)samples);
  const std::vector<double> h_fft16 = d_out_fft16.to_cpu(streams), h_classic = d_out_classic.to_cpu(streams);
  const double tol = std::pow(2.0, -20);
  for (int p = 0; p < samples; ++p)
    for (size_t f = 0; f < half; ++f) {
      const size_t c = bitreverse_10((half - f) % half);
      CHECK(std::fabs(h_fft16[((size_t)p * half + f) * 2] - h_classic[((size_t)p * half + c) * 2]) <= tol);
      CHECK(std::fabs(h_fft16[((size_t)p * half + f) * 2 + 1] - h_classic[((size_t)p * half + c) * 2 + 1]) <= tol);
    }
}
// gpu/algorithms/test/fft/mod.rs:268-294 `test_regression_fft16x4x16` on the golden spectrum of
// fft_data/fft16x4x16_golden_v1.rs (tests/golden/fft16x4x16_golden_v1.json; path in TFHE_FFT_GOLDEN).  The reference asserts
// the bits of its own kernel on an H100; another operation order has no bits in common to assert: values within 64 ulp of the
// spectrum's scale (measured: 0.3).
static void test_regression_fft16x4x16() {
  const char *path = std::getenv("TFHE_FFT_GOLDEN");
  CHECK(path != nullptr);
  std::FILE *fp = std::fopen(path, "r");
  CHECK(fp != nullptr);
  std::string text;
  char buf[4096];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, got);
  std::fclose(fp);
  std::vector<u64> words;
  for (size_t pos = text.find("\"0x"); pos != std::string::npos; pos = text.find("\"0x", pos + 1))
    words.push_back(std::strtoull(text.c_str() + pos + 1, nullptr, 16));
  const size_t n = 2048, half = 1024;
  CHECK(words.size() == 2 * half);
  std::vector<double> input(n);  // fft16x4x16_reference_input (mod.rs:51-71)
  auto poly_coeff = [](u64 k) {
    u64 bits = k * 0x517cc1b727220a95ull;
    bits = (bits << 17) | (bits >> 47);
    bits ^= 0xdeadbeefcafebabeull;
    return (double)(int64_t)bits / (double)INT64_MAX;
  };
  for (size_t i = 0; i < half; ++i) input[2 * i] = poly_coeff(i), input[2 * i + 1] = poly_coeff(i + half);
  CudaStreams stream = CudaStreams::new_single_gpu(GpuIndex(0));
  const auto d_input = CudaVec<double>::from_cpu_async(input, stream, 0);
  CudaVec<double> d_output(n, stream, 0);
  forward_fft16x4x16_async(stream, d_input, d_output, (uint32_t)n, 1);
  const std::vector<double> output = d_output.to_cpu(stream);
  double scale = 0, worst = 0;
  for (size_t i = 0; i < half; ++i) {
    double re, im;
    std::memcpy(&re, &words[i], 8);
    std::memcpy(&im, &words[half + i], 8);
    scale = std::max(scale, std::hypot(re, im));
    worst = std::max(worst, std::hypot(output[2 * i] - re, output[2 * i + 1] - im));
  }
  CHECK(worst < 64 * 2.220446049250313e-16 * scale);
}

// the `assert_eq!`s in front of every launch (gpu/algorithms/*.rs): mismatched operands panic before anything is enqueued
static void mismatched_dimensions_panic() {
  const CiphertextModulus m = CiphertextModulus::new_native();
  CudaStreams stream = CudaStreams::new_single_gpu(GpuIndex(0));
  const size_t n = 8, k = 1, N = 256, l = 1;
  std::vector<u64> bsk(n * (k + 1) * (k + 1) * l * N, 0), ksk(k * N * 2 * (n + 1), 0);
  const auto d_bsk = CudaLweBootstrapKey::from_lwe_bootstrap_key(bsk, n, k, N, 12, l, false, stream);
  const auto d_ksk = CudaLweKeyswitchKey<u64>::from_lwe_keyswitch_key(ksk, k * N, n, 4, 2, stream);
  CudaGlweCiphertextList<u64> acc(k, N, 1, m, stream), acc_bad(k, 2 * N, 1, m, stream);
  CudaLweCiphertextList<u64> in(n, 1, m, stream), in_bad(n + 1, 1, m, stream), out(k * N, 1, m, stream), out_bad(k * N - 1, 1, m, stream);
  const CudaVec<u64> idx = device_indexes({0}, stream);
  auto panics = [](const std::function<void()> &body, const char *expect) {
    try {
      body();
    } catch (const Panic &p) {
      if (std::strstr(p.what(), expect)) return true;
      throw std::runtime_error(std::string("wrong panic message: ") + p.what());
    }
    return false;
  };
  CHECK(panics([&] { cuda_programmable_bootstrap_lwe_ciphertext(in_bad, out, acc, idx, idx, idx, d_bsk, stream); }, "Mismatched input LweDimension"));
  CHECK(panics([&] { cuda_programmable_bootstrap_lwe_ciphertext(in, out_bad, acc, idx, idx, idx, d_bsk, stream); }, "Mismatched output LweDimension"));
  CHECK(panics([&] { cuda_programmable_bootstrap_lwe_ciphertext(in, out, acc_bad, idx, idx, idx, d_bsk, stream); }, "Mismatched PolynomialSize"));
  CHECK(panics([&] { cuda_keyswitch_lwe_ciphertext(d_ksk, in, in, idx, idx, true, stream, false); }, "Mismatched input LweDimension"));
  CHECK(panics([&] { cuda_keyswitch_lwe_ciphertext(d_ksk, out, out, idx, idx, true, stream, true); }, "Mismatched output LweDimension"));
  CHECK(panics([&] { cuda_extract_lwe_samples_from_glwe_ciphertext_list(acc, out_bad, {0}, 1, stream); }, "Mismatch between equivalent LweDimension"));
  CHECK(panics([&] { CudaVec<u64> v = CudaVec<u64>::new_async(2, stream, 0); v.copy_from_cpu_async(std::vector<u64>(3, 0), stream, 0); },
               "self.len() >= src.len()"));
  // and the well-formed call goes through (zero key: the output is the LUT's first coefficient, nothing to decrypt)
  CudaLweCiphertextList<u64> ok_out(k * N, 1, m, stream);
  cuda_programmable_bootstrap_lwe_ciphertext(in, ok_out, acc, idx, idx, idx, d_bsk, stream);
  stream.synchronize();
}

// ---------------------------------------------------------------------------------------------------------------------
struct Test {
  std::string name;
  std::function<void()> body;
};

int main(int argc, char **argv) {
  if (argc < 2 || (std::strcmp(argv[1], "toy") && std::strcmp(argv[1], "reference"))) {
    std::fprintf(stderr, "usage: %s <toy|reference> [test-name-substring]\n", argv[0]);
    return 2;
  }
  g_toy = !std::strcmp(argv[1], "toy");
  const char *filter = argc > 2 ? argv[2] : "";
  if (!is_cuda_available()) {
    std::fprintf(stderr, "no device visible: the backend has no CPU path\n");
    return 2;
  }
  std::vector<Test> tests;
  auto classic = [&](const ClassicTestParams &p) {  // create_gpu_parameterized_test! (test/mod.rs:104-124)
    tests.push_back({std::string("test_gpu_lwe_encrypt_pbs_decrypt_") + p.name, [&p] { lwe_encrypt_pbs_decrypt_impl(p, false); }});
    tests.push_back({std::string("test_gpu_lwe_encrypt_centered_ms_pbs_decrypt_") + p.name, [&p] { lwe_encrypt_pbs_decrypt_impl(p, true); }});
    tests.push_back({std::string("test_gpu_lwe_encrypt_ks_decrypt_custom_mod_") + p.name, [&p] { lwe_encrypt_ks_decrypt_custom_mod(p); }});
    tests.push_back({std::string("test_gpu_glwe_encrypt_sample_extract_decrypt_custom_mod_") + p.name,
                     [&p] { glwe_encrypt_sample_extract_decrypt_custom_mod(p); }});
  };
  auto multi_bit = [&](const MultiBitTestParams &p, bool ks) {  // create_gpu_multi_bit_parameterized_test! (test/mod.rs:125-146)
    tests.push_back({std::string("test_gpu_lwe_encrypt_multi_bit_pbs_decrypt_custom_mod_") + p.name,
                     [&p] { lwe_encrypt_multi_bit_pbs_decrypt_custom_mod(p); }});
    if (ks)
      tests.push_back({std::string("test_gpu_lwe_encrypt_ks_decrypt_custom_mod_mb_") + p.name, [&p] { lwe_encrypt_ks_decrypt_custom_mod_mb(p); }});
  };
  auto mb_noise = [&](const MultiBitTestParams &p) {
    tests.push_back({std::string("noise_tests_multi_bit_mod_switch_then_blind_rotation_") + p.name, [&p] { multi_bit_mod_switch_then_blind_rotation(p); }});
  };
  if (g_toy) {
    classic(TOY_4_BITS_N2048);
    classic(TOY_2_BITS_K2_N256);
    multi_bit(TOY_MB_2, true);
    multi_bit(TOY_MB_3, false);
    multi_bit(TOY_MB_4, false);
    mb_noise(TOY_MB_4);
  } else {
    mb_noise(MULTI_BIT_2_2_3_PARAMS);
    mb_noise(MULTI_BIT_2_2_4_PARAMS);
    classic(TEST_PARAMS_4_BITS_NATIVE_U64);
    multi_bit(MULTI_BIT_2_2_2_PARAMS, true);
    multi_bit(MULTI_BIT_2_2_3_PARAMS, true);
    multi_bit(MULTI_BIT_2_2_4_PARAMS, false);  // the keyswitch list of the reference has no g = 4 set (test/mod.rs:135-141)
    // MULTI_BIT_3_3_2 / 3_3_3 (N = 8192): the CPU-side key generation of the checker takes minutes; run by name only
    if (std::strstr(filter, "3_3_2")) multi_bit(MULTI_BIT_3_3_2_PARAMS, true);
  }
  tests.push_back({"test_gpu_lwe_encrypt_ks_decrypt_custom_mod_ks32_multi_bit_2_2_2_ks32_params", lwe_encrypt_ks_decrypt_custom_mod_ks32});
  tests.push_back({"test_closest_representable_gpu", test_closest_representable_gpu});
  tests.push_back({"test_round_to_closest_representable_gpu", test_round_to_closest_representable_gpu});
  tests.push_back({"compare_cpu_and_gpu_centered_modulus_switch", compare_cpu_and_gpu_centered_modulus_switch});
  tests.push_back({"compare_cpu_and_gpu_cooperative_centered_modulus_switch_throughput_pbs_block",
                   compare_cpu_and_gpu_cooperative_centered_modulus_switch_throughput_pbs_block});
  tests.push_back({"compare_cpu_and_gpu_cooperative_centered_modulus_switch_generic_block",
                   compare_cpu_and_gpu_cooperative_centered_modulus_switch_generic_block});
  tests.push_back({"cuda_fft_mult", cuda_fft_mult});
  tests.push_back({"forward_matches_classic_fft", forward_matches_classic_fft});
  tests.push_back({"test_regression_fft16x4x16", test_regression_fft16x4x16});
  tests.push_back({"mismatched_dimensions_panic", mismatched_dimensions_panic});

  size_t ran = 0, failed = 0;
  for (const Test &t : tests) {
    if (!std::strstr(t.name.c_str(), filter)) continue;
    ++ran;
    const auto t0 = std::chrono::steady_clock::now();
    std::string err;
    try {
      t.body();
    } catch (const std::exception &e) {
      err = e.what();
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("test %s ... %s (%.1f s)%s%s\n", t.name.c_str(), err.empty() ? "ok" : "FAILED", s, err.empty() ? "" : ": ", err.c_str());
    std::fflush(stdout);
    failed += !err.empty();
  }
  std::printf("test result: %s. %zu passed; %zu failed\n", failed ? "FAILED" : "ok", ran - failed, failed);
  return failed || !ran ? 1 : 0;
}
