// reference_integer_gpu_tests.cpp — the reference's GPU tests of the radix operations the backend wires,
// restated in C++ on the compiled host mirror tfhe_rs_amd/host/integer_gpu.hpp (C ABI only) and linked against the
// library.  Each test follows the generic test case the Rust GPU test instantiates with `GpuFunctionExecutor`:
//
//   integer_unchecked_add             integer/gpu/server_key/radix/tests_unsigned/test_add.rs:15-21 -> unchecked_add_test
//                                     (integer/server_key/radix_parallel/tests_unsigned/test_add.rs:276-330)
//   integer_add                       …/test_add.rs:29-35 -> default_add_test (…/test_add.rs:448-505)
//   multi_device_integer_add          …/test_add.rs:36-42 -> default_add_test through GpuMultiDeviceFunctionExecutor (only with > 1 GPU)
//   integer_default_overflowing_add   …/test_add.rs:44 -> default_overflowing_add_test (…/test_add.rs:553-680)
//   integer_mul                       integer/gpu/server_key/radix/tests_unsigned/test_mul.rs -> default_mul_test
//                                     (integer/server_key/radix_parallel/tests_cases_unsigned.rs:865-927)
//   integer_sub / bitop / comparison / if_then_else / scalar_shift   (round 6) tests_unsigned/{test_sub.rs:301-346, test_bitwise_op.rs,
//                                     test_comparison.rs, test_cmux.rs, test_scalar_shift.rs} default_* test cases
//   constants NB_CTXT = 4, MAX_NB_CTXT = 8, nb_tests(_smaller)_for_params   …/tests_unsigned/mod.rs:61-125
//
// on PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 and TEST_PARAM_GPU_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128
// (the first and third set of the reference's lists); "toy": small sets with the same shapes for the host emulation.
// Test infrastructure: client-side key generation, radix encryption and decryption use the oracle's primitives
// (oracle/tfhe_oracle.h); randomness is seeded.
//   usage: reference_integer_gpu_tests <toy|reference> [test-name-substring]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../oracle/tfhe_oracle.h"
#include "../../tfhe_rs_amd/host/integer_gpu.hpp"

using namespace tfhe::core_crypto::gpu;
using namespace tfhe::integer::gpu;
using u64 = uint64_t;

#define CHECK(c)                                                                                                        \
  do {                                                                                                                  \
    if (!(c)) throw std::runtime_error(std::string("assertion failed: ") + #c + " at line " + std::to_string(__LINE__)); \
  } while (0)
#define CHECK_EQ(a, b)                                                                                                             \
  do {                                                                                                                             \
    auto va = (a);                                                                                                                 \
    auto vb = (b);                                                                                                                 \
    if (!(va == vb))                                                                                                               \
      throw std::runtime_error(std::string("assertion `left == right` failed: ") + #a + " = " + std::to_string(va) + ", " + #b + \
                               " = " + std::to_string(vb) + " at line " + std::to_string(__LINE__));                               \
  } while (0)

struct TestParameters {  // shortint parameters (KS-PBS order, encryption under the big key)
  const char *name;
  size_t lwe_dimension, glwe_dimension, polynomial_size;
  uint32_t lwe_noise, glwe_noise;  // TUniform bounds
  size_t pbs_base_log, pbs_level, ks_base_log, ks_level;
  u64 message_modulus, carry_modulus;
  size_t grouping_factor;  // 0: classic PBS
  bool centered_ms;        // ModulusSwitchType::CenteredMeanNoiseReduction
};
// shortint/parameters/v1_*/classic/tuniform/p_fail_2_minus_128/ks_pbs.rs:28-47, v1_*/multi_bit/tuniform/p_fail_2_minus_128/ks_pbs_gpu.rs:205-228
static const TestParameters PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 = {"param_message_2_carry_2_ks_pbs_tuniform_2m128", 918, 1, 2048, 45, 17, 23, 1, 4, 4,
                                                                           4, 4, 0, true};
static const TestParameters TEST_PARAM_GPU_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 = {
    "test_param_gpu_multi_bit_group_4_message_2_carry_2_ks_pbs_tuniform_2m128", 920, 1, 2048, 45, 17, 22, 1, 3, 5, 4, 4, 4, false};
static const TestParameters TOY_MESSAGE_2_CARRY_2 = {"toy_message_2_carry_2", 12, 1, 2048, 45, 17, 23, 1, 4, 4, 4, 4, 0, true};
static const TestParameters TOY_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2 = {"toy_multi_bit_group_4_message_2_carry_2", 8, 1, 2048, 45, 17, 22, 1, 3, 6, 4, 4, 4,
                                                                      false};
static bool g_toy = false;  // small parameter sets AND short loops; REFERENCE_TESTS_SHORT_LOOPS=1 keeps the reference's sets with the short loops
constexpr size_t NB_CTXT = 4, MAX_NB_CTXT = 8;
static size_t nb_tests_for_params(const TestParameters &p) {
  const u64 full = p.message_modulus * p.carry_modulus;
  return g_toy ? 2 : full >= 256 ? 5 : full >= 64 ? 15 : 30;
}
static size_t nb_tests_smaller_for_params(const TestParameters &p) {
  const u64 full = p.message_modulus * p.carry_modulus;
  return g_toy ? 1 : full >= 256 ? 2 : full >= 64 ? 5 : 10;
}
static u64 unsigned_modulus(u64 message_modulus, uint32_t num_blocks) {  // message_modulus^num_blocks (< 2^64 here)
  u64 m = 1;
  for (uint32_t i = 0; i < num_blocks; ++i) m *= message_modulus;
  return m;
}

// RadixClientKey (integer/client_key/radix.rs) + the server key on the device, from one seed (KEY_CACHE's role)
struct Keys {
  TestParameters p;
  orc_rng rng;
  std::vector<u64> small_sk, big_sk;
  u64 delta;
  CudaStreams streams;
  std::unique_ptr<CudaServerKey> sks;
  explicit Keys(const TestParameters &params, u64 seed, CudaStreams set = CudaStreams::new_single_gpu(GpuIndex(0))) : p(params), streams(std::move(set)) {
    orc_rng_seed(&rng, seed);
    const size_t n = p.lwe_dimension, k = p.glwe_dimension, N = p.polynomial_size, g = p.grouping_factor;
    small_sk.resize(n);
    big_sk.resize(k * N);
    orc_gen_binary_key(&rng, small_sk.data(), (uint32_t)n);
    orc_gen_binary_key(&rng, big_sk.data(), (uint32_t)(k * N));
    delta = (u64(1) << 63) / (p.message_modulus * p.carry_modulus);
    std::vector<u64> ksk(k * N * p.ks_level * (n + 1));
    orc_gen_ksk(seed + 2, ksk.data(), big_sk.data(), (uint32_t)(k * N), small_sk.data(), (uint32_t)n, (uint32_t)p.ks_base_log, (uint32_t)p.ks_level, p.lwe_noise);
    auto d_ksk = CudaLweKeyswitchKey<u64>::from_lwe_keyswitch_key(ksk, k * N, n, p.ks_base_log, p.ks_level, streams);
    std::vector<u64> bsk((g ? (n / g) * (size_t(1) << g) : n) * (k + 1) * (k + 1) * p.pbs_level * N);
    if (g) {
      orc_gen_multi_bit_bsk(seed + 1, bsk.data(), small_sk.data(), (uint32_t)n, big_sk.data(), (uint32_t)k, (uint32_t)N, (uint32_t)p.pbs_base_log,
                            (uint32_t)p.pbs_level, (uint32_t)g, p.glwe_noise);
      sks = std::make_unique<CudaServerKey>(std::move(d_ksk),
                                            CudaLweMultiBitBootstrapKey::from_lwe_multi_bit_bootstrap_key(bsk, n, k, N, p.pbs_base_log, p.pbs_level, g, streams),
                                            p.message_modulus, p.carry_modulus);
    } else {
      orc_gen_bsk(seed + 1, bsk.data(), small_sk.data(), (uint32_t)n, big_sk.data(), (uint32_t)k, (uint32_t)N, (uint32_t)p.pbs_base_log, (uint32_t)p.pbs_level,
                  p.glwe_noise);
      sks = std::make_unique<CudaServerKey>(std::move(d_ksk),
                                            CudaLweBootstrapKey::from_lwe_bootstrap_key(bsk, n, k, N, p.pbs_base_log, p.pbs_level, p.centered_ms, streams),
                                            p.message_modulus, p.carry_modulus);
    }
  }
  u64 random() { return orc_rng_next(&rng); }
  // encrypt_radix: block i holds digit i of `clear` in base message_modulus, under the big key
  CudaUnsignedRadixCiphertext encrypt_radix(u64 clear, size_t num_blocks) {
    std::vector<u64> blocks(num_blocks * (big_sk.size() + 1));
    for (size_t i = 0; i < num_blocks; ++i, clear /= p.message_modulus)
      orc_lwe_encrypt(&rng, &blocks[i * (big_sk.size() + 1)], big_sk.data(), (uint32_t)big_sk.size(), (clear % p.message_modulus) * delta, p.glwe_noise);
    return CudaUnsignedRadixCiphertext::from_radix_ciphertext(blocks, big_sk.size(), p.message_modulus, p.carry_modulus, p.message_modulus - 1, streams);
  }
  CudaUnsignedRadixCiphertext encrypt(u64 clear) { return encrypt_radix(clear, NB_CTXT); }
  // create_trivial_radix: zero masks, bodies hold the digits; degrees are the digits themselves
  CudaUnsignedRadixCiphertext create_trivial_radix(u64 clear, size_t num_blocks) {
    std::vector<u64> blocks(num_blocks * (big_sk.size() + 1), 0);
    std::vector<u64> digits;
    for (size_t i = 0; i < num_blocks; ++i, clear /= p.message_modulus) {
      blocks[i * (big_sk.size() + 1) + big_sk.size()] = (clear % p.message_modulus) * delta;
      digits.push_back(clear % p.message_modulus);
    }
    auto ct = CudaUnsignedRadixCiphertext::from_radix_ciphertext(blocks, big_sk.size(), p.message_modulus, p.carry_modulus, 0, streams);
    ct.degrees = digits;
    return ct;
  }
  // ServerKey::unchecked_scalar_add on the client's copy: digit i of the scalar enters block i's body (makes blocks non-clean)
  CudaUnsignedRadixCiphertext unchecked_scalar_add(const CudaUnsignedRadixCiphertext &ct, u64 scalar) {
    std::vector<u64> blocks = ct.to_radix_ciphertext(streams);
    auto out = CudaUnsignedRadixCiphertext::from_radix_ciphertext(blocks, ct.lwe_dimension, p.message_modulus, p.carry_modulus, 0, streams);
    out.degrees = ct.degrees;
    out.noise_levels = ct.noise_levels;
    for (size_t i = 0; i < ct.num_blocks(); ++i, scalar /= p.message_modulus) {
      blocks[i * (ct.lwe_dimension + 1) + ct.lwe_dimension] += (scalar % p.message_modulus) * delta;
      out.degrees[i] += scalar % p.message_modulus;
    }
    out.d_blocks.copy_from_cpu_async(blocks, streams, 0);
    streams.synchronize();
    return out;
  }
  std::vector<u64> decrypt_blocks(const CudaUnsignedRadixCiphertext &ct) {  // message and carry of every block
    const std::vector<u64> blocks = ct.to_radix_ciphertext(streams);
    std::vector<u64> out;
    for (size_t i = 0; i < ct.num_blocks(); ++i) {
      const u64 ph = orc_lwe_decrypt(&blocks[i * (ct.lwe_dimension + 1)], big_sk.data(), (uint32_t)big_sk.size());
      out.push_back((ph / delta + (ph % delta >= (delta >> 1))) % (p.message_modulus * p.carry_modulus));
    }
    return out;
  }
  u64 decrypt(const CudaUnsignedRadixCiphertext &ct) {  // RadixClientKey::decrypt: sum of block values * message_modulus^i, wrapping
    u64 v = 0, w = 1;
    for (u64 b : decrypt_blocks(ct)) v += b * w, w *= p.message_modulus;
    return v % unsigned_modulus(p.message_modulus, (uint32_t)ct.num_blocks());
  }
  bool decrypt_bool(const CudaBooleanBlock &b) {
    const u64 v = decrypt_blocks(b)[0];
    CHECK(v <= 1);
    return v == 1;
  }
  void panic_if_any_block_is_not_clean(const CudaUnsignedRadixCiphertext &ct) {  // tests_unsigned/mod.rs
    const std::vector<u64> blocks = decrypt_blocks(ct);
    for (size_t i = 0; i < blocks.size(); ++i) {
      CHECK(ct.degrees[i] < p.message_modulus);
      CHECK(blocks[i] < p.message_modulus);
    }
  }
};
// KEY_CACHE.get_from_params (integer/keycache.rs): one key set per parameter set for the whole run
static Keys &key_cache(const TestParameters &param) {
  static std::map<std::string, std::unique_ptr<Keys>> cache;
  auto &slot = cache[param.name];
  if (!slot) slot = std::make_unique<Keys>(param, 101);
  return *slot;
}
static void assert_same_ciphertext(const Keys &k, const CudaUnsignedRadixCiphertext &a, const CudaUnsignedRadixCiphertext &b, const char *what) {
  if (a.to_radix_ciphertext(k.streams) != b.to_radix_ciphertext(k.streams) || a.degrees != b.degrees)
    throw std::runtime_error(std::string("Failed determinism check: ") + what);
}

// unchecked_add_test (test_add.rs:276-330)
static void integer_unchecked_add(const TestParameters &param) {
  Keys &k = key_cache(param);
  const u64 modulus = unsigned_modulus(param.message_modulus, NB_CTXT);
  for (size_t t = 0; t < nb_tests_for_params(param); ++t) {
    const u64 clear_0 = k.random() % modulus, clear_1 = k.random() % modulus;
    const auto ctxt_0 = k.encrypt(clear_0), ctxt_1 = k.encrypt(clear_1);
    const auto encrypted_result = k.sks->unchecked_add(ctxt_0, ctxt_1, k.streams);
    for (size_t i = 0; i < NB_CTXT; ++i) {  // ExpectedDegrees / ExpectedNoiseLevels::after_unchecked_add
      CHECK_EQ(encrypted_result.degrees[i], ctxt_0.degrees[i] + ctxt_1.degrees[i]);
      CHECK_EQ(encrypted_result.noise_levels[i], ctxt_0.noise_levels[i] + ctxt_1.noise_levels[i]);
    }
    const std::vector<u64> blocks = k.decrypt_blocks(encrypted_result);
    for (size_t i = 0; i < NB_CTXT; ++i) CHECK(blocks[i] <= encrypted_result.degrees[i]);  // panic_if_any_block_values_exceeds_its_degree
    CHECK_EQ(k.decrypt(encrypted_result), (clear_0 + clear_1) % modulus);
  }
}

// default_add_test (test_add.rs:448-505)
static void integer_add_on(Keys &k, const TestParameters &param);
static void integer_add(const TestParameters &param) { integer_add_on(key_cache(param), param); }
static void integer_add_on(Keys &k, const TestParameters &param) {
  const size_t nb_tests_smaller = nb_tests_smaller_for_params(param);
  for (size_t num_blocks = 1; num_blocks < (g_toy ? 4 : MAX_NB_CTXT); ++num_blocks) {
    const u64 modulus = unsigned_modulus(param.message_modulus, (uint32_t)num_blocks);
    const u64 clear_0 = k.random() % modulus, clear_1 = k.random() % modulus;
    const auto ctxt_0 = k.encrypt_radix(clear_0, num_blocks), ctxt_1 = k.encrypt_radix(clear_1, num_blocks);
    auto ct_res = k.sks->add(ctxt_0, ctxt_1, k.streams);
    const auto tmp_ct = k.sks->add(ctxt_0, ctxt_1, k.streams);
    k.panic_if_any_block_is_not_clean(ct_res);
    assert_same_ciphertext(k, ct_res, tmp_ct, "add");
    u64 clear = (clear_0 + clear_1) % modulus;
    CHECK_EQ(k.decrypt(ct_res), clear);
    for (size_t t = 0; t < nb_tests_smaller; ++t) {
      ct_res = k.sks->add(ct_res, ctxt_0, k.streams);
      k.panic_if_any_block_is_not_clean(ct_res);
      clear = (clear + clear_0) % modulus;
      CHECK_EQ(k.decrypt(ct_res), clear);
    }
  }
}

// multi_device_integer_add (tests_unsigned/test_add.rs:36-42): default_add_test through GpuMultiDeviceFunctionExecutor
// (tests_signed/mod.rs:653-693) — the server key on a random subset of the GPUs in a random order, one stream each.  The
// reference's thresholds keep a handful of blocks on the first GPU of the set; the spreading threshold is lowered to one
// block per GPU so that the blocks of a round really travel (hip_integer_set_multi_gpu_threshold).
static void integer_add_on(Keys &k, const TestParameters &param);
static void multi_device_integer_add(const TestParameters &param) {
  const uint32_t num_gpus = get_number_of_gpus();
  orc_rng r;
  orc_rng_seed(&r, 113);
  std::vector<uint32_t> all(num_gpus);
  for (uint32_t i = 0; i < num_gpus; ++i) all[i] = i;
  for (uint32_t i = num_gpus; i > 1; --i) std::swap(all[i - 1], all[orc_rng_next(&r) % i]);
  const uint32_t num_gpus_to_use = num_gpus > 1 ? 2 + (uint32_t)(orc_rng_next(&r) % (num_gpus - 1)) : 1;  // at least two when there are two
  std::vector<GpuIndex> gpu_indexes;
  std::string listed;
  for (uint32_t i = 0; i < num_gpus_to_use; ++i) gpu_indexes.emplace_back(all[i]), listed += " " + std::to_string(all[i]);
  std::printf("Setting up server key on GPUs: [%s ]\n", listed.c_str());
  Keys k(param, 113, CudaStreams::new_multi_gpu_with_indexes(gpu_indexes));
  hip_integer_set_multi_gpu_threshold(1);
  try {
    integer_add_on(k, param);
  } catch (...) {
    hip_integer_set_multi_gpu_threshold(0);
    throw;
  }
  hip_integer_set_multi_gpu_threshold(0);
}

// default_overflowing_add_test (test_add.rs:553-680)
static void integer_default_overflowing_add(const TestParameters &param) {
  Keys &k = key_cache(param);
  const size_t nb_tests_smaller = nb_tests_smaller_for_params(param);
  auto expect = [](u64 a, u64 b, u64 modulus) { return std::make_pair((a + b) % modulus, a + b >= modulus); };  // overflowing_add_under_modulus
  for (size_t num_blocks = 1; num_blocks < (g_toy ? 3 : MAX_NB_CTXT); ++num_blocks) {
    const u64 modulus = unsigned_modulus(param.message_modulus, (uint32_t)num_blocks);
    const u64 clear_0 = k.random() % modulus, clear_1 = k.random() % modulus;
    const auto ctxt_0 = k.encrypt_radix(clear_0, num_blocks), ctxt_1 = k.encrypt_radix(clear_1, num_blocks);
    const auto [ct_res, result_overflowed] = k.sks->unsigned_overflowing_add(ctxt_0, ctxt_1, k.streams);
    const auto [tmp_ct, tmp_o] = k.sks->unsigned_overflowing_add(ctxt_0, ctxt_1, k.streams);
    k.panic_if_any_block_is_not_clean(ct_res);
    assert_same_ciphertext(k, ct_res, tmp_ct, "unsigned_overflowing_add");
    assert_same_ciphertext(k, result_overflowed, tmp_o, "unsigned_overflowing_add (flag)");
    const auto [expected_result, expected_overflowed] = expect(clear_0, clear_1, modulus);
    CHECK_EQ(k.decrypt(ct_res), expected_result);
    CHECK_EQ(k.decrypt_bool(result_overflowed), expected_overflowed);
    CHECK_EQ(result_overflowed.degrees[0], u64(1));
    for (size_t t = 0; t < nb_tests_smaller; ++t) {  // non-zero scalars make the operands non-clean
      const u64 clear_2 = 1 + k.random() % (modulus - 1), clear_3 = 1 + k.random() % (modulus - 1);
      const auto nc_0 = k.unchecked_scalar_add(ctxt_0, clear_2), nc_1 = k.unchecked_scalar_add(ctxt_1, clear_3);
      const u64 clear_lhs = (clear_0 + clear_2) % modulus, clear_rhs = (clear_1 + clear_3) % modulus;
      CHECK_EQ(k.decrypt(nc_0), clear_lhs);  // "Failed sanity decryption check"
      CHECK_EQ(k.decrypt(nc_1), clear_rhs);
      const auto [res, overflowed] = k.sks->unsigned_overflowing_add(nc_0, nc_1, k.streams);
      k.panic_if_any_block_is_not_clean(res);
      const auto [want, want_overflowed] = expect(clear_lhs, clear_rhs, modulus);
      CHECK_EQ(k.decrypt(res), want);
      CHECK_EQ(k.decrypt_bool(overflowed), want_overflowed);
      CHECK_EQ(overflowed.degrees[0], u64(1));
    }
  }
  const u64 modulus = unsigned_modulus(param.message_modulus, NB_CTXT);  // trivial inputs
  for (int t = 0; t < (g_toy ? 1 : 4); ++t) {
    const u64 clear_0 = k.random() % modulus, clear_1 = k.random() % modulus;
    const auto a = k.create_trivial_radix(clear_0, NB_CTXT), b = k.create_trivial_radix(clear_1, NB_CTXT);
    const auto [encrypted_result, encrypted_overflow] = k.sks->unsigned_overflowing_add(a, b, k.streams);
    const auto [want, want_overflowed] = expect(clear_0, clear_1, modulus);
    CHECK_EQ(k.decrypt(encrypted_result), want);
    CHECK_EQ(k.decrypt_bool(encrypted_overflow), want_overflowed);
  }
}

// default_mul_test (tests_cases_unsigned.rs:865-927)
static void integer_mul(const TestParameters &param) {
  Keys &k = key_cache(param);
  const size_t nb_tests_smaller = nb_tests_smaller_for_params(param);
  const u64 modulus = unsigned_modulus(param.message_modulus, NB_CTXT);
  for (size_t outer = 0; outer < nb_tests_smaller; ++outer) {
    const u64 clear1 = k.random() % modulus, clear2 = k.random() % modulus;
    const auto ctxt_1 = k.encrypt(clear1), ctxt_2 = k.encrypt(clear2);
    u64 clear = clear1;
    auto res = k.sks->mul(ctxt_1, ctxt_2, k.streams);
    CHECK(res.block_carries_are_empty());
    for (size_t t = 0; t < nb_tests_smaller; ++t) {
      const auto tmp = k.sks->mul(res, ctxt_2, k.streams);
      res = k.sks->mul(res, ctxt_2, k.streams);
      CHECK(res.block_carries_are_empty());
      assert_same_ciphertext(k, res, tmp, "mul");
      clear = (clear * clear2) % modulus;
    }
    clear = (clear * clear2) % modulus;
    CHECK_EQ(k.decrypt(res), clear);
  }
  {  // x * y and y * x where y encrypts a boolean value
    const u64 clear1 = k.random() % modulus, clear2 = k.random() & 1;
    const auto ctxt_1 = k.encrypt(clear1), ctxt_2 = k.create_trivial_radix(clear2, NB_CTXT);
    CHECK(ctxt_2.holds_boolean_value());
    CHECK_EQ(k.decrypt(k.sks->mul(ctxt_1, ctxt_2, k.streams)), clear1 * clear2);
    CHECK_EQ(k.decrypt(k.sks->mul(ctxt_2, ctxt_1, k.streams)), clear1 * clear2);
  }
}

// default_sub_test (tests_unsigned/test_sub.rs:301-346)
static void integer_sub(const TestParameters &param) {
  Keys &k = key_cache(param);
  const size_t nb_tests_smaller = nb_tests_smaller_for_params(param);
  for (size_t num_blocks = 1; num_blocks < (g_toy ? 4 : MAX_NB_CTXT); ++num_blocks) {
    const u64 modulus = unsigned_modulus(param.message_modulus, (uint32_t)num_blocks);
    const u64 clear1 = k.random() % modulus, clear2 = k.random() % modulus;
    const auto ctxt_1 = k.encrypt_radix(clear1, num_blocks), ctxt_2 = k.encrypt_radix(clear2, num_blocks);
    auto res = ctxt_1.duplicate(k.streams);
    u64 clear = clear1;
    for (size_t t = 0; t < nb_tests_smaller; ++t) {  // subtract multiple times
      const auto tmp = k.sks->sub(res, ctxt_2, k.streams);
      res = k.sks->sub(res, ctxt_2, k.streams);
      CHECK(res.block_carries_are_empty());
      assert_same_ciphertext(k, res, tmp, "sub");
      k.panic_if_any_block_is_not_clean(res);
      clear = (clear - clear2) % modulus;
      CHECK_EQ(k.decrypt(res), clear);
    }
  }
}

// default_bitand / bitor / bitxor tests (tests_cases_unsigned.rs default_bit{and,or,xor}_test): an operand with carries (after an
// unchecked addition) is cleaned by the operation itself
static void integer_bitop(const TestParameters &param) {
  Keys &k = key_cache(param);
  const u64 modulus = unsigned_modulus(param.message_modulus, NB_CTXT);
  for (size_t t = 0; t < nb_tests_smaller_for_params(param); ++t) {
    const u64 clear1 = k.random() % modulus, clear2 = k.random() % modulus, clear3 = k.random() % modulus;
    const auto ctxt_1 = k.encrypt(clear1), ctxt_2 = k.encrypt(clear2), ctxt_3 = k.encrypt(clear3);
    const auto dirty = k.sks->unchecked_add(ctxt_1, ctxt_3, k.streams);  // carries not empty
    const u64 dirty_clear = (clear1 + clear3) % modulus;
    struct Case { BITOP_TYPE op; u64 want_clean, want_dirty; };
    for (const Case &c : {Case{BITAND, clear1 & clear2, dirty_clear & clear2}, Case{BITOR, clear1 | clear2, dirty_clear | clear2},
                          Case{BITXOR, clear1 ^ clear2, dirty_clear ^ clear2}}) {
      const auto r1 = k.sks->bitop(ctxt_1, ctxt_2, c.op, k.streams), r1b = k.sks->bitop(ctxt_1, ctxt_2, c.op, k.streams);
      CHECK(r1.block_carries_are_empty());
      assert_same_ciphertext(k, r1, r1b, "bitop");
      CHECK_EQ(k.decrypt(r1), c.want_clean);
      const auto r2 = k.sks->bitop(dirty, ctxt_2, c.op, k.streams);
      CHECK(r2.block_carries_are_empty());
      CHECK_EQ(k.decrypt(r2), c.want_dirty);
    }
  }
}

// default comparison tests (tests_unsigned/test_comparison.rs: test_default_function for eq, ne, gt, ge, lt, le; default_min / max):
// random pairs, an equal pair, pairs that differ by one
static void integer_comparison(const TestParameters &param) {
  Keys &k = key_cache(param);
  const u64 modulus = unsigned_modulus(param.message_modulus, NB_CTXT);
  const size_t rounds = g_toy ? 1 : std::max<size_t>(1, nb_tests_smaller_for_params(param) / 2);
  for (size_t t = 0; t < rounds + (g_toy ? 1 : 2); ++t) {
    u64 clear1 = k.random() % modulus, clear2 = t == rounds ? clear1 : t == rounds + 1 ? (clear1 + 1) % modulus : k.random() % modulus;
    const auto ctxt_1 = k.encrypt(clear1), ctxt_2 = k.encrypt(clear2);
    struct Case { COMPARISON_TYPE op; bool want; };
    for (const Case &c : {Case{EQ, clear1 == clear2}, Case{NE, clear1 != clear2}, Case{GT, clear1 > clear2}, Case{GE, clear1 >= clear2},
                          Case{LT, clear1 < clear2}, Case{LE, clear1 <= clear2}}) {
      const auto b = k.sks->comparison(ctxt_1, ctxt_2, c.op, k.streams), b2 = k.sks->comparison(ctxt_1, ctxt_2, c.op, k.streams);
      assert_same_ciphertext(k, b, b2, "comparison");
      CHECK(b.holds_boolean_value());
      CHECK_EQ(k.decrypt_bool(b), c.want);
    }
    if (g_toy && t > 0) continue;  // the emulation's short form: max / min once
    const auto mx = k.sks->comparison(ctxt_1, ctxt_2, MAX, k.streams), mn = k.sks->comparison(ctxt_1, ctxt_2, MIN, k.streams);
    CHECK(mx.block_carries_are_empty() && mn.block_carries_are_empty());
    CHECK_EQ(k.decrypt(mx), std::max(clear1, clear2));
    CHECK_EQ(k.decrypt(mn), std::min(clear1, clear2));
  }
}

// default scalar comparison tests (tests_unsigned/test_scalar_comparison.rs) and default_overflowing_sub_test
// (tests_unsigned/test_sub.rs): a scalar with fewer blocks than the ciphertext, zero, the value itself and its neighbours
static void integer_scalar_comparison_and_overflowing_sub(const TestParameters &param) {
  Keys &k = key_cache(param);
  const u64 modulus = unsigned_modulus(param.message_modulus, NB_CTXT);
  const u64 clear = k.random() % modulus;
  const auto ct = k.encrypt(clear);
  std::vector<u64> scalars = {clear, u64(3)};  // the emulation's short form: the value itself and a scalar of one block
  if (!g_toy) scalars.insert(scalars.end(), {u64(0), (clear + 1) % modulus, (clear + modulus - 1) % modulus, k.random() % modulus});
  for (u64 scalar : scalars) {
    struct Case { COMPARISON_TYPE op; bool want; };
    std::vector<Case> cases = {Case{EQ, clear == scalar}, Case{GT, clear > scalar}, Case{LE, clear <= scalar}};
    if (!g_toy) cases.insert(cases.end(), {Case{NE, clear != scalar}, Case{GE, clear >= scalar}, Case{LT, clear < scalar}});
    for (const Case &c : cases) {
      const auto b = k.sks->scalar_comparison(ct, scalar, c.op, k.streams);
      CHECK(b.holds_boolean_value());
      CHECK_EQ(k.decrypt_bool(b), c.want);
    }
    CHECK_EQ(k.decrypt(k.sks->scalar_comparison(ct, scalar, MAX, k.streams)), std::max(clear, scalar));
    if (!g_toy) CHECK_EQ(k.decrypt(k.sks->scalar_comparison(ct, scalar, MIN, k.streams)), std::min(clear, scalar));
  }
  for (size_t t = 0; t < (g_toy ? 2 : std::max<size_t>(2, nb_tests_smaller_for_params(param))); ++t) {
    const u64 clear_0 = k.random() % modulus, clear_1 = t == 0 ? clear_0 : k.random() % modulus;
    const auto ctxt_0 = k.encrypt(clear_0), ctxt_1 = k.encrypt(clear_1);
    auto [res, overflowed] = k.sks->unsigned_overflowing_sub(ctxt_0, ctxt_1, k.streams);
    auto [res2, overflowed2] = k.sks->unsigned_overflowing_sub(ctxt_0, ctxt_1, k.streams);
    assert_same_ciphertext(k, res, res2, "overflowing_sub");
    k.panic_if_any_block_is_not_clean(res);
    CHECK_EQ(k.decrypt(res), (clear_0 - clear_1) % modulus);
    CHECK_EQ(k.decrypt_bool(overflowed), clear_0 < clear_1);
    CHECK_EQ(k.decrypt_bool(overflowed2), clear_0 < clear_1);
  }
}

// default_if_then_else_test (tests_unsigned/test_cmux.rs): the condition from a comparison, both of its values
static void integer_if_then_else(const TestParameters &param) {
  Keys &k = key_cache(param);
  const u64 modulus = unsigned_modulus(param.message_modulus, NB_CTXT);
  for (size_t t = 0; t < std::max<size_t>(2, nb_tests_smaller_for_params(param)); ++t) {
    const u64 clear_0 = k.random() % modulus, clear_1 = k.random() % modulus;
    const auto ctxt_0 = k.encrypt(clear_0), ctxt_1 = k.encrypt(clear_1);
    const auto cond = k.sks->comparison(ctxt_0, ctxt_1, (t & 1) ? LT : GE, k.streams);
    const bool c = (t & 1) ? clear_0 < clear_1 : clear_0 >= clear_1;
    const auto r = k.sks->if_then_else(cond, ctxt_0, ctxt_1, k.streams), r2 = k.sks->if_then_else(cond, ctxt_0, ctxt_1, k.streams);
    assert_same_ciphertext(k, r, r2, "if_then_else");
    k.panic_if_any_block_is_not_clean(r);
    CHECK_EQ(k.decrypt(r), c ? clear_0 : clear_1);
  }
}

// default_scalar_left_shift_test / default_scalar_right_shift_test (tests_unsigned/test_scalar_shift.rs): every shift inside the
// width, then the overshift
static void integer_scalar_shift(const TestParameters &param) {
  Keys &k = key_cache(param);
  const u64 modulus = unsigned_modulus(param.message_modulus, NB_CTXT);
  uint32_t nbits = 0;
  for (u64 m = modulus; m > 1; m >>= 1) ++nbits;
  const u64 clear = k.random() % modulus;
  const auto ct = k.encrypt(clear);
  for (uint32_t shift = 0; shift <= nbits + 1; shift += (g_toy ? 3 : 1)) {
    const auto l = k.sks->scalar_shift(ct, shift, LEFT_SHIFT, k.streams), r = k.sks->scalar_shift(ct, shift, RIGHT_SHIFT, k.streams);
    CHECK(l.block_carries_are_empty() && r.block_carries_are_empty());
    CHECK_EQ(k.decrypt(l), shift >= nbits ? 0 : (clear << shift) % modulus);
    CHECK_EQ(k.decrypt(r), shift >= nbits ? 0 : clear >> shift);
  }
}

// One FheUint64 (32 blocks) addition / multiplication at a time the way the reference's host issues it — operands duplicated
// into a fresh CudaVec, scratch made, operation launched, scratch cleaned up, temporaries dropped, stream synchronised (radix/
// add.rs `add`, mul.rs `mul`) — timed on the host clock.  `latency` mode: one JSON line per (parameter set, operation), with the
// allocator's counters, so that TFHE_HIP_MALLOC_ASYNC=sync (hipMalloc / hipFree per CudaVec and per scratch array) and the
// default arena can be compared on the same box.
static void latency_mode(const TestParameters &param, size_t adds, size_t muls) {
  Keys &k = key_cache(param);
  const size_t blocks = 32;
  const auto a = k.encrypt_radix(k.random(), blocks), b = k.encrypt_radix(k.random(), blocks);
  auto timed = [&](const char *what, size_t reps, const std::function<CudaUnsignedRadixCiphertext()> &op) {
    { auto warm = op(); k.streams.synchronize(); }
    uint64_t s0[7], s1[7];
    hip_backend_allocator_stats(0, s0);
    std::vector<double> ms;
    for (size_t i = 0; i < reps; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      { auto r = op(); k.streams.synchronize(); }
      ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    hip_backend_allocator_stats(0, s1);
    std::sort(ms.begin(), ms.end());
    const char *mode = std::getenv("TFHE_HIP_MALLOC_ASYNC");
    std::printf("{\"what\": \"%s\", \"params\": \"%s\", \"blocks\": %zu, \"reps\": %zu, \"allocator\": \"%s\", \"ms_median\": %.3f, \"ms_min\": %.3f, "
                "\"ms_max\": %.3f, \"arena_allocations_per_op\": %.1f, \"arena_runtime_allocations\": %llu, \"arena_cached_bytes\": %llu}\n",
                what, param.name, blocks, reps, mode ? mode : "arena", ms[ms.size() / 2], ms.front(), ms.back(), double(s1[0] - s0[0]) / reps,
                (unsigned long long)(s1[2] - s0[2]), (unsigned long long)s1[6]);
    std::fflush(stdout);
  };
  timed("fheuint64_add", adds, [&] { return k.sks->add(a, b, k.streams); });
  timed("fheuint64_mul", muls, [&] { return k.sks->mul(a, b, k.streams); });
}

struct Test {
  std::string name;
  std::function<void()> body;
};
int main(int argc, char **argv) {
  if (argc >= 2 && !std::strcmp(argv[1], "latency")) {
    if (!is_cuda_available()) return 2;
    const bool toy = argc > 2 && !std::strcmp(argv[2], "toy");
    latency_mode(toy ? TOY_MESSAGE_2_CARRY_2 : PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128, toy ? 3 : 30, toy ? 1 : 8);
    latency_mode(toy ? TOY_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2 : TEST_PARAM_GPU_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128, toy ? 3 : 30,
                 toy ? 1 : 8);
    return 0;
  }
  if (argc < 2 || (std::strcmp(argv[1], "toy") && std::strcmp(argv[1], "reference"))) {
    std::fprintf(stderr, "usage: %s <toy|reference> [test-name-substring]\n", argv[0]);
    return 2;
  }
  g_toy = !std::strcmp(argv[1], "toy");
  const bool reference_sets = !g_toy;
  if (std::getenv("REFERENCE_TESTS_SHORT_LOOPS")) g_toy = true;  // (the emulation takes 14 s per bootstrap at these sizes)
  const char *filter = argc > 2 ? argv[2] : "";
  if (!is_cuda_available()) {
    std::fprintf(stderr, "no device visible: the backend has no CPU path\n");
    return 2;
  }
  std::vector<Test> tests;
  auto all = [&](const TestParameters &p) {  // create_gpu_parameterized_test! (tests_unsigned/mod.rs)
    tests.push_back({std::string("test_gpu_integer_unchecked_add_") + p.name, [&p] { integer_unchecked_add(p); }});
    tests.push_back({std::string("test_gpu_integer_add_") + p.name, [&p] { integer_add(p); }});
    tests.push_back({std::string("test_gpu_integer_default_overflowing_add_") + p.name, [&p] { integer_default_overflowing_add(p); }});
    tests.push_back({std::string("test_gpu_integer_mul_") + p.name, [&p] { integer_mul(p); }});
    tests.push_back({std::string("test_gpu_integer_sub_") + p.name, [&p] { integer_sub(p); }});
    tests.push_back({std::string("test_gpu_integer_bitop_") + p.name, [&p] { integer_bitop(p); }});
    tests.push_back({std::string("test_gpu_integer_comparison_") + p.name, [&p] { integer_comparison(p); }});
    tests.push_back({std::string("test_gpu_integer_if_then_else_") + p.name, [&p] { integer_if_then_else(p); }});
    tests.push_back({std::string("test_gpu_integer_scalar_comparison_and_overflowing_sub_") + p.name,
                     [&p] { integer_scalar_comparison_and_overflowing_sub(p); }});
    tests.push_back({std::string("test_gpu_integer_scalar_shift_") + p.name, [&p] { integer_scalar_shift(p); }});
    if (get_number_of_gpus() > 1) tests.push_back({std::string("test_gpu_multi_device_integer_add_") + p.name, [&p] { multi_device_integer_add(p); }});
  };
  if (!reference_sets) {
    all(TOY_MESSAGE_2_CARRY_2);
    all(TOY_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2);
  } else {
    all(PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128);
    all(TEST_PARAM_GPU_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128);
  }
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
