// integer_gpu.hpp — compiled host side of the radix layer: a C++17 mirror of the part of `tfhe::integer::gpu` the backend
// wires, over the C ABI only, on top of core_crypto_gpu.hpp.  Same names and argument meaning as the
// Rust module; a Rust panic is a `gpu::Panic`.  tfhe_rs_amd/integer_gpu.py is the same mirror for the Python harness (there a
// ciphertext object may hold a batch of integers; here, as in the reference, one integer).
//
//   CudaRadixCiphertextInfo / CudaRadixCiphertext / CudaUnsignedRadixCiphertext   tfhe/src/integer/gpu/ciphertext/{info.rs,mod.rs}
//   CudaBootstrappingKey / CudaServerKey                                          tfhe/src/integer/gpu/server_key/mod.rs:26-160
//   unchecked_add(_assign), add(_assign), unsigned_overflowing_add                tfhe/src/integer/gpu/server_key/radix/add.rs
//   propagate_single_carry_assign                                                 tfhe/src/integer/gpu/server_key/radix/mod.rs
//   mul(_assign)  (boolean operands: `holds_boolean_value`)                       tfhe/src/integer/gpu/server_key/radix/mul.rs
//   apply_lookup_table                                                            tfhe/src/integer/gpu/server_key/radix/mod.rs
//   the FFI wrappers they call                                                    tfhe/src/integer/gpu/mod.rs (cuda_backend_*), gpu/ffi.rs:2194-2237
#pragma once

#include <optional>
#include <variant>

#include "core_crypto_gpu.hpp"

namespace tfhe::integer::gpu {

using tfhe::core_crypto::gpu::CudaLweBootstrapKey;
using tfhe::core_crypto::gpu::CudaLweKeyswitchKey;
using tfhe::core_crypto::gpu::CudaLweMultiBitBootstrapKey;
using tfhe::core_crypto::gpu::CudaStreams;
using tfhe::core_crypto::gpu::CudaVec;
using tfhe::core_crypto::gpu::Panic;
namespace detail = tfhe::core_crypto::gpu::detail;

enum class PBSType : uint32_t { MultiBit = 0, Classical = 1 };                  // pbs/pbs_enums.h:4
enum class OutputFlag : uint32_t { None = 0, Overflow = 1, Carry = 2 };         // integer/integer.h:39

// ciphertext/info.rs: what the server tracks per block
struct CudaBlockInfo {
  uint64_t degree = 0, message_modulus = 0, carry_modulus = 0, noise_level = 0;
};

// ciphertext/mod.rs: device blocks [block][lwe_size] u64, least significant block first, plus the per-block info
class CudaUnsignedRadixCiphertext {
 public:
  CudaVec<uint64_t> d_blocks;
  size_t lwe_dimension = 0;
  std::vector<uint64_t> degrees, noise_levels;  // the info.blocks of the reference, in the layout the FFI struct points at
  uint64_t message_modulus = 0, carry_modulus = 0;

  // CudaUnsignedRadixCiphertext::from_radix_ciphertext: host blocks produced by the client key (fresh: degree = msg - 1, noise 1)
  static CudaUnsignedRadixCiphertext from_radix_ciphertext(const std::vector<uint64_t> &h_blocks, size_t lwe_dimension, uint64_t message_modulus,
                                                           uint64_t carry_modulus, uint64_t degree, const CudaStreams &streams) {
    detail::assert_true(h_blocks.size() % (lwe_dimension + 1) == 0, "the container is not a whole number of blocks");
    CudaUnsignedRadixCiphertext c;
    const size_t n = h_blocks.size() / (lwe_dimension + 1);
    c.d_blocks = CudaVec<uint64_t>::new_async(h_blocks.size(), streams, 0);
    c.d_blocks.copy_from_cpu_async(h_blocks, streams, 0);
    streams.synchronize();
    c.lwe_dimension = lwe_dimension;
    c.degrees.assign(n, degree);
    c.noise_levels.assign(n, degree ? 1 : 0);
    c.message_modulus = message_modulus;
    c.carry_modulus = carry_modulus;
    return c;
  }
  // CudaServerKey::create_trivial_zero_radix
  static CudaUnsignedRadixCiphertext zero(size_t num_blocks, size_t lwe_dimension, uint64_t message_modulus, uint64_t carry_modulus,
                                          const CudaStreams &streams) {
    return from_radix_ciphertext(std::vector<uint64_t>(num_blocks * (lwe_dimension + 1), 0), lwe_dimension, message_modulus, carry_modulus, 0,
                                 streams);
  }
  CudaUnsignedRadixCiphertext duplicate(const CudaStreams &streams) const {  // ciphertext/mod.rs duplicate
    CudaUnsignedRadixCiphertext c;
    c.d_blocks = CudaVec<uint64_t>::new_async(d_blocks.len, streams, 0);
    c.d_blocks.copy_from_gpu_async(d_blocks, streams, 0);
    streams.synchronize();
    c.lwe_dimension = lwe_dimension;
    c.degrees = degrees;
    c.noise_levels = noise_levels;
    c.message_modulus = message_modulus;
    c.carry_modulus = carry_modulus;
    return c;
  }
  std::vector<uint64_t> to_radix_ciphertext(const CudaStreams &streams) const { return d_blocks.to_cpu(streams, 0); }
  size_t num_blocks() const { return degrees.size(); }
  bool block_carries_are_empty() const {  // ciphertext/info.rs
    for (uint64_t d : degrees)
      if (d >= message_modulus) return false;
    return true;
  }
  bool holds_boolean_value() const {  // ciphertext/mod.rs: first block of degree <= 1, the others of degree 0
    if (degrees.empty() || degrees[0] > 1) return false;
    for (size_t i = 1; i < degrees.size(); ++i)
      if (degrees[i] != 0) return false;
    return true;
  }
  CudaRadixCiphertextFFI ffi() {  // integer/gpu/mod.rs prepare_cuda_radix_ffi
    return CudaRadixCiphertextFFI{d_blocks.as_mut_c_ptr(0), degrees.data(), noise_levels.data(), (uint32_t)num_blocks(), (uint32_t)num_blocks(),
                                  (uint32_t)lwe_dimension};
  }
  CudaRadixCiphertextFFI ffi() const { return const_cast<CudaUnsignedRadixCiphertext *>(this)->ffi(); }
};
using CudaBooleanBlock = CudaUnsignedRadixCiphertext;  // one block of degree <= 1 (ciphertext/boolean_value.rs)

// server_key/mod.rs:26-35
using CudaBootstrappingKey = std::variant<CudaLweBootstrapKey, CudaLweMultiBitBootstrapKey>;

// server_key/mod.rs:37-160
class CudaServerKey {
 public:
  CudaLweKeyswitchKey<uint64_t> key_switching_key;
  CudaBootstrappingKey bootstrapping_key;
  uint64_t message_modulus, carry_modulus;

  CudaServerKey(CudaLweKeyswitchKey<uint64_t> ksk, CudaBootstrappingKey bsk, uint64_t message_modulus_, uint64_t carry_modulus_)
      : key_switching_key(std::move(ksk)), bootstrapping_key(std::move(bsk)), message_modulus(message_modulus_), carry_modulus(carry_modulus_) {
    const CudaLweBootstrapKeyParamsFFI b = bsk_params();
    detail::assert_eq(key_switching_key.output_key_lwe_dimension(), (size_t)b.input_lwe_dimension, "keyswitch and bootstrap keys do not chain");
    detail::assert_eq(key_switching_key.input_key_lwe_dimension(), (size_t)b.big_lwe_dimension, "keyswitch and bootstrap keys do not chain");
  }

  // ---- the operations (radix/add.rs, radix/mul.rs, radix/mod.rs)
  void unchecked_add_assign(CudaUnsignedRadixCiphertext &ct_left, const CudaUnsignedRadixCiphertext &ct_right, const CudaStreams &streams) const {
    detail::assert_eq(ct_left.lwe_dimension, ct_right.lwe_dimension, "Mismatched lwe dimension between ct_left and ct_right");
    detail::assert_eq(ct_left.num_blocks(), ct_right.num_blocks(), "Mismatched number of blocks between ct_left and ct_right");
    CudaRadixCiphertextFFI l = ct_left.ffi(), r = ct_right.ffi();
    cuda_add_lwe_ciphertext_vector_inplace_64(streams.ptr[0], streams.gpu_indexes[0].get(), &l, &r);  // updates degrees / noise levels
  }
  CudaUnsignedRadixCiphertext unchecked_add(const CudaUnsignedRadixCiphertext &ct_left, const CudaUnsignedRadixCiphertext &ct_right,
                                            const CudaStreams &streams) const {
    CudaUnsignedRadixCiphertext result = ct_left.duplicate(streams);
    unchecked_add_assign(result, ct_right, streams);
    return result;
  }
  // full_propagate_assign on an operand that is not clean (radix/mod.rs): one carry propagation
  void propagate_single_carry_assign(CudaUnsignedRadixCiphertext &ct, const CudaStreams &streams, const CudaBooleanBlock *input_carry = nullptr,
                                     OutputFlag requested_flag = OutputFlag::None, CudaBooleanBlock *carry_out = nullptr) const {
    Ffi f(*this, streams);
    CudaUnsignedRadixCiphertext zero_in = CudaUnsignedRadixCiphertext::zero(1, ct.lwe_dimension, message_modulus, carry_modulus, streams),
                                zero_out = CudaUnsignedRadixCiphertext::zero(1, ct.lwe_dimension, message_modulus, carry_modulus, streams);
    CudaRadixCiphertextFFI c = ct.ffi(), cin = (input_carry ? *input_carry : zero_in).ffi(), cout = (carry_out ? *carry_out : zero_out).ffi();
    int8_t *mem = nullptr;
    scratch_cuda_propagate_single_carry_64_inplace_async(f.streams, &mem, bsk_params(), ksk_params(), (uint32_t)ct.num_blocks(), (uint32_t)message_modulus,
                                                         (uint32_t)carry_modulus, (uint32_t)requested_flag, true, noise_reduction());
    cuda_propagate_single_carry_64_inplace_async(f.streams, &c, &cout, &cin, mem, f.bsks.data(), f.ksks.data(), (uint32_t)requested_flag,
                                                 input_carry ? 1u : 0u);
    cleanup_cuda_propagate_single_carry_64_inplace(f.streams, &mem);
    if (carry_out) carry_out->degrees.assign(1, 1);
  }
  // add.rs `add_assign`: operands with non-empty carries are propagated first, then block additions + one propagation in ONE call
  void add_assign(CudaUnsignedRadixCiphertext &ct_left, const CudaUnsignedRadixCiphertext &ct_right, const CudaStreams &streams,
                  OutputFlag requested_flag = OutputFlag::None, CudaBooleanBlock *flag_out = nullptr) const {
    detail::assert_eq(ct_left.num_blocks(), ct_right.num_blocks(), "Mismatched number of blocks between ct_left and ct_right");
    std::optional<CudaUnsignedRadixCiphertext> clean_right;
    if (!ct_left.block_carries_are_empty()) propagate_single_carry_assign(ct_left, streams);
    if (!ct_right.block_carries_are_empty()) {
      clean_right = ct_right.duplicate(streams);
      propagate_single_carry_assign(*clean_right, streams);
    }
    const CudaUnsignedRadixCiphertext &rhs = clean_right ? *clean_right : ct_right;
    Ffi f(*this, streams);
    CudaUnsignedRadixCiphertext zero_in = CudaUnsignedRadixCiphertext::zero(1, ct_left.lwe_dimension, message_modulus, carry_modulus, streams),
                                zero_out = CudaUnsignedRadixCiphertext::zero(1, ct_left.lwe_dimension, message_modulus, carry_modulus, streams);
    CudaRadixCiphertextFFI l = ct_left.ffi(), r = rhs.ffi(), cin = zero_in.ffi(), cout = (flag_out ? *flag_out : zero_out).ffi();
    int8_t *mem = nullptr;
    scratch_cuda_add_and_propagate_single_carry_64_inplace_async(f.streams, &mem, bsk_params(), ksk_params(), (uint32_t)ct_left.num_blocks(),
                                                                 (uint32_t)message_modulus, (uint32_t)carry_modulus, (uint32_t)requested_flag, true,
                                                                 noise_reduction());
    cuda_add_and_propagate_single_carry_64_inplace_async(f.streams, &l, &r, &cout, &cin, mem, f.bsks.data(), f.ksks.data(), (uint32_t)requested_flag, 0);
    cleanup_cuda_add_and_propagate_single_carry_64_inplace(f.streams, &mem);
    if (flag_out) flag_out->degrees.assign(1, 1);
  }
  CudaUnsignedRadixCiphertext add(const CudaUnsignedRadixCiphertext &ct_left, const CudaUnsignedRadixCiphertext &ct_right,
                                  const CudaStreams &streams) const {
    CudaUnsignedRadixCiphertext result = ct_left.duplicate(streams);
    add_assign(result, ct_right, streams);
    return result;
  }
  // add.rs `unsigned_overflowing_add`: the sum and the carry leaving the last block (OutputFlag::Carry)
  std::pair<CudaUnsignedRadixCiphertext, CudaBooleanBlock> unsigned_overflowing_add(const CudaUnsignedRadixCiphertext &ct_left,
                                                                                    const CudaUnsignedRadixCiphertext &ct_right,
                                                                                    const CudaStreams &streams) const {
    CudaUnsignedRadixCiphertext result = ct_left.duplicate(streams);
    CudaBooleanBlock overflowed = CudaUnsignedRadixCiphertext::zero(1, ct_left.lwe_dimension, message_modulus, carry_modulus, streams);
    add_assign(result, ct_right, streams, OutputFlag::Carry, &overflowed);
    return {std::move(result), std::move(overflowed)};
  }
  // mul.rs `mul_assign`: clean operands; an operand that holds a boolean selects (is_boolean_left / is_boolean_right)
  void mul_assign(CudaUnsignedRadixCiphertext &ct_left, const CudaUnsignedRadixCiphertext &ct_right, const CudaStreams &streams) const {
    detail::assert_eq(ct_left.num_blocks(), ct_right.num_blocks(), "Mismatched number of blocks between ct_left and ct_right");
    std::optional<CudaUnsignedRadixCiphertext> clean_right;
    if (!ct_left.block_carries_are_empty()) propagate_single_carry_assign(ct_left, streams);
    if (!ct_right.block_carries_are_empty()) {
      clean_right = ct_right.duplicate(streams);
      propagate_single_carry_assign(*clean_right, streams);
    }
    const CudaUnsignedRadixCiphertext &rhs = clean_right ? *clean_right : ct_right;
    const bool is_boolean_left = ct_left.holds_boolean_value(), is_boolean_right = !is_boolean_left && rhs.holds_boolean_value();
    Ffi f(*this, streams);
    const CudaLweBootstrapKeyParamsFFI b = bsk_params();
    CudaRadixCiphertextFFI l = ct_left.ffi(), r = rhs.ffi();
    int8_t *mem = nullptr;
    scratch_cuda_integer_mult_inplace_64_async(f.streams, &mem, is_boolean_left, is_boolean_right, (uint32_t)message_modulus, (uint32_t)carry_modulus, b,
                                               ksk_params(), (uint32_t)ct_left.num_blocks(), true, noise_reduction());
    cuda_integer_mult_inplace_64_async(f.streams, &l, is_boolean_left, &r, is_boolean_right, f.bsks.data(), f.ksks.data(), mem, b.polynomial_size,
                                       (uint32_t)ct_left.num_blocks());
    cleanup_cuda_integer_mult_inplace_64(f.streams, &mem);
  }
  CudaUnsignedRadixCiphertext mul(const CudaUnsignedRadixCiphertext &ct_left, const CudaUnsignedRadixCiphertext &ct_right,
                                  const CudaStreams &streams) const {
    CudaUnsignedRadixCiphertext result = ct_left.duplicate(streams);
    mul_assign(result, ct_right, streams);
    return result;
  }
  // ---- round 6: sub.rs, bitwise_op.rs, comparison.rs, cmux.rs, scalar_shift.rs — the default forms: an operand whose carries
  // are not empty is propagated first
  const CudaUnsignedRadixCiphertext &cleaned(const CudaUnsignedRadixCiphertext &ct, std::optional<CudaUnsignedRadixCiphertext> &hold,
                                             const CudaStreams &streams) const {
    if (ct.block_carries_are_empty()) return ct;
    hold = ct.duplicate(streams);
    propagate_single_carry_assign(*hold, streams);
    return *hold;
  }
  void sub_assign(CudaUnsignedRadixCiphertext &ct_left, const CudaUnsignedRadixCiphertext &ct_right, const CudaStreams &streams) const {
    detail::assert_eq(ct_left.num_blocks(), ct_right.num_blocks(), "Mismatched number of blocks between ct_left and ct_right");
    if (!ct_left.block_carries_are_empty()) propagate_single_carry_assign(ct_left, streams);
    std::optional<CudaUnsignedRadixCiphertext> hold;
    const CudaUnsignedRadixCiphertext &rhs = cleaned(ct_right, hold, streams);
    Ffi f(*this, streams);
    CudaUnsignedRadixCiphertext zero_in = CudaUnsignedRadixCiphertext::zero(1, ct_left.lwe_dimension, message_modulus, carry_modulus, streams),
                                zero_out = CudaUnsignedRadixCiphertext::zero(1, ct_left.lwe_dimension, message_modulus, carry_modulus, streams);
    CudaRadixCiphertextFFI l = ct_left.ffi(), r = rhs.ffi(), cin = zero_in.ffi(), cout = zero_out.ffi();
    int8_t *mem = nullptr;
    scratch_cuda_sub_and_propagate_single_carry_64_inplace_async(f.streams, &mem, bsk_params(), ksk_params(), (uint32_t)ct_left.num_blocks(),
                                                                 (uint32_t)message_modulus, (uint32_t)carry_modulus, 0, true, noise_reduction());
    cuda_sub_and_propagate_single_carry_64_inplace_async(f.streams, &l, &r, &cout, &cin, mem, f.bsks.data(), f.ksks.data(), 0, 0);
    cleanup_cuda_sub_and_propagate_single_carry_64_inplace(f.streams, &mem);
  }
  CudaUnsignedRadixCiphertext sub(const CudaUnsignedRadixCiphertext &ct_left, const CudaUnsignedRadixCiphertext &ct_right,
                                  const CudaStreams &streams) const {
    CudaUnsignedRadixCiphertext result = ct_left.duplicate(streams);
    sub_assign(result, ct_right, streams);
    return result;
  }
  CudaUnsignedRadixCiphertext bitop(const CudaUnsignedRadixCiphertext &ct_left, const CudaUnsignedRadixCiphertext &ct_right, BITOP_TYPE op,
                                    const CudaStreams &streams) const {
    detail::assert_eq(ct_left.num_blocks(), ct_right.num_blocks(), "Mismatched number of blocks between ct_left and ct_right");
    CudaUnsignedRadixCiphertext result = ct_left.duplicate(streams);
    if (!result.block_carries_are_empty()) propagate_single_carry_assign(result, streams);
    std::optional<CudaUnsignedRadixCiphertext> hold;
    const CudaUnsignedRadixCiphertext &rhs = cleaned(ct_right, hold, streams);
    Ffi f(*this, streams);
    CudaRadixCiphertextFFI l = result.ffi(), r = rhs.ffi();
    int8_t *mem = nullptr;
    scratch_cuda_integer_bitop_inplace_64_async(f.streams, &mem, bsk_params(), ksk_params(), (uint32_t)result.num_blocks(), (uint32_t)message_modulus,
                                                (uint32_t)carry_modulus, op, true, noise_reduction());
    cuda_integer_bitop_inplace_64_async(f.streams, &l, &r, mem, f.bsks.data(), f.ksks.data());
    cleanup_cuda_integer_bitop_inplace_64(f.streams, &mem);
    return result;
  }
  CudaUnsignedRadixCiphertext bitand_(const CudaUnsignedRadixCiphertext &a, const CudaUnsignedRadixCiphertext &b, const CudaStreams &s) const { return bitop(a, b, BITAND, s); }
  CudaUnsignedRadixCiphertext bitor_(const CudaUnsignedRadixCiphertext &a, const CudaUnsignedRadixCiphertext &b, const CudaStreams &s) const { return bitop(a, b, BITOR, s); }
  CudaUnsignedRadixCiphertext bitxor_(const CudaUnsignedRadixCiphertext &a, const CudaUnsignedRadixCiphertext &b, const CudaStreams &s) const { return bitop(a, b, BITXOR, s); }
  // comparison.rs: eq ... le return a boolean block, max / min an integer
  CudaUnsignedRadixCiphertext comparison(const CudaUnsignedRadixCiphertext &ct_left, const CudaUnsignedRadixCiphertext &ct_right, COMPARISON_TYPE op,
                                         const CudaStreams &streams) const {
    detail::assert_eq(ct_left.num_blocks(), ct_right.num_blocks(), "Mismatched number of blocks between ct_left and ct_right");
    std::optional<CudaUnsignedRadixCiphertext> hold_l, hold_r;
    const CudaUnsignedRadixCiphertext &lhs = cleaned(ct_left, hold_l, streams), &rhs = cleaned(ct_right, hold_r, streams);
    const bool select = op == MAX || op == MIN;
    CudaUnsignedRadixCiphertext result =
        CudaUnsignedRadixCiphertext::zero(select ? lhs.num_blocks() : 1, lhs.lwe_dimension, message_modulus, carry_modulus, streams);
    Ffi f(*this, streams);
    CudaRadixCiphertextFFI o = result.ffi(), l = lhs.ffi(), r = rhs.ffi();
    int8_t *mem = nullptr;
    scratch_cuda_integer_comparison_64_async(f.streams, &mem, bsk_params(), ksk_params(), (uint32_t)lhs.num_blocks(), (uint32_t)message_modulus,
                                             (uint32_t)carry_modulus, op, false, true, noise_reduction());
    cuda_integer_comparison_64_async(f.streams, &o, &l, &r, mem, f.bsks.data(), f.ksks.data());
    cleanup_cuda_integer_comparison_64(f.streams, &mem);
    return result;
  }
  // scalar_comparison.rs: the scalar's clear blocks (message_modulus per block, least significant first, stopping at the last
  // non-zero one and truncated to the ciphertext's length — the caller has dealt with a scalar that does not fit)
  CudaUnsignedRadixCiphertext scalar_comparison(const CudaUnsignedRadixCiphertext &ct, uint64_t scalar, COMPARISON_TYPE op,
                                                const CudaStreams &streams) const {
    std::optional<CudaUnsignedRadixCiphertext> hold;
    const CudaUnsignedRadixCiphertext &lhs = cleaned(ct, hold, streams);
    std::vector<uint64_t> blocks;
    for (uint64_t v = scalar; v != 0 && blocks.size() < lhs.num_blocks(); v /= message_modulus) blocks.push_back(v % message_modulus);
    core_crypto::gpu::CudaVec<uint64_t> d_blocks = core_crypto::gpu::CudaVec<uint64_t>::from_cpu_async(
        blocks.empty() ? std::vector<uint64_t>{0} : blocks, streams, 0);
    const bool select = op == MAX || op == MIN;
    CudaUnsignedRadixCiphertext result =
        CudaUnsignedRadixCiphertext::zero(select ? lhs.num_blocks() : 1, lhs.lwe_dimension, message_modulus, carry_modulus, streams);
    Ffi f(*this, streams);
    CudaRadixCiphertextFFI o = result.ffi(), l = lhs.ffi();
    int8_t *mem = nullptr;
    scratch_cuda_integer_scalar_comparison_64_async(f.streams, &mem, bsk_params(), ksk_params(), (uint32_t)lhs.num_blocks(),
                                                    (uint32_t)message_modulus, (uint32_t)carry_modulus, op, false, true, noise_reduction());
    cuda_integer_scalar_comparison_64_async(f.streams, &o, &l, d_blocks.ptr[0], blocks.data(), mem, f.bsks.data(), f.ksks.data(),
                                            (uint32_t)blocks.size());
    cleanup_cuda_integer_scalar_comparison_64(f.streams, &mem);
    streams.synchronize();
    return result;
  }
  // sub.rs unsigned_overflowing_sub: the difference and the borrow
  std::pair<CudaUnsignedRadixCiphertext, CudaBooleanBlock> unsigned_overflowing_sub(const CudaUnsignedRadixCiphertext &ct_left,
                                                                                    const CudaUnsignedRadixCiphertext &ct_right,
                                                                                    const CudaStreams &streams) const {
    detail::assert_eq(ct_left.num_blocks(), ct_right.num_blocks(), "Mismatched number of blocks between ct_left and ct_right");
    CudaUnsignedRadixCiphertext result = ct_left.duplicate(streams);
    if (!result.block_carries_are_empty()) propagate_single_carry_assign(result, streams);
    std::optional<CudaUnsignedRadixCiphertext> hold;
    const CudaUnsignedRadixCiphertext &rhs = cleaned(ct_right, hold, streams);
    CudaBooleanBlock overflowed = CudaUnsignedRadixCiphertext::zero(1, ct_left.lwe_dimension, message_modulus, carry_modulus, streams);
    CudaUnsignedRadixCiphertext zero_in = CudaUnsignedRadixCiphertext::zero(1, ct_left.lwe_dimension, message_modulus, carry_modulus, streams);
    Ffi f(*this, streams);
    CudaRadixCiphertextFFI l = result.ffi(), r = rhs.ffi(), o = overflowed.ffi(), b = zero_in.ffi();
    int8_t *mem = nullptr;
    scratch_cuda_integer_overflowing_sub_64_inplace_async(f.streams, &mem, bsk_params(), ksk_params(), (uint32_t)result.num_blocks(),
                                                          (uint32_t)message_modulus, (uint32_t)carry_modulus, 1, true, noise_reduction());
    cuda_integer_overflowing_sub_64_inplace_async(f.streams, &l, &r, &o, &b, mem, f.bsks.data(), f.ksks.data(), 1, 0);
    cleanup_cuda_integer_overflowing_sub_64_inplace(f.streams, &mem);
    overflowed.degrees.assign(1, 1);
    return {std::move(result), std::move(overflowed)};
  }
  // cmux.rs if_then_else
  CudaUnsignedRadixCiphertext if_then_else(const CudaBooleanBlock &condition, const CudaUnsignedRadixCiphertext &true_ct,
                                           const CudaUnsignedRadixCiphertext &false_ct, const CudaStreams &streams) const {
    detail::assert_eq(true_ct.num_blocks(), false_ct.num_blocks(), "Mismatched number of blocks between true_ct and false_ct");
    std::optional<CudaUnsignedRadixCiphertext> hold_t, hold_f;
    const CudaUnsignedRadixCiphertext &t = cleaned(true_ct, hold_t, streams), &e = cleaned(false_ct, hold_f, streams);
    CudaUnsignedRadixCiphertext result = CudaUnsignedRadixCiphertext::zero(t.num_blocks(), t.lwe_dimension, message_modulus, carry_modulus, streams);
    Ffi f(*this, streams);
    CudaRadixCiphertextFFI o = result.ffi(), c = condition.ffi(), tt = t.ffi(), ee = e.ffi();
    int8_t *mem = nullptr;
    scratch_cuda_cmux_64_async(f.streams, &mem, bsk_params(), ksk_params(), (uint32_t)t.num_blocks(), (uint32_t)message_modulus, (uint32_t)carry_modulus,
                               true, noise_reduction());
    cuda_cmux_64_async(f.streams, &o, &c, &tt, &ee, mem, f.bsks.data(), f.ksks.data());
    cleanup_cuda_cmux_64(f.streams, &mem);
    return result;
  }
  // scalar_shift.rs scalar_left_shift / scalar_right_shift (unsigned: logical)
  CudaUnsignedRadixCiphertext scalar_shift(const CudaUnsignedRadixCiphertext &ct, uint32_t shift, SHIFT_OR_ROTATE_TYPE direction,
                                           const CudaStreams &streams) const {
    CudaUnsignedRadixCiphertext result = ct.duplicate(streams);
    if (!result.block_carries_are_empty()) propagate_single_carry_assign(result, streams);
    Ffi f(*this, streams);
    CudaRadixCiphertextFFI c = result.ffi();
    int8_t *mem = nullptr;
    scratch_cuda_logical_scalar_shift_64_inplace_async(f.streams, &mem, bsk_params(), ksk_params(), (uint32_t)result.num_blocks(),
                                                       (uint32_t)message_modulus, (uint32_t)carry_modulus, direction, true, noise_reduction());
    cuda_logical_scalar_shift_64_inplace_async(f.streams, &c, shift, mem, f.bsks.data(), f.ksks.data());
    cleanup_cuda_logical_scalar_shift_64_inplace(f.streams, &mem);
    return result;
  }

  // radix/mod.rs apply_lookup_table: `lut` is the (k+1)*N accumulator of a shortint LookupTable, `degree` its maximum output
  CudaUnsignedRadixCiphertext apply_lookup_table(const CudaUnsignedRadixCiphertext &input, const std::vector<uint64_t> &lut, uint64_t degree,
                                                 const CudaStreams &streams) const {
    Ffi f(*this, streams);
    CudaUnsignedRadixCiphertext output = CudaUnsignedRadixCiphertext::zero(input.num_blocks(), input.lwe_dimension, message_modulus, carry_modulus, streams);
    CudaRadixCiphertextFFI o = output.ffi(), i = input.ffi();
    int8_t *mem = nullptr;
    scratch_cuda_apply_univariate_lut_64_async(f.streams, &mem, lut.data(), bsk_params(), ksk_params(), (uint32_t)input.num_blocks(), (uint32_t)message_modulus,
                                               (uint32_t)carry_modulus, degree, true, noise_reduction());
    cuda_apply_univariate_lut_64_async(f.streams, &o, &i, mem, f.ksks.data(), f.bsks.data());
    cleanup_cuda_apply_univariate_lut_64(f.streams, &mem);
    return output;
  }

  CudaLweBootstrapKeyParamsFFI bsk_params() const {  // integer/gpu/mod.rs prepare_cuda_lwe_bsk_ffi
    if (const auto *c = std::get_if<CudaLweBootstrapKey>(&bootstrapping_key))
      return {(uint32_t)c->input_lwe_dimension(), (uint32_t)c->glwe_dimension(), (uint32_t)c->polynomial_size(), (uint32_t)c->decomp_base_log(),
              (uint32_t)c->decomp_level_count(), (uint32_t)c->output_lwe_dimension(), (uint32_t)PBSType::Classical, 0};
    const auto &m = std::get<CudaLweMultiBitBootstrapKey>(bootstrapping_key);
    return {(uint32_t)m.input_lwe_dimension(), (uint32_t)m.glwe_dimension(), (uint32_t)m.polynomial_size(), (uint32_t)m.decomp_base_log(),
            (uint32_t)m.decomp_level_count(), (uint32_t)m.output_lwe_dimension(), (uint32_t)PBSType::MultiBit, (uint32_t)m.grouping_factor()};
  }
  CudaLweKeyswitchKeyParamsFFI ksk_params() const {
    return {(uint32_t)key_switching_key.input_key_lwe_dimension(), (uint32_t)key_switching_key.output_key_lwe_dimension(),
            (uint32_t)key_switching_key.decomposition_base_log(), (uint32_t)key_switching_key.decomposition_level_count()};
  }

 private:
  PBS_MS_REDUCTION_T noise_reduction() const {
    const auto *c = std::get_if<CudaLweBootstrapKey>(&bootstrapping_key);
    return c && c->ms_noise_reduction_configuration ? PBS_MS_REDUCTION_T::CENTERED : PBS_MS_REDUCTION_T::NO_REDUCTION;
  }
  // CudaStreamsFFI + the per-GPU key pointer arrays the wrappers of integer/gpu/mod.rs pass (one replica per GPU of the set)
  struct Ffi {
    std::vector<uint32_t> gpu_indexes;
    std::vector<void *> ksks, bsks;
    CudaStreamsFFI streams;
    Ffi(const CudaServerKey &k, const CudaStreams &s) {
      for (const auto &g : s.gpu_indexes) gpu_indexes.push_back(g.get());
      const bool classic = std::holds_alternative<CudaLweBootstrapKey>(k.bootstrapping_key);
      for (size_t i = 0; i < s.len(); ++i) {
        detail::assert_true(i < k.key_switching_key.d_vec.ptr.size(), "server key has fewer GPU replicas than the stream set has streams");
        ksks.push_back(k.key_switching_key.d_vec.ptr[i]);
        bsks.push_back(classic ? std::get<CudaLweBootstrapKey>(k.bootstrapping_key).d_vec.ptr[i]
                               : std::get<CudaLweMultiBitBootstrapKey>(k.bootstrapping_key).d_vec.ptr[i]);
      }
      streams = CudaStreamsFFI{s.ptr.data(), gpu_indexes.data(), (uint32_t)s.len()};
    }
  };
};

}  // namespace tfhe::integer::gpu
