/*
 * tfhe_hip_backend.h — C ABI of the MI355X (gfx950) TFHE programmable-bootstrapping backend.
 *
 * This is the drop-in boundary for the PBS hot path of tfhe-rs (keyswitch -> modulus switch
 * -> blind rotation -> sample extract).  Every prototype below is EXACTLY the prototype the
 * reference's Rust FFI binds for this path (bindgen over
 * backends/tfhe-cuda-backend/cuda/include/ — build.rs:78-137 — and
 * backends/tfhe-cuda-common/src/cuda_bind.rs), with the same symbol name, argument meaning,
 * asynchrony and error behaviour, so `tfhe::core_crypto::gpu` links against
 * libtfhe_hip_backend.so unchanged for this path (INTEGRATION.md shows the binding).
 *
 * Conventions inherited from the reference boundary:
 *   - `stream` is an opaque handle made by cuda_create_stream_ffi (a hipStream_t here);
 *     `*_async` functions only enqueue on it; `cleanup_*` frees then synchronises;
 *     non-`_async` `cuda_*` functions synchronise internally
 *     (scripts/check_scratch_cleanup.py:1-20).
 *   - all ciphertext / key pointers are DEVICE pointers owned by the caller, except the
 *     `src` of the key-conversion functions, which is a HOST pointer
 *     (tfhe/src/core_crypto/gpu/ffi.rs:744-787).
 *   - no error codes: misuse prints to stderr and abort()s
 *     (backends/tfhe-cuda-common/cuda/include/device.h:13-41).
 *   - `*_indexes` are u64 arrays ON THE DEVICE selecting list element i
 *     (cuda/src/pbs/programmable_bootstrap_classic.cuh:821-826).
 *   - the bootstrap-key device buffer is opaque to the caller and has the reference's byte
 *     size: n*(k+1)^2*l*N doubles (gpu/entities/lwe_bootstrap_key.rs:57-104).
 *
 * Functions prefixed `hip_` are extensions with no counterpart in the reference GPU ABI
 * (the Goldilocks-NTT engine of cc/algorithms/lwe_programmable_bootstrapping/ntt64_bnf_pbs.rs
 * exists only on the reference's CPU side) plus test hooks.
 */
#ifndef TFHE_HIP_BACKEND_H
#define TFHE_HIP_BACKEND_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* backends/tfhe-cuda-backend/cuda/include/pbs/pbs_enums.h:4-6 */
enum PBS_TYPE { MULTI_BIT = 0, CLASSICAL = 1 };
enum PBS_VARIANT { DEFAULT = 0, CG = 1, TBC = 2 };
enum PBS_MS_REDUCTION_T { NO_REDUCTION = 0, CENTERED = 1 };
/* cuda/include/integer/integer.h:9-22 */
enum SHIFT_OR_ROTATE_TYPE { LEFT_SHIFT = 0, RIGHT_SHIFT = 1, LEFT_ROTATE = 2, RIGHT_ROTATE = 3 };
enum BITOP_TYPE { BITAND = 0, BITOR = 1, BITXOR = 2, SCALAR_BITAND = 3, SCALAR_BITOR = 4, SCALAR_BITXOR = 5 };
/* cuda/include/integer/integer.h:24-33 */
enum COMPARISON_TYPE { EQ = 0, NE = 1, GT = 2, GE = 3, LT = 4, LE = 5, MAX = 6, MIN = 7 };

/* ------------------------------------------------------------------ device runtime
 * backends/tfhe-cuda-common/cuda/include/device.h:58-92 (cuda_bind.rs:5-150) */
void *cuda_create_stream_ffi(uint32_t gpu_index);
void cuda_destroy_stream(void *stream, uint32_t gpu_index);
void cuda_synchronize_stream(void *stream, uint32_t gpu_index);
uint32_t cuda_is_available(void);
void *cuda_malloc(uint64_t size, uint32_t gpu_index);
/* stream-ordered: an enqueue on `stream`, no device synchronisation, usable under stream capture — served by the library's own
 * arena (hip_backend_trim_allocator below; TFHE_HIP_MALLOC_ASYNC=sync|pool|pool_hipfree select a plain hipMalloc or the
 * runtime's hipMallocAsync pool, which corrupted live allocations on ROCm 7.2.0: INTEGRATION.md).  cuda_drop returns such a
 * block in stream order behind the work queued on `stream` so far (device.cu:176-226, 457-491). */
void *cuda_malloc_async(uint64_t size, void *stream, uint32_t gpu_index);
bool cuda_check_valid_malloc(uint64_t size, uint32_t gpu_index);
uint64_t cuda_device_total_memory(uint32_t gpu_index);
void cuda_memcpy_async_to_gpu(void *dest, const void *src, uint64_t size, void *stream, uint32_t gpu_index);
void cuda_memcpy_async_gpu_to_gpu(void *dest, void const *src, uint64_t size, void *stream, uint32_t gpu_index);
void cuda_memcpy_gpu_to_gpu(void *dest, void const *src, uint64_t size, uint32_t gpu_index);
void cuda_memcpy_async_to_cpu(void *dest, const void *src, uint64_t size, void *stream, uint32_t gpu_index);
void cuda_memset_async(void *dest, uint64_t val, uint64_t size, void *stream, uint32_t gpu_index);
int cuda_get_number_of_gpus(void);
int cuda_get_number_of_sms(void);
void cuda_synchronize_device(uint32_t gpu_index);
void cuda_drop(void *ptr, uint32_t gpu_index);

/* ------------------------------------------------------------------ classic PBS
 * backends/tfhe-cuda-backend/cuda/include/pbs/programmable_bootstrap.h:52-55,62-66,83-90,99-100
 * called from tfhe/src/core_crypto/gpu/ffi.rs:21-92
 * polynomial_size: a power of two in 256..16384 (the reference's range); glwe_dimension 1..3 up to 2048,
 * 1 from 4096 up.  The scratch query returns 0 device bytes up to 4096 (the accumulator stays in LDS) and
 * num_samples * (k+1) * N * 8 beyond. */
void cuda_convert_lwe_programmable_bootstrap_key_64_async(
    void *stream, uint32_t gpu_index, void *dest, void const *src,
    uint32_t input_lwe_dim, uint32_t glwe_dim, uint32_t level_count,
    uint32_t polynomial_size);

uint64_t scratch_cuda_programmable_bootstrap_64_async(
    void *stream, uint32_t gpu_index, int8_t **buffer, uint32_t lwe_dimension,
    uint32_t glwe_dimension, uint32_t polynomial_size, uint32_t level_count,
    uint32_t input_lwe_ciphertext_count, bool allocate_gpu_memory,
    enum PBS_MS_REDUCTION_T noise_reduction_type);

void cuda_programmable_bootstrap_64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lut_vector,
    void const *lut_vector_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *bootstrapping_key,
    int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
    uint32_t polynomial_size, uint32_t base_log, uint32_t level_count,
    uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride);

void cleanup_cuda_programmable_bootstrap_64(void *stream, uint32_t gpu_index, int8_t **pbs_buffer);

/* extension: the shortint atomic pattern KS -> PBS (tfhe/src/shortint/atomic_pattern/standard.rs:162-199) in ONE
 * call on a scratch made by hip_scratch_keyswitch_programmable_bootstrap_64_async (the classic PBS scratch plus the
 * keyswitched list and its indexes; same parameters and cleanup as scratch_cuda_programmable_bootstrap_64_async, whose
 * own size the reference's callers track): lwe_array_in holds ciphertexts under the BIG key (dimension
 * glwe_dimension * polynomial_size), ksk the big -> small keyswitch key.  Two launches on `stream`, no allocation,
 * no host synchronisation. */
uint64_t hip_scratch_keyswitch_programmable_bootstrap_64_async(
    void *stream, uint32_t gpu_index, int8_t **buffer, uint32_t lwe_dimension,
    uint32_t glwe_dimension, uint32_t polynomial_size, uint32_t level_count,
    uint32_t input_lwe_ciphertext_count, bool allocate_gpu_memory,
    enum PBS_MS_REDUCTION_T noise_reduction_type);
void hip_keyswitch_programmable_bootstrap_64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out, void const *lwe_output_indexes,
    void const *lut_vector, void const *lut_vector_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *ksk, void const *bootstrapping_key, int8_t *buffer,
    uint32_t lwe_dimension, uint32_t glwe_dimension, uint32_t polynomial_size, uint32_t ks_base_log,
    uint32_t ks_level, uint32_t base_log, uint32_t level_count, uint32_t num_samples,
    uint32_t num_many_lut, uint32_t lut_stride);

/* extension: the same call for CHAINS of rounds in which a round's keyswitch reads exactly what the previous round's
 * bootstrap wrote (apply-lookup-table chains): the sample extraction of the bootstrap is fused with the digit pass of
 * the NEXT keyswitch.  flags: HIP_KSPBS_EMIT_DIGITS — the bootstrap also writes, for every output ciphertext, the
 * shifted keyswitch digits of its mask (decomposition ks_base_log / ks_level of this call) as the int8 A operands of the
 * keyswitch GEMM, into the scratch; HIP_KSPBS_INPUT_FROM_PREVIOUS — the caller states that lwe_array_in /
 * lwe_input_indexes are the lwe_array_out / lwe_output_indexes of the previous call on this scratch and that nothing
 * wrote to them since: the keyswitch starts from the emitted operands (the library checks pointers, count and
 * decomposition against what it recorded and falls back to its digit pass otherwise).  Emission is done by the
 * N = 2048, k = 1 throughput kernel (more than 256 LWEs) for keyswitch level counts padded to 4 or 8 with
 * base_log * level <= 30; elsewhere the flags change nothing.  Identical bits with and without the flags. */
#define HIP_KSPBS_EMIT_DIGITS 1u
#define HIP_KSPBS_INPUT_FROM_PREVIOUS 2u
void hip_keyswitch_programmable_bootstrap_chain_64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out, void const *lwe_output_indexes,
    void const *lut_vector, void const *lut_vector_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *ksk, void const *bootstrapping_key, int8_t *buffer,
    uint32_t lwe_dimension, uint32_t glwe_dimension, uint32_t polynomial_size, uint32_t ks_base_log,
    uint32_t ks_level, uint32_t base_log, uint32_t level_count, uint32_t num_samples,
    uint32_t num_many_lut, uint32_t lut_stride, uint32_t flags);

/* ------------------------------------------------------------------ multi-bit PBS
 * backends/tfhe-cuda-backend/cuda/include/pbs/programmable_bootstrap_multibit.h:9-40
 * called from tfhe/src/core_crypto/gpu/ffi.rs:208-309,789-835
 * polynomial_size: a power of two in 256..16384, glwe_dimension as for the classic PBS (1..3 up to 1024, 1..2 at
 * 2048, 1 beyond); grouping_factor 1..4.
 * The scratch holds everything a launch needs (nothing is allocated by the launch itself). */
bool has_support_to_cuda_programmable_bootstrap_cg_multi_bit(
    uint32_t glwe_dimension, uint32_t polynomial_size, uint32_t level_count,
    uint32_t num_samples, uint32_t max_shared_memory);

void cuda_convert_lwe_multi_bit_programmable_bootstrap_key_64_async(
    void *stream, uint32_t gpu_index, void *dest, void const *src,
    uint32_t input_lwe_dim, uint32_t glwe_dim, uint32_t level_count,
    uint32_t polynomial_size, uint32_t grouping_factor);

uint64_t scratch_cuda_multi_bit_programmable_bootstrap_64_async(
    void *stream, uint32_t gpu_index, int8_t **pbs_buffer,
    uint32_t glwe_dimension, uint32_t polynomial_size, uint32_t level_count,
    uint32_t input_lwe_ciphertext_count, bool allocate_gpu_memory);

void cuda_multi_bit_programmable_bootstrap_64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lut_vector,
    void const *lut_vector_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *bootstrapping_key,
    int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
    uint32_t polynomial_size, uint32_t grouping_factor, uint32_t base_log,
    uint32_t level_count, uint32_t num_samples, uint32_t num_many_lut,
    uint32_t lut_stride);

void cleanup_cuda_multi_bit_programmable_bootstrap_64(void *stream, uint32_t gpu_index, int8_t **pbs_buffer);

/* The variant the reference's noise tests call (cuda/include/pbs/programmable_bootstrap_multibit.h:44-60; bound by
 * tfhe/src/core_crypto/gpu/ffi.rs:322-397): the input has ALREADY been through cuda_modulus_switch_multi_bit_64_async —
 * lwe_array_in = [ ciphertext, lwe_dimension + 1 words | its output, (lwe_dimension / grouping_factor) * 2^grouping_factor
 * words ] — and the keybundle reads its monomial degrees from the second part.  num_samples must be 1 and
 * polynomial_size 2048 (anything else panics, as there); scratch and cleanup are the standard ones. */
uint64_t scratch_cuda_multi_bit_programmable_bootstrap_noise_tests_64_async(
    void *stream, uint32_t gpu_index, int8_t **pbs_buffer,
    uint32_t glwe_dimension, uint32_t polynomial_size, uint32_t level_count,
    uint32_t input_lwe_ciphertext_count, bool allocate_gpu_memory);

void cleanup_cuda_multi_bit_programmable_bootstrap_noise_tests_64(
    void *stream, uint32_t gpu_index, int8_t **pbs_buffer);

void cuda_multi_bit_programmable_bootstrap_noise_tests_64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lut_vector,
    void const *lut_vector_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *bootstrapping_key,
    int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
    uint32_t polynomial_size, uint32_t grouping_factor, uint32_t base_log,
    uint32_t level_count, uint32_t num_samples, uint32_t num_many_lut,
    uint32_t lut_stride);

/* ------------------------------------------------------------------ keyswitch
 * backends/tfhe-cuda-backend/cuda/include/keyswitch/keyswitch.h:16-21,35-41,69-72
 * called from tfhe/src/core_crypto/gpu/ffi.rs:503-618 (KSK upload is a plain memcpy, :620-627) */
void cuda_keyswitch_lwe_ciphertext_vector_64_64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *ksk, uint32_t lwe_dimension_in,
    uint32_t lwe_dimension_out, uint32_t base_log, uint32_t level_count,
    uint32_t num_samples);

void cuda_keyswitch_gemm_64_64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *ksk, uint32_t lwe_dimension_in,
    uint32_t lwe_dimension_out, uint32_t base_log, uint32_t level_count,
    uint32_t num_samples, bool uses_trivial_indexes);

/* u64 input ciphertexts, u32 key and u32 output ciphertexts (the KS32 atomic pattern):
 * backends/tfhe-cuda-backend/cuda/include/keyswitch/keyswitch.h:23-28,42-47; semantics
 * tfhe/src/core_crypto/algorithms/lwe_keyswitch.rs:331-447 (body rounded to 32 bits, base_log*level <= 32);
 * called from tfhe/src/core_crypto/gpu/ffi.rs:503-618 when the key scalar is u32. */
void cuda_keyswitch_lwe_ciphertext_vector_64_32_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *ksk, uint32_t lwe_dimension_in,
    uint32_t lwe_dimension_out, uint32_t base_log, uint32_t level_count,
    uint32_t num_samples);

void cuda_keyswitch_gemm_64_32_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *ksk, uint32_t lwe_dimension_in,
    uint32_t lwe_dimension_out, uint32_t base_log, uint32_t level_count,
    uint32_t num_samples, bool uses_trivial_indexes);

void cuda_closest_representable_64_async(void *stream, uint32_t gpu_index,
                                         void const *input, void *output,
                                         uint32_t base_log, uint32_t level_count);

/* ------------------------------------------------------------------ ciphertext helpers
 * backends/tfhe-cuda-backend/cuda/include/ciphertext.h:5-42
 * called from tfhe/src/core_crypto/gpu/ffi.rs:838-898 */
void cuda_convert_lwe_ciphertext_vector_to_gpu_64_async(
    void *stream, uint32_t gpu_index, void *dest, void const *src,
    uint32_t number_of_cts, uint32_t lwe_dimension);
void cuda_convert_lwe_ciphertext_vector_to_cpu_64_async(
    void *stream, uint32_t gpu_index, void *dest, void const *src,
    uint32_t number_of_cts, uint32_t lwe_dimension);
void cuda_glwe_sample_extract_64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *glwe_array_in, uint32_t const *nth_array, uint32_t num_nths,
    uint32_t num_lwes_to_extract_per_glwe, uint32_t num_lwes_stored_per_glwe,
    uint32_t glwe_dimension, uint32_t polynomial_size);
void cuda_modulus_switch_inplace_64_async(void *stream, uint32_t gpu_index,
                                          void *lwe_array_out, uint32_t size,
                                          uint32_t log_modulus);
void cuda_modulus_switch_64_async(void *stream, uint32_t gpu_index,
                                  void *lwe_out, const void *lwe_in,
                                  uint32_t size, uint32_t log_modulus);
void cuda_centered_modulus_switch_64_async(void *stream, uint32_t gpu_index,
                                           void *lwe_out, const void *lwe_in,
                                           uint32_t lwe_dimension,
                                           uint32_t log_modulus);
/* cuda/include/ciphertext.h:34-37 (modulus_switch.rs:386-488 tests it): the same switch with the body correction
   reduced as the bootstrap kernels reduce it, in one block of shape (block_dim_x, block_dim_y): 128 threads (the
   (64, 2) block of the throughput kernel: per wave, as pbs_fft_wave.hip's prologue) or 512 (the block kernels'
   prologue); any other size aborts as in the reference (torus.cuh:460-463) */
void cuda_centered_modulus_switch_cooperative_64_async(
    void *stream, uint32_t gpu_index, void *lwe_out, const void *lwe_in,
    uint32_t lwe_dimension, uint32_t log_modulus, uint32_t block_dim_x,
    uint32_t block_dim_y);
/* cuda/include/pbs/programmable_bootstrap.h:8-45: the transform of the path as launches of its own, what the reference's
   backend tests drive (tests_and_benchmarks/tests/test_fft.cpp, test_forward_fft16x4x16.cpp, test_fft16x4x16.cpp;
   gpu/algorithms/test/fft/mod.rs:268-294 + its golden spectrum; gpu/ffi.rs:1070-1085).  Polynomials "compressed":
   complex[i] = (p[i], p[i + N/2]).  polynomial_mul: negacyclic product, input1 is overwritten with its spectrum (as in the
   reference), sizes 256 .. 16384.  forward_fft_classic: spectrum in the native (tree) order; forward_fft16x4x16: in NATURAL
   frequency order, natural f <-> tree index bitreverse((N/2 - f) mod N/2); backward_fft16x4x16: its inverse WITHOUT the 1/(N/2)
   scaling.  The last four take polynomial_size 2048 only (abort otherwise, as the reference); is_supported: true. */
void cuda_fourier_polynomial_mul_async(void *stream, uint32_t gpu_index,
                                       void const *input1, void const *input2,
                                       void *output, uint32_t polynomial_size,
                                       uint32_t total_polynomials);
void cuda_fourier_polynomial_mul_fft16x4x16_async(
    void *stream, uint32_t gpu_index, void const *input1, void const *input2,
    void *output, uint32_t polynomial_size, uint32_t total_polynomials);
void cuda_forward_fft_classic_async(void *stream, uint32_t gpu_index,
                                    void const *input, void *output,
                                    uint32_t polynomial_size,
                                    uint32_t total_polynomials);
void cuda_forward_fft16x4x16_async(void *stream, uint32_t gpu_index,
                                   void const *input, void *output,
                                   uint32_t polynomial_size,
                                   uint32_t total_polynomials);
void cuda_backward_fft16x4x16_async(void *stream, uint32_t gpu_index,
                                    void const *input, void *output,
                                    uint32_t polynomial_size,
                                    uint32_t total_polynomials);
bool cuda_fft16x4x16_is_supported_async(uint32_t gpu_index);
/* cuda/include/ciphertext.h:45-50 (tfhe/src/core_crypto/gpu/ffi.rs:914-936): the multi-bit switch as its own launch
 * (noise tests only; production fuses it into the keybundle).  `size` words of lwe_array_in are read as
 * size / grouping_factor groups, 2^grouping_factor degrees are written per group ([group][subset], subset 0 = 0).
 * As in the reference the switch goes to 2 * degree whatever log_modulus says, and only degree 2048 is accepted. */
void cuda_modulus_switch_multi_bit_64_async(void *stream, uint32_t gpu_index,
                                            void *lwe_array_out,
                                            void *lwe_array_in, uint32_t size,
                                            uint32_t log_modulus,
                                            uint32_t degree,
                                            uint32_t grouping_factor);

/* ------------------------------------------------------------------ extensions (hip_*)
 * Goldilocks-NTT engine: same argument meaning as the classic PBS triple above, the key
 * buffer has the same byte size (n*(k+1)^2*l*N u64).  Semantics:
 * cc/algorithms/lwe_programmable_bootstrapping/ntt64_bnf_pbs.rs:469-539 and
 * cc/algorithms/lwe_bootstrap_key_conversion.rs:367-434 (bit-exact, integer only). */
void hip_convert_lwe_programmable_bootstrap_key_ntt64_async(
    void *stream, uint32_t gpu_index, void *dest, void const *src,
    uint32_t input_lwe_dim, uint32_t glwe_dim, uint32_t level_count,
    uint32_t polynomial_size);
void hip_programmable_bootstrap_ntt64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lut_vector,
    void const *lut_vector_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *bootstrapping_key,
    int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
    uint32_t polynomial_size, uint32_t base_log, uint32_t level_count,
    uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride);

/* The same engine on the f64 transform machinery of the throughput kernel (pbs_fft_wave.hip, split-key form): every
 * key word, switched to the prime and centred, is cut into 4 balanced 16-bit limbs kept in the Fourier domain; per
 * CMUX the digit transform is multiplied with each limb, the inverse transforms are rounded to the exact integer
 * products (|.| < 2^49; the distance from an integer is checked on every coefficient, the launch traps above 1/4) and
 * recombined modulo the prime.  Identical outputs to hip_programmable_bootstrap_ntt64_async.  Accepts N = 2048,
 * k = 1, one level, base_log 22 or 23 (hip_programmable_bootstrap_ntt64_split_supported).  The key buffer takes FOUR
 * times the bytes of the standard key: n*(k+1)^2*N*32.  The first launch on a scratch allocates the accumulators'
 * device buffer (not capturable; later launches are). */
bool hip_programmable_bootstrap_ntt64_split_supported(
    uint32_t glwe_dimension, uint32_t polynomial_size, uint32_t level_count,
    uint32_t base_log);
void hip_convert_lwe_programmable_bootstrap_key_ntt64_split_async(
    void *stream, uint32_t gpu_index, void *dest, void const *src,
    uint32_t input_lwe_dim, uint32_t glwe_dim, uint32_t level_count,
    uint32_t polynomial_size);
void hip_programmable_bootstrap_ntt64_split_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lut_vector,
    void const *lut_vector_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *bootstrapping_key,
    int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
    uint32_t polynomial_size, uint32_t base_log, uint32_t level_count,
    uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride);

/* Round-off check of the split-key engine.  A ciphertext whose f64 limb products were further than 1/4 from integers (the
 * bound on them is statistical: worst-case products of 2^49 leave 4 bits of headroom; never observed on a real parameter
 * set's data, reachable with adversarial constant polynomials) is RECOMPUTED by the integer Goldilocks kernel in a second
 * launch on the same stream, with the NTT-domain twin of the key that the conversion above made and the library keeps while
 * the split key's memory lives: the entry point's outputs are the exact ones whatever the data.  This call returns how many
 * ciphertexts went that way on this scratch since the last call (synchronises the stream, clears the count).  A split key
 * the library did not convert itself (copied in) has no twin: a raised flag then panics here or at cleanup_*. */
uint32_t hip_programmable_bootstrap_ntt64_split_roundoff_status(
    void *stream, uint32_t gpu_index, int8_t *buffer);

/* Exact-integer engine (negacyclic convolution mod 2^64 on the standard-domain key,
 * cc/algorithms/lwe_programmable_bootstrapping/karatsuba_pbs.rs:71-116,199-413).  O(N^2): a
 * verification engine; it reproduces the reference's golden *_karatsuba vectors bit for bit. */
void hip_convert_lwe_programmable_bootstrap_key_exact64_async(
    void *stream, uint32_t gpu_index, void *dest, void const *src,
    uint32_t input_lwe_dim, uint32_t glwe_dim, uint32_t level_count,
    uint32_t polynomial_size);
void hip_programmable_bootstrap_exact64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lut_vector,
    void const *lut_vector_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *bootstrapping_key,
    int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
    uint32_t polynomial_size, uint32_t base_log, uint32_t level_count,
    uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride);

/* Reference-order f64 engine: the blind rotation with tfhe-fft's radix-4 DIF Stockham plan and the reference's x86
 * conversion / multiply-accumulate forms, operation for operation (csrc/pbs_ref64.hip).  A verification engine:
 * it reproduces the reference's f64 golden vectors (apps/test-vectors, lwe_after_{id,spec}_pbs.cbor, made with
 * `experimental-force_fft_algo_dif4`) bit for bit on the MI355X.  glwe_dimension 1, polynomial_size <= 2048. */
void hip_convert_lwe_programmable_bootstrap_key_ref64_async(
    void *stream, uint32_t gpu_index, void *dest, void const *src,
    uint32_t input_lwe_dim, uint32_t glwe_dim, uint32_t level_count,
    uint32_t polynomial_size);
void hip_programmable_bootstrap_ref64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out,
    void const *lwe_output_indexes, void const *lut_vector,
    void const *lut_vector_indexes, void const *lwe_array_in,
    void const *lwe_input_indexes, void const *bootstrapping_key,
    int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
    uint32_t polynomial_size, uint32_t base_log, uint32_t level_count,
    uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride);

/* ------------------------------------------------------------------ radix integers ("next" row N1)
 * backends/tfhe-cuda-backend/cuda/include/integer/integer.h:52-65,100-113 (FFI structs),
 * :127-148 (apply_univariate_lut), :173-187 (integer_mult_inplace), :383-413 (propagate_single_carry,
 * add_and_propagate_single_carry); include/keyswitch/keyswitch.h:7-12; include/linear_algebra.h:26-28.
 * Called from tfhe/src/integer/gpu/mod.rs (cuda_backend_apply_univariate_lut,
 * cuda_backend_propagate_single_carry_assign, cuda_backend_unchecked_mul_assign, ...).
 *
 * Wired: classic and multi-bit keys (pbs_type), carry-in (uses_carry), FLAG_CARRY, FLAG_OVERFLOW (add_and_propagate),
 * boolean operands of integer_mult, many-LUT application; a round spreads over the GPUs of the CudaStreamsFFI by the
 * reference's thresholds (hip_integer_set_multi_gpu_threshold below).  Operands must be clean up to what an operation
 * accepts (degrees are checked when given: INTEGRATION.md).  Extension: a CudaRadixCiphertextFFI may hold a batch of integers
 * ([integer][block]); the scratch's num_blocks is the blocks PER integer and the launch handles
 * num_radix_blocks / num_blocks integers in the same rounds (capacity: hip_integer_scratch_batch). */
typedef struct {
  void *const *streams;
  uint32_t const *gpu_indexes;
  uint32_t gpu_count;
} CudaStreamsFFI;

typedef struct {
  void *ptr;
  uint64_t *degrees;
  uint64_t *noise_levels;
  uint32_t num_radix_blocks;
  uint32_t max_num_radix_blocks;
  uint32_t lwe_dimension;
} CudaRadixCiphertextFFI;

typedef struct {
  uint32_t input_lwe_dimension;
  uint32_t glwe_dimension;
  uint32_t polynomial_size;
  uint32_t base_log;
  uint32_t level_count;
  uint32_t big_lwe_dimension;
  uint32_t pbs_type; /* PBS_TYPE: MULTI_BIT = 0, CLASSICAL = 1 (pbs/pbs_enums.h:4) */
  uint32_t grouping_factor;
} CudaLweBootstrapKeyParamsFFI;

typedef struct {
  uint32_t input_lwe_dimension;
  uint32_t output_lwe_dimension;
  uint32_t base_log;
  uint32_t level_count;
} CudaLweKeyswitchKeyParamsFFI;

uint64_t scratch_cuda_apply_univariate_lut_64_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, void const *input_lut,
    CudaLweBootstrapKeyParamsFFI bsk_params, CudaLweKeyswitchKeyParamsFFI ksk_params,
    uint32_t input_lwe_ciphertext_count, uint32_t message_modulus, uint32_t carry_modulus,
    uint64_t lut_degree, bool allocate_gpu_memory, enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_apply_univariate_lut_64_async(
    CudaStreamsFFI streams, CudaRadixCiphertextFFI *output_radix_lwe,
    CudaRadixCiphertextFFI const *input_radix_lwe, int8_t *mem_ptr, void *const *ksks, void *const *bsks);
void cleanup_cuda_apply_univariate_lut_64(CudaStreamsFFI streams, int8_t **mem_ptr_void);
/* integer.h:135-160: `input_lut` is ONE accumulator that packs num_many_lut functions in sub-tables of lut_stride
 * coefficients (tfhe/src/shortint/engine/mod.rs:169-254); one keyswitch and one PBS per block, the PBS extracts
 * num_luts samples; function t of input block s is written to output block t * n + s. */
uint64_t scratch_cuda_apply_many_univariate_lut_64_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, void const *input_lut,
    CudaLweBootstrapKeyParamsFFI bsk_params, CudaLweKeyswitchKeyParamsFFI ksk_params,
    uint32_t num_radix_blocks, uint32_t message_modulus, uint32_t carry_modulus, uint32_t num_many_lut,
    uint64_t lut_degree, bool allocate_gpu_memory, enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_apply_many_univariate_lut_64_async(
    CudaStreamsFFI streams, CudaRadixCiphertextFFI *output_radix_lwe,
    CudaRadixCiphertextFFI const *input_radix_lwe, int8_t *mem_ptr, void *const *ksks, void *const *bsks,
    uint32_t num_luts, uint32_t lut_stride);
void cleanup_cuda_apply_many_univariate_lut_64(CudaStreamsFFI streams, int8_t **mem_ptr_void);

void cuda_add_lwe_ciphertext_vector_inplace_64(
    void *stream, uint32_t gpu_index, CudaRadixCiphertextFFI *lwe_array_inout,
    CudaRadixCiphertextFFI const *input_2);

uint64_t scratch_cuda_propagate_single_carry_64_inplace_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks, uint32_t message_modulus,
    uint32_t carry_modulus, uint32_t requested_flag, bool allocate_gpu_memory,
    enum PBS_MS_REDUCTION_T noise_reduction_type);
uint64_t scratch_cuda_add_and_propagate_single_carry_64_inplace_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks, uint32_t message_modulus,
    uint32_t carry_modulus, uint32_t requested_flag, bool allocate_gpu_memory,
    enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_propagate_single_carry_64_inplace_async(
    CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array, CudaRadixCiphertextFFI *carry_out,
    const CudaRadixCiphertextFFI *carry_in, int8_t *mem_ptr, void *const *bsks, void *const *ksks,
    uint32_t requested_flag, uint32_t uses_carry);
void cuda_add_and_propagate_single_carry_64_inplace_async(
    CudaStreamsFFI streams, CudaRadixCiphertextFFI *lhs_array, const CudaRadixCiphertextFFI *rhs_array,
    CudaRadixCiphertextFFI *carry_out, const CudaRadixCiphertextFFI *carry_in, int8_t *mem_ptr,
    void *const *bsks, void *const *ksks, uint32_t requested_flag, uint32_t uses_carry);
void cleanup_cuda_propagate_single_carry_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void);
void cleanup_cuda_add_and_propagate_single_carry_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void);

uint64_t scratch_cuda_integer_mult_inplace_64_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, bool const is_boolean_left, bool const is_boolean_right,
    uint32_t message_modulus, uint32_t carry_modulus, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks, bool allocate_gpu_memory,
    enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_integer_mult_inplace_64_async(
    CudaStreamsFFI streams, CudaRadixCiphertextFFI *radix_lwe_inout, bool const is_bool_left,
    CudaRadixCiphertextFFI const *radix_lwe_right, bool const is_bool_right, void *const *bsks,
    void *const *ksks, int8_t *mem_ptr, uint32_t polynomial_size, uint32_t num_blocks);
void cleanup_cuda_integer_mult_inplace_64(CudaStreamsFFI streams, int8_t **mem_ptr_void);

/* Levelled block operations and the one-round / sequential operations built on the round driver (round 6).
 * integer.h:189-198 (negate with the correcting term, scalar addition), :312-347 (bitnot, bitop, scalar bitop),
 * :559-573 (sub_and_propagate_single_carry: requested_flag 0 or 2, no input carry — what the reference's callers pass,
 * integer/gpu/server_key/radix/sub.rs:222,347-400), :159-171 (full propagation, block after block).
 * negate / scalar_addition / bitnot / bitop / scalar_bitop / full_propagation take ONE integer per ciphertext as in the
 * reference (bitop: any number of blocks up to lwe_ciphertext_count x hip_integer_scratch_batch); sub takes the batch
 * extension described above. */
void cuda_negate_ciphertext_64(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_out,
                               CudaRadixCiphertextFFI const *lwe_array_in, uint32_t message_modulus,
                               uint32_t carry_modulus, uint32_t num_radix_blocks);
void cuda_scalar_addition_ciphertext_64_inplace(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array,
                                                void const *scalar_input, void const *h_scalar_input,
                                                uint32_t num_scalars, uint32_t message_modulus, uint32_t carry_modulus);
void cuda_bitnot_ciphertext_64(CudaStreamsFFI streams, CudaRadixCiphertextFFI *radix_ciphertext,
                               uint32_t ct_message_modulus, uint32_t param_message_modulus,
                               uint32_t param_carry_modulus);
uint64_t scratch_cuda_integer_bitop_inplace_64_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t lwe_ciphertext_count, uint32_t message_modulus,
    uint32_t carry_modulus, enum BITOP_TYPE op_type, bool allocate_gpu_memory,
    enum PBS_MS_REDUCTION_T noise_reduction_type);
uint64_t scratch_cuda_integer_scalar_bitop_inplace_64_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t lwe_ciphertext_count, uint32_t message_modulus,
    uint32_t carry_modulus, enum BITOP_TYPE op_type, bool allocate_gpu_memory,
    enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_integer_bitop_inplace_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_inout,
                                         CudaRadixCiphertextFFI const *lwe_array_2, int8_t *mem_ptr,
                                         void *const *bsks, void *const *ksks);
void cuda_integer_scalar_bitop_inplace_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_inout,
                                                void const *clear_blocks, void const *h_clear_blocks,
                                                uint32_t num_clear_blocks, int8_t *mem_ptr, void *const *bsks,
                                                void *const *ksks);
void cleanup_cuda_integer_bitop_inplace_64(CudaStreamsFFI streams, int8_t **mem_ptr_void);
void cleanup_cuda_integer_scalar_bitop_inplace_64(CudaStreamsFFI streams, int8_t **mem_ptr_void);
uint64_t scratch_cuda_sub_and_propagate_single_carry_64_inplace_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks, uint32_t message_modulus, uint32_t carry_modulus,
    uint32_t requested_flag, bool allocate_gpu_memory, enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_sub_and_propagate_single_carry_64_inplace_async(
    CudaStreamsFFI streams, CudaRadixCiphertextFFI *lhs_array, const CudaRadixCiphertextFFI *rhs_array,
    CudaRadixCiphertextFFI *carry_out, const CudaRadixCiphertextFFI *carry_in, int8_t *mem_ptr,
    void *const *bsks, void *const *ksks, uint32_t requested_flag, uint32_t uses_carry);
void cleanup_cuda_sub_and_propagate_single_carry_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void);
/* integer.h:415-431: lhs -= rhs and the borrow (1 - the carry of lhs + (2^bits - rhs)); no input borrow */
uint64_t scratch_cuda_integer_overflowing_sub_64_inplace_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks, uint32_t message_modulus, uint32_t carry_modulus,
    uint32_t compute_overflow, bool allocate_gpu_memory, enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_integer_overflowing_sub_64_inplace_async(
    CudaStreamsFFI streams, CudaRadixCiphertextFFI *lhs_array, const CudaRadixCiphertextFFI *rhs_array,
    CudaRadixCiphertextFFI *overflow_block, const CudaRadixCiphertextFFI *input_borrow, int8_t *mem_ptr,
    void *const *bsks, void *const *ksks, uint32_t compute_overflow, uint32_t uses_input_borrow);
void cleanup_cuda_integer_overflowing_sub_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void);
uint64_t scratch_cuda_full_propagation_64_inplace_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t message_modulus, uint32_t carry_modulus,
    bool allocate_gpu_memory, enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_full_propagation_64_inplace_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *input_blocks,
                                            int8_t *mem_ptr, void *const *ksks, void *const *bsks,
                                            uint32_t num_blocks);
void cleanup_cuda_full_propagation_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void);
/* integer.h:246-276 (comparison: unsigned operands, EQ ... LE into block 0 of the output, MAX / MIN over all blocks) and
 * :349-365 (cmux).  Orderings ride on the subtraction's carry tree without its result round, equality on sums of block results. */
uint64_t scratch_cuda_integer_comparison_64_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t lwe_ciphertext_count, uint32_t message_modulus,
    uint32_t carry_modulus, enum COMPARISON_TYPE op_type, bool is_signed, bool allocate_gpu_memory,
    enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_integer_comparison_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_out,
                                      CudaRadixCiphertextFFI const *lwe_array_1,
                                      CudaRadixCiphertextFFI const *lwe_array_2, int8_t *mem_ptr, void *const *bsks,
                                      void *const *ksks);
void cleanup_cuda_integer_comparison_64(CudaStreamsFFI streams, int8_t **mem_ptr_void);
uint64_t scratch_cuda_integer_scalar_comparison_64_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t lwe_ciphertext_count, uint32_t message_modulus,
    uint32_t carry_modulus, enum COMPARISON_TYPE op_type, bool is_signed, bool allocate_gpu_memory,
    enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_integer_scalar_comparison_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_out,
                                             CudaRadixCiphertextFFI const *lwe_array_in, void const *scalar_blocks,
                                             void const *h_scalar_blocks, int8_t *mem_ptr, void *const *bsks,
                                             void *const *ksks, uint32_t num_scalar_blocks);
void cleanup_cuda_integer_scalar_comparison_64(CudaStreamsFFI streams, int8_t **mem_ptr_void);
uint64_t scratch_cuda_cmux_64_async(CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
                                    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t lwe_ciphertext_count,
                                    uint32_t message_modulus, uint32_t carry_modulus, bool allocate_gpu_memory,
                                    enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_cmux_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_out,
                        CudaRadixCiphertextFFI const *lwe_condition, CudaRadixCiphertextFFI const *lwe_array_true,
                        CudaRadixCiphertextFFI const *lwe_array_false, int8_t *mem_ptr, void *const *bsks,
                        void *const *ksks);
void cleanup_cuda_cmux_64(CudaStreamsFFI streams, int8_t **mem_ptr_void);
/* integer.h:200-228: logical shift by a clear amount (LEFT_SHIFT / RIGHT_SHIFT; the rotations are refused): a move by whole
 * blocks plus one bivariate round when bits remain */
uint64_t scratch_cuda_logical_scalar_shift_64_inplace_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks, uint32_t message_modulus, uint32_t carry_modulus,
    enum SHIFT_OR_ROTATE_TYPE shift_type, bool allocate_gpu_memory, enum PBS_MS_REDUCTION_T noise_reduction_type);
void cuda_logical_scalar_shift_64_inplace_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array, uint32_t shift,
                                                int8_t *mem_ptr, void *const *bsks, void *const *ksks);
void cleanup_cuda_logical_scalar_shift_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void);

/* extensions: integers per launch the NEXT scratch_* call is sized for (default 1), and the number
 * of PBS one multiplication issues per integer (for throughput accounting) */
void hip_integer_scratch_batch(uint32_t num_integers);
/* blocks per GPU from which a KS -> PBS round of the radix layer spreads over one more GPU of its CudaStreamsFFI.
 * 0 (default): the reference's rule, cuda/src/utils/helper_multi_gpu.cu:16-48 (get_active_gpu_count) — 12 for
 * multi-bit keys, (compute units of the first GPU) + 1 for classic ones.  The stream set may also name the same GPU
 * several times (one stream each).  hip_integer_active_gpu_count: how many GPUs of a set of gpu_count a round of
 * num_blocks blocks uses under the current setting (pbs_type as PBS_TYPE: MULTI_BIT = 0, CLASSICAL = 1). */
void hip_integer_set_multi_gpu_threshold(uint32_t blocks_per_gpu);
uint32_t hip_integer_active_gpu_count(uint32_t num_blocks, uint32_t gpu_count, uint32_t pbs_type, uint32_t first_gpu);
uint64_t hip_integer_mult_pbs_count(int8_t *mem_ptr);
uint64_t hip_integer_propagate_pbs_count(uint32_t num_blocks);

/* Select which f64 kernel serves cuda_programmable_bootstrap_64_async (all give identical bits):
 * 0 = automatic (N=2048,k=1: latency kernel up to 256 LWEs, throughput kernel beyond; N=1024,k<=2: its
 *     throughput kernel; generic otherwise),
 * 1 = generic LDS kernel, 2 = throughput (wave) kernel of the parameter set, 3 = latency (block) kernel, 4 = its dual-stream
 * variant; 2..4 abort on unsupported parameter sets.  For the multi-bit entry point 2 selects the
 * multi-bit mode of the throughput kernel, 1 the generic multi-bit kernel, 5 the two-launch latency path (all
 * keybundles first, one workgroup per polynomial; then the products, for N=2048,k=1 on the latency kernel) that 0
 * takes up to 128 LWEs (256 for N = 2048, k = 1), 6 the same path with the products on the generic kernels (comparison), 7 the throughput
 * kernel without the sharing of key loads between the two LWEs of a quad of waves, 8 the throughput kernel with
 * sharing among quads only, not among all eight waves of a full workgroup of a one-level set (both: comparison). */
void hip_backend_set_fft_kernel(uint32_t which);
/* The stream-ordered arena behind cuda_malloc_async / cuda_drop (tfhe_rs_amd/csrc/arena.hip; the reference's device pool,
 * tfhe-cuda-common/cuda/src/device.cu:70-120,176-226).  trim: every idle cached block goes back to the runtime, returns the
 * bytes released (the reference's pool keeps a release threshold instead).  stats: out7 = allocations, re-uses, blocks taken
 * from the runtime, drops, re-uses that made the new stream wait for the old one's event, live bytes, cached bytes. */
uint64_t hip_backend_trim_allocator(uint32_t gpu_index);
void hip_backend_allocator_stats(uint32_t gpu_index, uint64_t *out7);
/* Debug mode TFHE_HIP_ARENA_REDZONE=1 (environment, read once): every device block of the library and of cuda_malloc[_async]
 * sits in a shared slab between two 4 KiB canaries that are checked when it is dropped, a dropped payload is poisoned and
 * checked when it is handed out again; a finding aborts with the block, its owner and the offset (the memcheck the
 * reference runs over its GPU tests, scripts/check_memory_errors.sh:1-60).  Returns the canary checks made so far
 * (0: the mode is off). */
uint64_t hip_backend_redzone_checks(uint32_t gpu_index);
/* TFHE_HIP_PROFILE=1 (environment, read once): the radix layer brackets its rounds with roctx ranges — "apply lut round",
 * "keyswitch", "bootstrap", "scatter ... to gpu", "gather ... from gpu", "carry propagation", "integer mul", "block products",
 * "column sums" — for rocprofv3 --marker-trace (the reference's NVTX ranges, tfhe-cuda-common/cuda/include/helper_profile.cuh,
 * cuda/src/integer/integer.cuh:874,958,981).  Returns the ranges pushed so far (0: off). */
uint64_t hip_backend_profile_ranges(void);
/* test hook: cap of the groups the multi-bit latency path processes per pass (0 = what the scratch holds) */
void hip_backend_set_multibit_latency_groups(uint32_t groups);
/* keyswitch kernel: 0 = automatic (int8 matrix-core GEMM, any batch size, when level <= 16 (padded to a power of
 * two), base_log <= 6 and n_in*padded level is a multiple of 32 — one launch up to 768 LWEs (K shared by the waves of
 * a workgroup and by up to 8 workgroups per column tile), from 769 on a digit pass followed by an LDS-staged GEMM;
 * scalar kernels otherwise), 1 = scalar kernels only, 2 = the one-launch matrix-core kernel at every batch size, 3 = the
 * digit pass + GEMM from 129 LWEs on (tests).  Identical bits in all four. */
void hip_backend_set_keyswitch_kernel(uint32_t which);
/* generic kernels (f64 and NTT engines): 0 = one thread group per GLWE polynomial (default), 1 = single-group
 * kernels. Same bits. */
void hip_backend_set_ntt_kernel(uint32_t which);
/* last launched PBS kernel, for tests: 1 generic f64, 2 wave f64, 3 generic ntt, 4 generic multi-bit,
 * 5 exact, 6 wave multi-bit, 7 block (latency), 8 block dual-stream, 9 wave f64 for N = 1024,
 * 10 multi-bit latency path, 11 reference-order f64 engine, 12 NTT engine in its two-prime FP64 form */
uint32_t hip_backend_last_pbs_kernel(void);
/* last 64-bit keyswitch, for tests: 0 scalar kernels, 1 one-launch matrix-core kernel, 2 digit pass + staged GEMM,
 * 3 staged GEMM on the digits the previous bootstrap emitted */
uint32_t hip_backend_last_keyswitch_path(void);
/* small-batch keyswitch (<= 32 LWEs): workgroups per column tile that share the K dimension (default 8; 1 = one
 * workgroup per column tile, no atomics).  Identical bits. */
void hip_backend_set_keyswitch_kparts(uint32_t parts);
const char *hip_backend_version(void);

/* HIP events recorded on a backend stream (used by bench.py to time launches) */
void *hip_event_create(void);
void hip_event_record(void *event, void *stream);
float hip_event_elapsed_ms(void *start, void *stop); /* synchronises `stop` */
void hip_event_destroy(void *event);

/* test hooks: run single device functions / transforms so that tests can compare them with
 * the oracle (ops documented in tfhe_rs_amd/csrc/testhooks.hip) */
void hip_test_arith_async(void *stream, uint32_t gpu_index, uint32_t op, void const *in,
                          void *out, uint32_t count, uint32_t p0, uint32_t p1);
void hip_test_transform_async(void *stream, uint32_t gpu_index, uint32_t op,
                              uint32_t polynomial_size, void const *in, void *out);
void hip_test_fft_tables_host(uint32_t polynomial_size, double *fwd, double *inv, double *untwist);
void hip_test_monomial_table_host(uint32_t polynomial_size, double *mono /* 4 * polynomial_size doubles */);

#ifdef __cplusplus
}
#endif
#endif /* TFHE_HIP_BACKEND_H */
