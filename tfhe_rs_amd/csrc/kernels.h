// kernels.h — host-callable launchers of the backend's kernels (internal to the library).
#pragma once
#include <atomic>
#include "pbs_common.h"

namespace tfhe_hip {

// generic (any supported N,k,l) engines — pbs_generic.hip
void launch_pbs_fft_generic(hipStream_t st, uint32_t N, uint32_t glwe_dim, const PbsArgs &a, const FftTables &tb);
void launch_pbs_ntt_generic(hipStream_t st, uint32_t N, uint32_t glwe_dim, const PbsArgs &a, const NttTables &tb);
void launch_pbs_exact_generic(hipStream_t st, uint32_t N, uint32_t glwe_dim, const PbsArgs &a);
void launch_bsk_to_fourier(hipStream_t st, uint32_t N, uint32_t glwe_dim, const uint64_t *src_dev, void *dst, size_t polys, const FftTables &tb);
// exact engine in its split-key f64 form (pbs_fft_wave.hip, LIMBS mode): 4 balanced 16-bit limbs per key word
constexpr int NTT_SPLIT_LIMBS = 4;
bool pbs_ntt_split_supported(uint32_t N, uint32_t glwe_dim, uint32_t level, uint32_t base_log);
void launch_pbs_ntt_split_wave(hipStream_t st, const PbsArgs &a, const FftTables &tb);
// polys standard-domain key polynomials (level_count = 1) -> NTT_SPLIT_LIMBS Fourier-domain limb polynomials each
void launch_bsk_to_split(hipStream_t st, uint32_t N, const uint64_t *src_dev, void *dst, size_t polys, const FftTables &tb);
void launch_bsk_to_ntt(hipStream_t st, uint32_t N, const uint64_t *src_dev, void *dst, size_t polys, const NttTables &tb);

// reference-order f64 verification engine (tfhe-fft radix-4 DIF plan, x86 conversion forms) — pbs_ref64.hip
void launch_pbs_ref64(hipStream_t st, uint32_t N, uint32_t glwe_dim, const PbsArgs &a, const RefTables &tb);
void launch_bsk_to_ref64(hipStream_t st, uint32_t N, const uint64_t *src_dev, void *dst, size_t polys, const RefTables &tb);

// throughput kernel for N=2048, k=1 (any l) — pbs_fft_wave.hip
bool pbs_fft_wave_supported(uint32_t N, uint32_t glwe_dim, uint32_t level);
void launch_pbs_fft_wave(hipStream_t st, const PbsArgs &a, const FftTables &tb);
// throughput kernel for N = 1024, k = 1 or 2 (one wave per polynomial, 512-point transforms)
bool pbs_fft_wave3_supported(uint32_t N, uint32_t glwe_dim, uint32_t level);
void launch_pbs_fft_wave3(hipStream_t st, uint32_t glwe_dim, const PbsArgs &a, const FftTables &tb);
// multi-bit PBS on the same kernel (a.grouping set; a.bsk = Fourier-domain multi-bit key)
bool pbs_multi_bit_wave_supported(uint32_t N, uint32_t glwe_dim, uint32_t level, uint32_t base_log, uint32_t grouping);
void launch_pbs_multi_bit_wave(hipStream_t st, const PbsArgs &a, const FftTables &tb);

// latency kernel for N=2048, k=1: one workgroup per LWE — pbs_fft_block.hip
bool pbs_fft_block_supported(uint32_t N, uint32_t glwe_dim, uint32_t level);
void launch_pbs_fft_block(hipStream_t st, const PbsArgs &a, const FftTables &tb, int variant);
// the same kernel running the products of the multi-bit latency path from parked keybundles (multibit.hip)
void launch_mb_accumulate_block(hipStream_t st, const PbsArgs &a, const FftTables &tb, const cplx *kb_lat, uint64_t *acc_g,
                                uint32_t gcount, uint32_t gpass, int first, int last, int slots);

// keyswitch — keyswitch.hip
// A operands of the large-batch keyswitch written ahead of it (by the bootstrap that produced its input)
struct KsDigits {
  const int8_t *aplanes;
  const int32_t *suma;
  uint32_t steps, base_log, level;
};
bool keyswitch_digits_emittable(uint32_t n_in, uint32_t base_log, uint32_t level, uint32_t *level_pad, uint32_t *steps);
void launch_keyswitch(hipStream_t st, uint64_t *lwe_out, const uint64_t *out_idx, const uint64_t *lwe_in,
                      const uint64_t *in_idx, const uint64_t *ksk, uint32_t n_in, uint32_t n_out,
                      uint32_t base_log, uint32_t level, uint32_t num_samples, const KsDigits *ready = nullptr);
void launch_keyswitch_64_32(hipStream_t st, uint32_t *lwe_out, const uint64_t *out_idx, const uint64_t *lwe_in,
                            const uint64_t *in_idx, const uint32_t *ksk, uint32_t n_in, uint32_t n_out,
                            uint32_t base_log, uint32_t level, uint32_t num_samples);

// cache of the matrix-core key layout (keyswitch.hip): drop what overlaps device memory about to be freed / written
void ksm_invalidate_range(int device, const void *p, size_t bytes);
size_t ksm_cache_entries();
extern std::atomic<bool> g_keyswitch_use_mfma;
extern std::atomic<bool> g_keyswitch_split_digits;
extern std::atomic<uint32_t> g_last_keyswitch_path;
extern std::atomic<uint32_t> g_keyswitch_kparts;
extern std::atomic<uint32_t> g_keyswitch_gemm_min;
bool stream_is_capturing(hipStream_t st);  // keyswitch.hip: the stream is recording a graph (nothing may allocate)
void ksd_release_stream(int device, hipStream_t st);  // the large-batch keyswitch's per-stream scratch
extern bool g_ntt_kernel_serial;

// small helpers — ciphertext.hip
void launch_iota_u64(hipStream_t st, uint64_t *out, uint32_t count);  // out[i] = i
void launch_modulus_switch(hipStream_t st, uint64_t *out, const uint64_t *in, uint32_t size, uint32_t log_modulus);
void launch_modulus_switch_multi_bit(hipStream_t st, uint64_t *out, const uint64_t *in, uint32_t groups, uint32_t log_modulus,
                                     uint32_t g);
void launch_centered_modulus_switch(hipStream_t st, uint64_t *out, const uint64_t *in, uint32_t lwe_dim, uint32_t log_modulus);
// false = a block size the reference does not take either (128 and 512 threads only)
bool launch_centered_modulus_switch_cooperative(hipStream_t st, uint64_t *out, const uint64_t *in, uint32_t lwe_dim, uint32_t log_modulus,
                                                uint32_t block_dim_x, uint32_t block_dim_y);
void launch_sample_extract(hipStream_t st, uint64_t *lwe_out, const uint64_t *glwe_in, const uint32_t *nth,
                           uint32_t num_nths, uint32_t lwe_per_glwe, uint32_t stored_per_glwe, uint32_t glwe_dim,
                           uint32_t N);
void launch_closest_representable(hipStream_t st, const uint64_t *in, uint64_t *out, uint32_t base_log, uint32_t level);

// multi-bit — multibit.hip
struct MultiBitArgs {
  PbsArgs pbs;             // bsk = Fourier-domain multi-bit key on the device ([group][subset][level][row][col][slot])
  uint32_t grouping_factor;
};
void launch_pbs_multi_bit(hipStream_t st, uint32_t N, uint32_t glwe_dim, const MultiBitArgs &a, const FftTables &tb,
                          uint64_t *acc_scratch);
// small batches: all keybundles of `group_chunk` groups first (one workgroup per polynomial), then the products
void launch_pbs_multi_bit_latency(hipStream_t st, uint32_t N, uint32_t glwe_dim, const MultiBitArgs &a,
                                  const FftTables &tb, cplx *kb_lat, uint32_t group_chunk, uint64_t *acc_g);

// unit-test kernels (device functions exposed for parity tests) — testhooks.hip
void launch_test_arith(hipStream_t st, uint32_t op, const uint64_t *in, uint64_t *out, uint32_t count, uint32_t p0, uint32_t p1);
void launch_test_transform(hipStream_t st, uint32_t op, uint32_t N, const void *in, void *out, uint32_t gpu_index);
// fourier.hip: the transform as launches of its own (op: 0 forward, tree order; 1 forward, natural order; 2 backward from
// natural order, unscaled; 3 negacyclic product, in1 is overwritten with its spectrum).  false: no transform of that size
bool launch_fourier(hipStream_t st, uint32_t gpu_index, int op, void *in1, const void *in2, void *out, uint32_t N, uint32_t total);

}  // namespace tfhe_hip
