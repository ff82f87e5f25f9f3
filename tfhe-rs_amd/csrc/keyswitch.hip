// keyswitch.hip — batched LWE->LWE keyswitch (big key -> small key), exact u64 arithmetic.
//
// Semantics: cc/algorithms/lwe_keyswitch.rs:137-227 — out = (0,…,0,b_in); for every input
// mask element i and level (l first): out -= digit_{i,level} * KSK[i][level][:].
// Replaces backends/tfhe-cuda-backend/cuda/src/crypto/keyswitch.cuh:200-560 (per-LWE kernel
// and the u64 "GEMM" variant).  Integer sums mod 2^64 are order-independent, so the tiling
// below is bit-exact by construction.
//
// Shape (HBM/L2-bound streaming of the 60 MB key): a workgroup owns TB samples x 256 output
// columns; KSK rows are read once per workgroup with 8-byte coalesced loads (lane = column)
// and reused across the TB samples held in registers; the signed digits of a chunk of IC mask
// elements are staged in LDS and broadcast-read.
#include "kernels.h"

namespace tfhe_hip {

constexpr int KS_TPB = 256;  // output columns per workgroup
constexpr int KS_TB = 16;    // samples per workgroup
constexpr int KS_IC = 32;    // mask elements decomposed per LDS stage
constexpr int KS_MAXL = 8;   // max levels staged (level_count <= 8 for every shortint set)

// DigitT: int32_t when base_log <= 31 (every shortint set), int64_t for wider bases
template <typename DigitT>
__global__ void __launch_bounds__(KS_TPB) keyswitch_kernel(uint64_t *lwe_out, const uint64_t *out_idx,
                                                           const uint64_t *lwe_in, const uint64_t *in_idx,
                                                           const uint64_t *ksk, uint32_t n_in, uint32_t n_out,
                                                           uint32_t base_log, uint32_t level, uint32_t num_samples) {
  HX_DYN_SMEM(smem);
  DigitT *dig = (DigitT *)smem;  // [KS_IC][level][KS_TB]
  const int tid = threadIdx.x;
  const uint32_t col = blockIdx.x * KS_TPB + tid;
  const uint32_t s0 = blockIdx.y * KS_TB;
  const uint32_t ns = (num_samples - s0 < (uint32_t)KS_TB) ? num_samples - s0 : KS_TB;
  const bool active = col <= n_out;

  uint64_t accv[KS_TB];
  HX_UNROLL
  for (int s = 0; s < KS_TB; ++s) accv[s] = 0;

  for (uint32_t i0 = 0; i0 < n_in; i0 += KS_IC) {
    const uint32_t ic = (n_in - i0 < (uint32_t)KS_IC) ? n_in - i0 : KS_IC;
    // stage digits: one (sample, mask element) pair per thread iteration
    for (uint32_t w = tid; w < (uint32_t)(KS_IC * KS_TB); w += KS_TPB) {
      const uint32_t ii = w / KS_TB, s = w - ii * KS_TB;
      uint64_t st = 0;
      const bool valid = ii < ic && s < ns;
      if (valid) {
        const uint64_t x = lwe_in[(size_t)in_idx[s0 + s] * (n_in + 1) + i0 + ii];
        st = decomp_init_state(x, base_log, level);
      }
      for (uint32_t lv = 0; lv < level; ++lv) {
        const int64_t d = valid ? decompose_one_level(base_log, st) : 0;
        dig[(ii * level + lv) * KS_TB + s] = (DigitT)d;
      }
    }
    __syncthreads();
    if (active) {
      for (uint32_t ii = 0; ii < ic; ++ii)
        for (uint32_t lv = 0; lv < level; ++lv) {
          const uint64_t w = ksk[((size_t)(i0 + ii) * level + lv) * (n_out + 1) + col];
          const DigitT *d = dig + (ii * level + lv) * KS_TB;
          HX_UNROLL
          for (int s = 0; s < KS_TB; ++s) accv[s] -= w * (uint64_t)(int64_t)d[s];
        }
    }
    __syncthreads();
  }
  if (active) {
    for (uint32_t s = 0; s < ns; ++s) {
      uint64_t v = accv[s];
      if (col == n_out) v += lwe_in[(size_t)in_idx[s0 + s] * (n_in + 1) + n_in];
      lwe_out[(size_t)out_idx[s0 + s] * (n_out + 1) + col] = v;
    }
  }
}

// Small bases (base_log + 33 + log2(n_in * level) <= 64, true for every shortint set: base_log 2..5):
// digits are shifted to d' = d + B/2 >= 0 and the two 32-bit halves of every key word are accumulated
// separately, acc_lo += d' * w_lo and acc_hi += d' * w_hi (one v_mad_u64_u32 each, no carry chains, no
// overflow: < 2^(base_log + 32) * n_in * level <= 2^64).  The shift is undone with the column sum of the
// key, which the workgroup accumulates on the fly from the rows it streams anyway:
//   sum d*w = sum d'*w - (B/2) * sum w ;   out = -(acc_lo + (acc_hi << 32)) + (B/2) * sum w  (mod 2^64)
// Same integer result as the generic kernel, bit for bit.
__global__ void __launch_bounds__(KS_TPB) keyswitch_small_base_kernel(uint64_t *lwe_out, const uint64_t *out_idx,
                                                                      const uint64_t *lwe_in, const uint64_t *in_idx,
                                                                      const uint64_t *ksk, uint32_t n_in,
                                                                      uint32_t n_out, uint32_t base_log,
                                                                      uint32_t level, uint32_t num_samples) {
  HX_DYN_SMEM(smem);
  uint32_t *dig = (uint32_t *)smem;  // [KS_IC][level][KS_TB], shifted digits
  const int tid = threadIdx.x;
  const uint32_t col = blockIdx.x * KS_TPB + tid;
  const uint32_t s0 = blockIdx.y * KS_TB;
  const uint32_t ns = (num_samples - s0 < (uint32_t)KS_TB) ? num_samples - s0 : KS_TB;
  const bool active = col <= n_out;
  const uint32_t half_b = 1u << (base_log - 1);

  uint64_t acc_lo[KS_TB], acc_hi[KS_TB], wsum = 0;
  HX_UNROLL
  for (int s = 0; s < KS_TB; ++s) acc_lo[s] = acc_hi[s] = 0;

  for (uint32_t i0 = 0; i0 < n_in; i0 += KS_IC) {
    const uint32_t ic = (n_in - i0 < (uint32_t)KS_IC) ? n_in - i0 : KS_IC;
    for (uint32_t w = tid; w < (uint32_t)(KS_IC * KS_TB); w += KS_TPB) {
      const uint32_t ii = w / KS_TB, s = w - ii * KS_TB;
      uint64_t st = 0;
      const bool valid = ii < ic && s < ns;
      if (valid) {
        const uint64_t x = lwe_in[(size_t)in_idx[s0 + s] * (n_in + 1) + i0 + ii];
        st = decomp_init_state(x, base_log, level);
      }
      for (uint32_t lv = 0; lv < level; ++lv) {
        // padding rows/samples get d' = B/2 (d = 0): they cancel against the column-sum term only if the
        // key row is also skipped, so padded mask elements are never multiplied (loop bound ic below) and
        // padded samples are never stored
        const int64_t d = valid ? decompose_one_level(base_log, st) : 0;
        dig[(ii * level + lv) * KS_TB + s] = (uint32_t)((int32_t)d + (int32_t)half_b);
      }
    }
    __syncthreads();
    if (active) {
      for (uint32_t ii = 0; ii < ic; ++ii)
        for (uint32_t lv = 0; lv < level; ++lv) {
          const uint64_t w = ksk[((size_t)(i0 + ii) * level + lv) * (n_out + 1) + col];
          const uint32_t w_lo = (uint32_t)w, w_hi = (uint32_t)(w >> 32);
          wsum += w;
          const uint32_t *d = dig + (ii * level + lv) * KS_TB;
          HX_UNROLL
          for (int s = 0; s < KS_TB; ++s) {
            acc_lo[s] += (uint64_t)d[s] * w_lo;
            acc_hi[s] += (uint64_t)d[s] * w_hi;
          }
        }
    }
    __syncthreads();
  }
  if (active) {
    const uint64_t corr = (uint64_t)half_b * wsum;
    for (uint32_t s = 0; s < ns; ++s) {
      uint64_t v = corr - (acc_lo[s] + (acc_hi[s] << 32));
      if (col == n_out) v += lwe_in[(size_t)in_idx[s0 + s] * (n_in + 1) + n_in];
      lwe_out[(size_t)out_idx[s0 + s] * (n_out + 1) + col] = v;
    }
  }
}

void launch_keyswitch(hipStream_t st, uint64_t *lwe_out, const uint64_t *out_idx, const uint64_t *lwe_in,
                      const uint64_t *in_idx, const uint64_t *ksk, uint32_t n_in, uint32_t n_out,
                      uint32_t base_log, uint32_t level, uint32_t num_samples) {
  HX_PANIC_IF_FALSE(base_log >= 1 && level >= 1 && level <= KS_MAXL && base_log * level < 64,
                    "keyswitch: unsupported decomposition (base_log=%u, level=%u)", base_log, level);
  if (num_samples == 0) return;
  const dim3 grid((n_out + 1 + KS_TPB - 1) / KS_TPB, (num_samples + KS_TB - 1) / KS_TB);
  uint32_t log_terms = 0;
  while (((uint64_t)1 << log_terms) < (uint64_t)n_in * level) ++log_terms;
  if (base_log + 33 + log_terms <= 64) {  // d' <= 2^base_log, so one spare bit
    const size_t smem = sizeof(uint32_t) * KS_IC * level * KS_TB;
    HX_LAUNCH(keyswitch_small_base_kernel, grid, dim3(KS_TPB), smem, st, lwe_out, out_idx, lwe_in, in_idx, ksk, n_in,
              n_out, base_log, level, num_samples);
  } else if (base_log <= 31) {
    const size_t smem = sizeof(int32_t) * KS_IC * level * KS_TB;
    HX_LAUNCH((keyswitch_kernel<int32_t>), grid, dim3(KS_TPB), smem, st, lwe_out, out_idx, lwe_in, in_idx, ksk, n_in,
              n_out, base_log, level, num_samples);
  } else {
    const size_t smem = sizeof(int64_t) * KS_IC * level * KS_TB;
    HX_LAUNCH((keyswitch_kernel<int64_t>), grid, dim3(KS_TPB), smem, st, lwe_out, out_idx, lwe_in, in_idx, ksk, n_in,
              n_out, base_log, level, num_samples);
  }
}

}  // namespace tfhe_hip
