// ciphertext.hip — stand-alone modulus switch / sample extraction helpers of the boundary
// (backends/tfhe-cuda-backend/cuda/include/ciphertext.h; used by tests and by callers that
// run the PBS stages separately).
#include "kernels.h"

namespace tfhe_hip {

// cc/fft_impl/common.rs:10-23 applied element-wise (cuda/src/crypto/torus.cuh:133-147)
__global__ void modulus_switch_kernel(uint64_t *out, const uint64_t *in, uint32_t size, uint32_t log_modulus) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < size) out[i] = modulus_switch(in[i], log_modulus);
}

// one LWE: mask with the plain switch, body with the centered-mean correction
// (cc/algorithms/modulus_switch.rs:35-103; cuda/src/crypto/torus.cuh:364-433)
__global__ void __launch_bounds__(256) centered_modulus_switch_kernel(uint64_t *out, const uint64_t *in,
                                                                     uint32_t lwe_dim, uint32_t log_modulus) {
  HX_DYN_SMEM(smem);
  const int tid = threadIdx.x;
  const uint32_t b = block_body_modulus_switch<256>(in, lwe_dim, log_modulus, 1, (uint64_t *)smem, tid);
  if (tid == 0) out[lwe_dim] = b;
  for (uint32_t i = tid; i < lwe_dim; i += 256) out[i] = modulus_switch(in[i], log_modulus);
}

// The same switch with the body correction reduced the way the bootstrap kernels reduce it in their prologue, in a block
// of the shape of the kernel to mimic (cuda/src/crypto/torus.cuh:404-465: 128 threads = the (64, 2) block of the
// throughput kernel, 512 = the generic one; the thread index is linearised).  Here: a block whose x extent is one
// wavefront reduces per wave and redundantly, through 128 words of LDS of its own, as pbs_fft_wave.hip's prologue does;
// any other shape goes through block_body_modulus_switch, the prologue of the block kernels.  Both are exact integer sums,
// so every shape gives the words of centered_modulus_switch_kernel.
template <int TPB, bool WAVES>
__global__ void __launch_bounds__(TPB) centered_modulus_switch_cooperative_kernel(uint64_t *out, const uint64_t *in,
                                                                                uint32_t lwe_dim, uint32_t log_modulus) {
  __shared__ uint64_t scratch[2 * TPB];
  const int tid = threadIdx.x + threadIdx.y * blockDim.x;
  uint32_t b;
  if (WAVES) {
    const int lane = tid & 63;
    uint64_t *buf64 = scratch + (tid >> 6) * 128;
    uint64_t sh = 0;
    int64_t sd = 0;
    for (uint32_t i = lane; i < lwe_dim; i += 64) {
      uint64_t h;
      int64_t dd;
      centered_ms_terms(in[i], log_modulus, h, dd);
      sh += h;
      sd += dd;
    }
    buf64[lane] = sh;
    buf64[64 + lane] = (uint64_t)sd;
    HX_WAVE_SYNC();
    uint64_t th = 0, td = 0;
    for (int l = 0; l < 64; ++l) {
      th += buf64[l];
      td += buf64[64 + l];
    }
    b = (uint32_t)modulus_switch(in[lwe_dim] + centered_ms_finish(th, (int64_t)td, log_modulus), log_modulus);
  } else {
    b = block_body_modulus_switch<TPB>(in, lwe_dim, log_modulus, 1, scratch, tid);
  }
  if (tid == 0) out[lwe_dim] = b;
  for (uint32_t i = tid; i < lwe_dim; i += TPB) out[i] = modulus_switch(in[i], log_modulus);
}

// cc/algorithms/glwe_sample_extraction.rs:89-164 ; indexing of cuda/src/crypto/ciphertext.cuh:32-54
__global__ void sample_extract_kernel(uint64_t *lwe_out, const uint64_t *glwe_in, const uint32_t *nth_array,
                                      uint32_t lwe_per_glwe, uint32_t stored_per_glwe, uint32_t glwe_dim, uint32_t N) {
  const uint32_t id = blockIdx.x;
  const size_t glwe_sz = (size_t)(glwe_dim + 1) * N, lwe_sz = (size_t)glwe_dim * N + 1;
  uint64_t *out = lwe_out + id * lwe_sz;
  const uint64_t *g = glwe_in + (size_t)(id / lwe_per_glwe) * glwe_sz;
  const uint32_t nth = nth_array[id] % stored_per_glwe;
  for (uint32_t p = 0; p < glwe_dim; ++p)
    for (uint32_t j = threadIdx.x; j < N; j += blockDim.x)
      out[(size_t)p * N + j] = (j <= nth) ? g[(size_t)p * N + nth - j] : (uint64_t)0 - g[(size_t)p * N + N + nth - j];
  if (threadIdx.x == 0) out[(size_t)glwe_dim * N] = g[(size_t)glwe_dim * N + nth];
}

// cc/commons/math/decomposition/decomposer.rs:25-50 on one value
__global__ void closest_representable_kernel(const uint64_t *in, uint64_t *out, uint32_t base_log, uint32_t level) {
  const uint32_t shift = 64 - base_log * level - 1;
  uint64_t res = in[0] >> shift;
  res += 1;
  res &= ~1ull;
  out[0] = res << shift;
}

// The multi-bit switch as a launch of its own (only the reference's noise tests call it; in production it is fused into the
// keybundle): per group of g mask words the 2^g subset degrees, subset s summing word m when bit (g - 1 - m) of s is set,
// the sum wrapping BEFORE the switch (cuda/src/crypto/torus.cuh:148-159,612-630; slot 0 = switch(0) = 0, as there)
__global__ void modulus_switch_multi_bit_kernel(uint64_t *out, const uint64_t *in, uint32_t groups, uint32_t log_modulus,
                                                uint32_t g) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= groups) return;
  const uint32_t per = 1u << g;
  const uint64_t *grp = in + (size_t)t * g;
  for (uint32_t s = 0; s < per; ++s) {
    uint64_t sum = 0;
    for (uint32_t m = 0; m < g; ++m)
      if ((s >> (g - 1 - m)) & 1) sum += grp[m];
    out[(size_t)t * per + s] = modulus_switch(sum, log_modulus);
  }
}
void launch_modulus_switch_multi_bit(hipStream_t st, uint64_t *out, const uint64_t *in, uint32_t groups, uint32_t log_modulus,
                                     uint32_t g) {
  if (!groups) return;
  HX_LAUNCH(modulus_switch_multi_bit_kernel, dim3((groups + 255) / 256), dim3(256), 0, st, out, in, groups, log_modulus, g);
}

void launch_modulus_switch(hipStream_t st, uint64_t *out, const uint64_t *in, uint32_t size, uint32_t log_modulus) {
  if (!size) return;
  HX_LAUNCH(modulus_switch_kernel, dim3((size + 255) / 256), dim3(256), 0, st, out, in, size, log_modulus);
}
void launch_centered_modulus_switch(hipStream_t st, uint64_t *out, const uint64_t *in, uint32_t lwe_dim,
                                    uint32_t log_modulus) {
  HX_LAUNCH(centered_modulus_switch_kernel, dim3(1), dim3(256), 2 * 256 * sizeof(uint64_t), st, out, in, lwe_dim,
            log_modulus);
}
bool launch_centered_modulus_switch_cooperative(hipStream_t st, uint64_t *out, const uint64_t *in, uint32_t lwe_dim,
                                                uint32_t log_modulus, uint32_t block_dim_x, uint32_t block_dim_y) {
  const dim3 block(block_dim_x, block_dim_y, 1);
  const bool waves = block_dim_x == 64;
  switch (block_dim_x * block_dim_y) {
  case 128:
    if (waves)
      HX_LAUNCH((centered_modulus_switch_cooperative_kernel<128, true>), dim3(1), block, 0, st, out, in, lwe_dim, log_modulus);
    else
      HX_LAUNCH((centered_modulus_switch_cooperative_kernel<128, false>), dim3(1), block, 0, st, out, in, lwe_dim, log_modulus);
    return true;
  case 512:
    if (waves)
      HX_LAUNCH((centered_modulus_switch_cooperative_kernel<512, true>), dim3(1), block, 0, st, out, in, lwe_dim, log_modulus);
    else
      HX_LAUNCH((centered_modulus_switch_cooperative_kernel<512, false>), dim3(1), block, 0, st, out, in, lwe_dim, log_modulus);
    return true;
  }
  return false;
}
void launch_sample_extract(hipStream_t st, uint64_t *lwe_out, const uint64_t *glwe_in, const uint32_t *nth,
                           uint32_t num_nths, uint32_t lwe_per_glwe, uint32_t stored_per_glwe, uint32_t glwe_dim,
                           uint32_t N) {
  if (!num_nths) return;
  HX_LAUNCH(sample_extract_kernel, dim3(num_nths), dim3(256), 0, st, lwe_out, glwe_in, nth, lwe_per_glwe,
            stored_per_glwe, glwe_dim, N);
}
void launch_closest_representable(hipStream_t st, const uint64_t *in, uint64_t *out, uint32_t base_log, uint32_t level) {
  HX_LAUNCH(closest_representable_kernel, dim3(1), dim3(1), 0, st, in, out, base_log, level);
}

__global__ void iota_u64_kernel(uint64_t *out, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = i;
}
void launch_iota_u64(hipStream_t st, uint64_t *out, uint32_t count) {
  if (count) HX_LAUNCH(iota_u64_kernel, dim3((count + 255) / 256), dim3(256), 0, st, out, count);
}

}  // namespace tfhe_hip
