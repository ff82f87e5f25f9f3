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
#include "arena.h"
#include <mutex>
#include <vector>

namespace tfhe_hip {

constexpr int KS_TPB = 256;  // output columns per workgroup
constexpr int KS_TB = 16;    // samples per workgroup
constexpr int KS_IC = 32;    // mask elements decomposed per LDS stage
constexpr int KS_MAXL = 8;   // max levels staged (level_count <= 8 for every shortint set)
// kernel selection (hip_backend_set_keyswitch_kernel: 0 = automatic, 1 = scalar kernels, 2 = the one-launch matrix-core
// kernel at every batch size, 3 = digit pass + GEMM from 129 LWEs on; tests and measurements)
std::atomic<bool> g_keyswitch_use_mfma{true};       // false: scalar kernels only
std::atomic<bool> g_keyswitch_split_digits{true};   // false: never the digit pass + GEMM
std::atomic<uint32_t> g_last_keyswitch_path{0};     // which path the last launch took: 0 scalar kernels, 1 one-launch
                                                    // matrix-core kernel, 2 digit pass + GEMM, 3 GEMM on emitted digits
std::atomic<uint32_t> g_keyswitch_kparts{8};        // workgroups per column tile of the one-launch kernel at small
                                                    // batches (hip_backend_set_keyswitch_kparts)

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

// u64 input, u32 key and output ("KS32", cc/algorithms/lwe_keyswitch.rs:331-447; replaces
// cuda_keyswitch_lwe_ciphertext_vector_64_32_async / cuda_keyswitch_gemm_64_32_async,
// backends/tfhe-cuda-backend/cuda/include/keyswitch/keyswitch.h:23-47).  Same tiling as keyswitch_kernel;
// the arithmetic is mod 2^32 (one full-rate v_mul_lo_u32 and a subtraction per term), the body is the
// input body rounded to 32 bits.
__global__ void __launch_bounds__(KS_TPB) keyswitch_64_32_kernel(uint32_t *lwe_out, const uint64_t *out_idx,
                                                                 const uint64_t *lwe_in, const uint64_t *in_idx,
                                                                 const uint32_t *ksk, uint32_t n_in, uint32_t n_out,
                                                                 uint32_t base_log, uint32_t level,
                                                                 uint32_t num_samples) {
  HX_DYN_SMEM(smem);
  uint32_t *dig = (uint32_t *)smem;  // [KS_IC][level][KS_TB], digits mod 2^32
  const int tid = threadIdx.x;
  const uint32_t col = blockIdx.x * KS_TPB + tid;
  const uint32_t s0 = blockIdx.y * KS_TB;
  const uint32_t ns = (num_samples - s0 < (uint32_t)KS_TB) ? num_samples - s0 : KS_TB;
  const bool active = col <= n_out;

  uint32_t accv[KS_TB];
  HX_UNROLL
  for (int s = 0; s < KS_TB; ++s) accv[s] = 0;

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
        const int64_t d = valid ? decompose_one_level(base_log, st) : 0;
        dig[(ii * level + lv) * KS_TB + s] = (uint32_t)(uint64_t)d;
      }
    }
    __syncthreads();
    if (active) {
      for (uint32_t ii = 0; ii < ic; ++ii)
        for (uint32_t lv = 0; lv < level; ++lv) {
          const uint32_t w = ksk[((size_t)(i0 + ii) * level + lv) * (n_out + 1) + col];
          const uint32_t *d = dig + (ii * level + lv) * KS_TB;
          HX_UNROLL
          for (int s = 0; s < KS_TB; ++s) accv[s] -= w * d[s];
        }
    }
    __syncthreads();
  }
  if (active) {
    for (uint32_t s = 0; s < ns; ++s) {
      uint32_t v = accv[s];
      if (col == n_out) {
        const uint64_t b = lwe_in[(size_t)in_idx[s0 + s] * (n_in + 1) + n_in];
        // closest representable on 32 bits, one level (decomposer.rs:25-50), shifted down to the output width
        v += (uint32_t)(((b >> 31) + 1) >> 1);
      }
      lwe_out[(size_t)out_idx[s0 + s] * (n_out + 1) + col] = v;
    }
  }
}

// ------------------------------------------------------------------ keyswitch on the int8 matrix cores
// The keyswitch is a GEMM: out[s][col] = -sum_k digit[s][k] * KSK[k][col], k = (mask element, level).  With
// shifted digits d' = d + B/2 in [0, B] (an i8) and the key split in its 8 byte planes, re-centred to
// b' = byte - 128 (an i8),
//   sum_k d'*w = sum_p 2^(8p) * ( sum_k d'*b'_p  +  128 * sum_k d' ),      sum_k d*w = sum_k d'*w - (B/2) * sum_k w
// and sum_k d'*b'_p is exactly what v_mfma_i32_32x32x32_i8 computes (|.| <= B * 128 * K < 2^31).  All integer,
// so the result equals the scalar kernels' bit for bit.
//   * ksk_planes_kernel: key -> [k block of 16][column tile][plane][column][16 bytes]  (+ column sums), so that a
//     lane's B operand is one 16-byte load and a wave's loads are contiguous; redone at every call (60 MB in,
//     60 MB out, ~40 us) because the C ABI hands the key over as a plain device array.
//   * ks_mfma_kernel: a wave owns 32 samples x 32 columns x 8 planes (8 accumulators of 16 VGPRs); per step of
//     32 k it decomposes 16/level mask words of its row into the A operand (no LDS: a lane's 16 consecutive k
//     ARE the digits of consecutive mask words) and issues 8 MFMAs.
constexpr int KSM_CT = 32;  // columns per tile

// level_pad >= level: the K dimension is laid out with level_pad rows per mask word (a power of two, so that a
// lane's 16 consecutive k cover whole mask words); the rows level..level_pad-1 are zero key rows
// KeyT = uint64_t: 8 byte planes; uint32_t (the 64->32 keyswitch): 4
template <typename KeyT>
__global__ void __launch_bounds__(256) ksk_planes_kernel(int8_t *planes, uint64_t *colsum, const KeyT *ksk,
                                                         uint32_t K, uint32_t ncols, uint32_t col_tiles,
                                                         uint32_t level, uint32_t level_pad) {
  constexpr int PLANES = (int)sizeof(KeyT);
  // one thread per (k block, column): 16 key words down the column
  const uint32_t col = blockIdx.x * 256 + threadIdx.x, kb = blockIdx.y;
  if (col >= col_tiles * KSM_CT) return;
  uint64_t wv[16], sum = 0;
  for (int j = 0; j < 16; ++j) {
    const uint32_t kp = kb * 16 + j, word = kp / level_pad, lv = kp - word * level_pad;
    wv[j] = (col < ncols && lv < level) ? ksk[((size_t)word * level + lv) * ncols + col] : 0;
    sum += wv[j];
  }
  if (col < ncols && sum) atomicAdd((unsigned long long *)&colsum[col], (unsigned long long)sum);
  const uint32_t ct = col / KSM_CT, cl = col % KSM_CT;
  for (int p = 0; p < PLANES; ++p) {
    uint32_t pk[4];
    for (int q = 0; q < 4; ++q) {
      uint32_t v = 0;
      for (int j = 0; j < 4; ++j) v |= (uint32_t)(uint8_t)((int)((wv[q * 4 + j] >> (8 * p)) & 0xFF) - 128) << (8 * j);
      pk[q] = v;
    }
    uint32_t *dst = (uint32_t *)(planes + ((((size_t)kb * col_tiles + ct) * PLANES + p) * KSM_CT + cl) * 16);
    dst[0] = pk[0];
    dst[1] = pk[1];
    dst[2] = pk[2];
    dst[3] = pk[3];
  }
}

// Fingerprint of a key: KSM_FP words sampled from end to end.  Taken when the matrix-core layout is built and
// compared by every keyswitch launch (one wave of workgroup (0, 0): 64 loads): a key REWRITTEN IN PLACE behind the
// library's back (a caller's own kernel, a raw hipMemcpy, a peer copy issued under another gpu_index) while its
// cached layout is still alive makes the launch trap — loud — instead of keyswitching with stale planes.  Writes
// through this library's own entry points invalidate the cache and never get here (ksm_invalidate_range).
constexpr int KSM_FP = 64;
HX_DEV size_t ksm_fp_index(int l, size_t words) {
  const size_t step = words / KSM_FP;
  return step ? (size_t)l * step + ((size_t)l * 977u) % step : (size_t)l % (words ? words : 1);
}
template <typename KeyT>
__global__ void __launch_bounds__(64) ksk_fingerprint_kernel(uint64_t *fp, const KeyT *ksk, size_t words) {
  fp[threadIdx.x] = (uint64_t)ksk[ksm_fp_index((int)threadIdx.x, words)];
}

// PADDED: `level` < LEVEL real levels per mask word, the rest zero digits against zero key rows
// OutT = uint32_t: the 64->32 keyswitch (4 planes, arithmetic mod 2^32, body rounded to 32 bits)
// KSPLIT = 4 (batches of at most 32 LWEs: one tile of samples per workgroup): the four waves of a workgroup take a
// quarter of the K dimension each for the SAME tile and add their integer accumulators through LDS — the key is
// streamed four times faster for the latency-bound rounds of the radix layer (7 to 32 blocks)
#define PLANES_OF(T) ((int)sizeof(T))
template <int LEVEL, bool PADDED, typename OutT, int KSPLIT = 1>
__global__ void __launch_bounds__(256) ks_mfma_kernel(OutT *lwe_out, const uint64_t *out_idx, const uint64_t *lwe_in,
                                                      const uint64_t *in_idx, const int8_t *planes,
                                                      const uint64_t *colsum, uint32_t n_in, uint32_t n_out,
                                                      uint32_t base_log, uint32_t num_samples, uint32_t col_tiles,
                                                      uint32_t level,  // level <= LEVEL (the padded count)
                                                      const OutT *ksk_raw, size_t ksk_words) {
  constexpr int PLANES = (int)sizeof(OutT);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < KSM_FP) {  // the planes must still be this key's
    const uint64_t *fp = colsum + (size_t)col_tiles * KSM_CT;
    if ((uint64_t)ksk_raw[ksm_fp_index((int)threadIdx.x, ksk_words)] != fp[threadIdx.x]) __builtin_trap();
  }
  __shared__ int32_t sa[4][2][32];  // per wave: sum of the shifted digits of every row, per k half
  __shared__ int32_t red[KSPLIT > 1 ? 2 : 1][KSPLIT > 1 ? PLANES_OF(OutT) : 1][KSPLIT > 1 ? 16 : 1][64];
  // KSPLIT = 4 with a 64-bit key: 64 KiB of static LDS for the two-round reduction — gfx950 only (160 KiB per CU; this
  // library is built for that target alone, Makefile ARCH), two workgroups per CU
  static_assert(sizeof(red) + sizeof(sa) <= 160 * 1024 / 2, "split-K reduction buffers: more than half of a gfx950 CU's LDS");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 31, h = lane >> 5;
  const uint32_t ct = blockIdx.x;
  // KSPLIT: one tile of samples (<= 32 LWEs); blockIdx.y = which of the gridDim.y parts of K this workgroup takes (the
  // parts add their integer shares into the zeroed output with atomics: the whole chip streams the key, not 29 CUs)
  const uint32_t stile = KSPLIT > 1 ? 0u : blockIdx.y * 4 + wave;
  // (without KSPLIT — 33 to 128 LWEs, a wave per tile — the parts are the z dimension of the grid)
  const uint32_t kparts = KSPLIT > 1 ? gridDim.y : gridDim.z, kpart = KSPLIT > 1 ? blockIdx.y : blockIdx.z;
  const uint32_t s = stile * 32 + row;
  const bool live = stile * 32 < num_samples;       // whole wave
  const uint32_t s_ld = s < num_samples ? s : 0;    // rows past the batch compute on sample 0 and store nothing
  const uint64_t *x = lwe_in + (size_t)in_idx[s_ld] * (n_in + 1);
  const uint32_t half_b = 1u << (base_log - 1);
  constexpr int WORDS = 16 / LEVEL;                 // mask words per lane per step
  hx_i32x16 acc[PLANES];
  for (int p = 0; p < PLANES; ++p)
    for (int r = 0; r < 16; ++r) acc[p].v[r] = 0;
  int32_t my_sa = 0;
  const uint32_t steps = n_in * LEVEL / 32;
  const bool narrow = base_log * level <= 30;
  if (live) {
    // mask words of step st+1 are requested before step st is decomposed (n_in + 1 words per LWE: the last
    // request of the last step reads at most the body, in range)
    // my share of the steps (all of them unless the waves split K)
    const uint32_t st_lo = KSPLIT > 1 ? (steps / (KSPLIT * kparts)) * (kpart * KSPLIT + (uint32_t)wave) : (steps / kparts) * kpart;
    const uint32_t st_hi = KSPLIT > 1 ? st_lo + steps / (KSPLIT * kparts) : st_lo + steps / kparts;
    uint64_t xn[WORDS];
    HX_UNROLL
    for (int q = 0; q < WORDS; ++q) xn[q] = x[(st_lo * 32 + h * 16) / LEVEL + q];
    for (uint32_t st = st_lo; st < st_hi; ++st) {
      // A operand: k = st*32 + h*16 + j  <->  mask word (k / LEVEL), level index (k % LEVEL), level l first
      hx_i8x16 av;
      uint32_t bytes[16];
      uint64_t xc[WORDS];
      HX_UNROLL
      for (int q = 0; q < WORDS; ++q) xc[q] = xn[q];
      if (st + 1 < st_hi) {
        const uint32_t w1 = ((st + 1) * 32 + h * 16) / LEVEL;
        HX_UNROLL
        for (int q = 0; q < WORDS; ++q) xn[q] = x[w1 + q];
      }
      // B operands of this step: one 16-byte load per plane
      const int8_t *bp = planes + ((((size_t)(st * 2 + h) * col_tiles + ct) * PLANES) * KSM_CT + row) * 16;
      hx_i8x16 bv[PLANES];
      HX_UNROLL
      for (int p = 0; p < PLANES; ++p) {
        const int32_t *src = (const int32_t *)(bp + (size_t)p * KSM_CT * 16);
        bv[p].w[0] = src[0];
        bv[p].w[1] = src[1];
        bv[p].w[2] = src[2];
        bv[p].w[3] = src[3];
      }
      HX_UNROLL
      for (int q = 0; q < WORDS; ++q) {
        const uint32_t real = PADDED ? level : (uint32_t)LEVEL;
        if (narrow) {  // wave-uniform: the decomposition on 32-bit registers (arith.h)
          int32_t state = decomp_init_state32((uint32_t)(xc[q] >> 32), base_log, real);
          HX_UNROLL
          for (int lv = 0; lv < LEVEL; ++lv) {
            // padded levels: digit 0 against a zero key row (its two shift corrections cancel)
            const int32_t d = ((!PADDED || (uint32_t)lv < level) ? decompose_one_level32(base_log, state) : 0) +
                              (int32_t)half_b;
            bytes[q * LEVEL + lv] = (uint32_t)d;
            my_sa += d;
          }
        } else {
          uint64_t state = decomp_init_state(xc[q], base_log, real);
          HX_UNROLL
          for (int lv = 0; lv < LEVEL; ++lv) {
            const int32_t d = ((!PADDED || (uint32_t)lv < level) ? (int32_t)decompose_one_level(base_log, state) : 0) +
                              (int32_t)half_b;
            bytes[q * LEVEL + lv] = (uint32_t)d;
            my_sa += d;
          }
        }
      }
      HX_UNROLL
      for (int q = 0; q < 4; ++q)
        av.w[q] = (int32_t)(bytes[4 * q] | (bytes[4 * q + 1] << 8) | (bytes[4 * q + 2] << 16) | (bytes[4 * q + 3] << 24));
      HX_UNROLL
      for (int p = 0; p < PLANES; ++p) acc[p] = hx_mfma_i32_32x32x32_i8(av, bv[p], acc[p]);
    }
  }
  sa[wave][h][row] = my_sa;
  if constexpr (KSPLIT > 1) {  // waves 2, 3 -> 0, 1; then wave 1 -> 0 (integer sums: exact in any order)
    HX_UNROLL
    for (int round = 0; round < 2; ++round) {
      const int senders_from = round == 0 ? 2 : 1, senders_to = round == 0 ? 4 : 2;
      if (wave >= senders_from && wave < senders_to) {
        HX_UNROLL
        for (int p = 0; p < PLANES; ++p)
          HX_UNROLL
          for (int r = 0; r < 16; ++r) red[wave - senders_from][p][r][lane] = acc[p].v[r];
      }
      __syncthreads();
      if (wave < senders_to - senders_from) {  // round 0: waves 0, 1 take 2, 3; round 1: wave 0 takes 1
        HX_UNROLL
        for (int p = 0; p < PLANES; ++p)
          HX_UNROLL
          for (int r = 0; r < 16; ++r) acc[p].v[r] += red[wave][p][r][lane];
      }
      __syncthreads();
    }
    if (wave != 0) return;
  } else {
    __syncthreads();
  }
  if (!live) return;
  const uint32_t col = ct * KSM_CT + (lane & 31);
  if (col > n_out) return;
  const uint64_t corr = (uint64_t)half_b * colsum[col];
  HX_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int orow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const uint32_t so = stile * 32 + orow;
    if (so >= num_samples) continue;
    int64_t sum_a = (int64_t)sa[wave][0][orow] + sa[wave][1][orow];
    if constexpr (KSPLIT > 1) {
      HX_UNROLL
      for (int ww = 1; ww < KSPLIT; ++ww) sum_a += (int64_t)sa[ww][0][orow] + sa[ww][1][orow];
    }
    uint64_t v = 0;
    HX_UNROLL
    for (int p = 0; p < PLANES; ++p) v += (uint64_t)((int64_t)acc[p].v[r] + 128 * sum_a) << (8 * p);
    uint64_t o = (kpart == 0 ? corr : 0) - v;  // the part's share: wrapping sums, any order
    if (col == n_out && kpart == 0) {
      const uint64_t b = lwe_in[(size_t)in_idx[so] * (n_in + 1) + n_in];
      // 32-bit output: the body rounded to the closest multiple of 2^32, as keyswitch_64_32_kernel does
      o += sizeof(OutT) == 8 ? b : ((b >> 31) + 1) >> 1;
    }
    OutT *dst = &lwe_out[(size_t)out_idx[so] * (n_out + 1) + col];
    if (kparts > 1) {
      if constexpr (sizeof(OutT) == 8) atomicAdd((unsigned long long *)dst, (unsigned long long)o);
      else atomicAdd((unsigned int *)dst, (unsigned int)o);
    } else {
      *dst = (OutT)o;
    }
  }
}

// zeroes the output ciphertexts of a keyswitch whose K dimension is split over workgroups (ks_mfma_kernel, KSPLIT)
template <typename OutT>
__global__ void ks_zero_outputs_kernel(OutT *lwe_out, const uint64_t *out_idx, uint32_t n_out, uint32_t num_samples) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
  if (col <= n_out && s < num_samples) lwe_out[(size_t)out_idx[s] * (n_out + 1) + col] = 0;
}

// ------------------------------------------------------------------ large batches: digits once, then a plain int8 GEMM
// ks_mfma_kernel above rebuilds the shifted digits of its 32 samples for every one of the col_tiles column tiles
// (about 165 vector instructions per step of 8 matrix instructions) and every wave fetches its own copy of the B
// operand through the CU's vector L1 (8 KB per wave and step against 64 B/clk: twice what the matrix pipe can
// consume).  Beyond KSD_MIN_SAMPLES LWEs the work is split:
//   * ks_digits_kernel: one workgroup per tile of 32 samples decomposes the tile ONCE into the A operands of the
//     GEMM, laid out as the matrix instruction wants them — [tile][step][lane][16 bytes], lane = (k half, row),
//     byte j <-> k = step*32 + half*16 + j — plus the per-sample sum of the shifted digits (the shift correction);
//   * ks_gemm_kernel: 4 waves x 32 samples against one column tile; per step of 32 k the workgroup stages the 8 KB
//     of B (8 byte planes x 32 columns x 32 k) in LDS ONCE (double buffered; round 4: two steps — four with u32 keys —
//     per workgroup barrier, which pays the barrier, the drain of the global -> LDS loads in front of it and the LDS
//     round trip behind it once per 16 matrix instructions: 0.373 -> 0.307 ms per 4096 on one box), every wave
//     reads its operands from there and its A operand as one coalesced 16-byte load: no vector arithmetic in the
//     loop, L1 traffic a quarter.  Same integer sums as ks_mfma_kernel: identical bits.
// below: ks_mfma_kernel (one launch; K shared by the waves of a workgroup up to 32 LWEs and by up to 8 workgroups per
// column tile).  Measured at the 2_2 sizes (tools/measure_all.py kscross), one launch / digit pass + GEMM: 256 LWEs 0.063 /
// 0.148 ms, 512: 0.114 / 0.152, 1024: 0.214 / 0.162, 4096: 0.82 / 0.38
constexpr uint32_t KSD_MIN_SAMPLES = 769;
std::atomic<uint32_t> g_keyswitch_gemm_min{KSD_MIN_SAMPLES};  // hip_backend_set_keyswitch_kernel(3): 129 (tests)

template <int LEVEL, bool PADDED>
HX_DEV hx_i8x16 ksm_build_a(const uint64_t (&xc)[16 / LEVEL], uint32_t base_log, uint32_t level, bool narrow,
                            uint32_t half_b, int32_t &my_sa) {
  constexpr int WORDS = 16 / LEVEL;
  uint32_t bytes[16];
  HX_UNROLL
  for (int q = 0; q < WORDS; ++q) {
    const uint32_t real = PADDED ? level : (uint32_t)LEVEL;
    if (narrow) {  // wave-uniform: the decomposition on 32-bit registers (arith.h)
      int32_t state = decomp_init_state32((uint32_t)(xc[q] >> 32), base_log, real);
      HX_UNROLL
      for (int lv = 0; lv < LEVEL; ++lv) {
        const int32_t d = ((!PADDED || (uint32_t)lv < level) ? decompose_one_level32(base_log, state) : 0) + (int32_t)half_b;
        bytes[q * LEVEL + lv] = (uint32_t)d;
        my_sa += d;
      }
    } else {
      uint64_t state = decomp_init_state(xc[q], base_log, real);
      HX_UNROLL
      for (int lv = 0; lv < LEVEL; ++lv) {
        const int32_t d = ((!PADDED || (uint32_t)lv < level) ? (int32_t)decompose_one_level(base_log, state) : 0) + (int32_t)half_b;
        bytes[q * LEVEL + lv] = (uint32_t)d;
        my_sa += d;
      }
    }
  }
  hx_i8x16 av;
  HX_UNROLL
  for (int q = 0; q < 4; ++q)
    av.w[q] = (int32_t)(bytes[4 * q] | (bytes[4 * q + 1] << 8) | (bytes[4 * q + 2] << 16) | (bytes[4 * q + 3] << 24));
  return av;
}

// grid: (tiles of 32 samples, KSD_SPLIT): workgroup (t, y) takes the steps y*4 + w, y*4 + w + 4*KSD_SPLIT, ... of tile
// t (w = its wave) and adds its share of the digit sums to suma (zeroed by the launcher; integer sums, any order)
constexpr int KSD_SPLIT = 4;
template <int LEVEL, bool PADDED>
__global__ void __launch_bounds__(256) ks_digits_kernel(int8_t *aplanes, int32_t *suma, const uint64_t *lwe_in,
                                                        const uint64_t *in_idx, uint32_t n_in, uint32_t base_log,
                                                        uint32_t level, uint32_t num_samples) {
  __shared__ int32_t sa[4][2][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 31, h = lane >> 5;
  const uint32_t stile = blockIdx.x, s = stile * 32 + row;
  const uint32_t s_ld = s < num_samples ? s : 0;  // rows past the batch decompose sample 0; the GEMM stores nothing for them
  const uint64_t *x = lwe_in + (size_t)in_idx[s_ld] * (n_in + 1);
  const uint32_t half_b = 1u << (base_log - 1), steps = n_in * LEVEL / 32;
  const bool narrow = base_log * level <= 30;
  constexpr int WORDS = 16 / LEVEL;
  int32_t my_sa = 0;
  int8_t *dst = aplanes + ((size_t)stile * steps) * 1024 + (size_t)lane * 16;
  for (uint32_t st = blockIdx.y * 4 + (uint32_t)wave; st < steps; st += 4 * KSD_SPLIT) {
    uint64_t xc[WORDS];
    HX_UNROLL
    for (int q = 0; q < WORDS; ++q) xc[q] = x[(st * 32 + h * 16) / LEVEL + q];
    const hx_i8x16 av = ksm_build_a<LEVEL, PADDED>(xc, base_log, level, narrow, half_b, my_sa);
    int32_t *o = (int32_t *)(dst + (size_t)st * 1024);
    o[0] = av.w[0];
    o[1] = av.w[1];
    o[2] = av.w[2];
    o[3] = av.w[3];
  }
  sa[wave][h][row] = my_sa;
  __syncthreads();
  if (threadIdx.x < 32) {
    int32_t t = 0;
    for (int w = 0; w < 4; ++w) t += sa[w][0][threadIdx.x] + sa[w][1][threadIdx.x];
    atomicAdd(&suma[stile * 32 + threadIdx.x], t);
  }
}

// steps of 32 k between two workgroup barriers of ks_gemm_kernel, by key word size: measured per 4096 LWEs on one box,
// u64 keys 0.373 / 0.307 / 0.325 ms with 1 / 2 / 4 (4 fills the 64 KB of static LDS), u32 keys 0.279 / 0.216 / 0.203 ms
#ifndef KSG_STEPS_PER_BARRIER_U64
#define KSG_STEPS_PER_BARRIER_U64 2
#endif
#ifndef KSG_STEPS_PER_BARRIER_U32
#define KSG_STEPS_PER_BARRIER_U32 4
#endif
template <typename OutT, int SPB>
__global__ void ks_gemm_kernel(OutT *lwe_out, const uint64_t *out_idx, const uint64_t *lwe_in, const uint64_t *in_idx,
                               const int8_t *planes, const uint64_t *colsum, const int8_t *aplanes, const int32_t *suma,
                               uint32_t n_in, uint32_t n_out, uint32_t base_log, uint32_t num_samples, uint32_t col_tiles,
                               uint32_t steps, const OutT *ksk_raw, size_t ksk_words);
// the widest supported step count per barrier that divides `steps`
template <typename OutT, typename... Args>
static void launch_ks_gemm(dim3 grid, hipStream_t st, uint32_t steps, Args... args) {
  constexpr int WANT = sizeof(OutT) == 8 ? KSG_STEPS_PER_BARRIER_U64 : KSG_STEPS_PER_BARRIER_U32;
  if (WANT >= 4 && steps % 4 == 0) HX_LAUNCH((ks_gemm_kernel<OutT, 4>), grid, dim3(256), 0, st, args...);
  else if (WANT >= 2 && steps % 2 == 0) HX_LAUNCH((ks_gemm_kernel<OutT, 2>), grid, dim3(256), 0, st, args...);
  else HX_LAUNCH((ks_gemm_kernel<OutT, 1>), grid, dim3(256), 0, st, args...);
}
// SPB = steps of 32 k per workgroup barrier: with 2 the barrier, the drain of the global -> LDS loads in front of it and
// the LDS round trip behind it are paid once per 16 matrix instructions instead of once per 8 (32 KB of LDS for u64 keys)
#ifndef KS_GEMM_MIN_WAVES
#define KS_GEMM_MIN_WAVES 2  // waves per SIMD the register budget is cut for (2: 256 registers; the u64 forms spill 40-61 in their epilogue)
#endif
template <typename OutT, int SPB>
__global__ void __launch_bounds__(256, KS_GEMM_MIN_WAVES) ks_gemm_kernel(OutT *lwe_out, const uint64_t *out_idx, const uint64_t *lwe_in,
                                                      const uint64_t *in_idx, const int8_t *planes,
                                                      const uint64_t *colsum, const int8_t *aplanes,
                                                      const int32_t *suma, uint32_t n_in, uint32_t n_out,
                                                      uint32_t base_log, uint32_t num_samples, uint32_t col_tiles,
                                                      uint32_t steps, const OutT *ksk_raw, size_t ksk_words) {
  constexpr int PLANES = (int)sizeof(OutT);
  constexpr int HALF_BYTES = PLANES * KSM_CT * 16;  // one k half of a step: [plane][column][16 bytes]
  constexpr int CHUNKS = 2 * HALF_BYTES / 16 / 256; // 16-byte chunks of B per thread and step (2 for u64 keys, 1 for u32)
  __shared__ alignas(16) int8_t bs[2][SPB][2][HALF_BYTES];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < KSM_FP) {  // the planes must still be this key's
    const uint64_t *fp = colsum + (size_t)col_tiles * KSM_CT;
    if ((uint64_t)ksk_raw[ksm_fp_index((int)threadIdx.x, ksk_words)] != fp[threadIdx.x]) __builtin_trap();
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = lane & 31, h = lane >> 5;
  const uint32_t ct = blockIdx.x, stile = blockIdx.y * 4 + wave;
  const bool live = stile * 32 < num_samples;  // whole wave; dead waves still stage B and meet the barriers
  hx_i32x16 acc[PLANES];
  for (int p = 0; p < PLANES; ++p)
    for (int r = 0; r < 16; ++r) acc[p].v[r] = 0;
  // B: 16-byte chunk c = tid + q*256 of step st -> k half c / (HALF_BYTES/16), byte offset (c % (HALF_BYTES/16)) * 16.
  // A wave's 64 chunks are 1 KiB contiguous in the key layout AND in the LDS image: one direct global -> LDS load
  // each (no staging registers, no ds_write pass).
  const int8_t *bsrc[CHUNKS];
  int8_t *bdst[CHUNKS];  // wave-uniform: where lane 0's chunk goes
  HX_UNROLL
  for (int q = 0; q < CHUNKS; ++q) {
    const int c = tid + q * 256, half = c / (HALF_BYTES / 16), off = (c % (HALF_BYTES / 16)) * 16;
    bsrc[q] = planes + ((size_t)half * col_tiles + ct) * HALF_BYTES + off;  // + st * 2 * col_tiles * HALF_BYTES per step
    const int c0 = wave * 64 + q * 256, half0 = c0 / (HALF_BYTES / 16), off0 = (c0 % (HALF_BYTES / 16)) * 16;
    bdst[q] = &bs[0][0][half0][off0];
  }
  const size_t bstep = (size_t)2 * col_tiles * HALF_BYTES;
  const hx_i8x16 *ap = (const hx_i8x16 *)(aplanes + ((size_t)(live ? stile : 0) * steps) * 1024) + lane;
  // macro step ms = steps ms * SPB .. ms * SPB + SPB - 1
  auto stage_b = [&](uint32_t ms, int buf) {
    HX_UNROLL
    for (int u = 0; u < SPB; ++u)
      HX_UNROLL
      for (int q = 0; q < CHUNKS; ++q)
        HX_GLOBAL_TO_LDS16(bsrc[q] + (size_t)(ms * SPB + u) * bstep, bdst[q] + (buf * SPB + u) * (2 * HALF_BYTES), lane);
  };
  // Software pipeline: at the top of a macro step the B images of the next one are requested into the other LDS buffer
  // and its A operands into registers; both land under the matrix instructions of this one.  Branch-free (a conditional
  // around the matrix instructions makes the compiler shuttle the 128 accumulator registers between the register files
  // every step): the last macro step re-requests its own operands, waves past the batch multiply the first tile's digits
  // and store nothing.
  const uint32_t msteps = steps / SPB, last = msteps - 1;
  hx_i8x16 av[SPB], a1[SPB];
  stage_b(0, 0);
  HX_UNROLL
  for (int u = 0; u < SPB; ++u) av[u] = ap[(size_t)u * 64];
  __syncthreads();
  for (uint32_t ms = 0; ms < msteps; ++ms) {
    const int cur = (int)(ms & 1);
    const uint32_t nx = ms + 1 < msteps ? ms + 1 : last;
    HX_UNROLL
    for (int u = 0; u < SPB; ++u) {
      // this step's operands out of LDS first: the compiler drains every outstanding global -> LDS load before an LDS
      // read, so the requests for the next macro step go out behind the LAST step's reads (u == SPB - 1) and land under the
      // matrix instructions of that step and the barrier
      hx_i8x16 bv[PLANES];
      HX_UNROLL
      for (int p = 0; p < PLANES; ++p) bv[p] = *(const hx_i8x16 *)&bs[cur][u][h][p * (KSM_CT * 16) + row * 16];
      HX_SCHED_FENCE();
      if (u == SPB - 1) {
#ifndef KSG_SKIP_B  // (timing experiments: wrong results)
        stage_b(nx, cur ^ 1);
#endif
#ifndef KSG_SKIP_A
        HX_UNROLL
        for (int v = 0; v < SPB; ++v) a1[v] = ap[((size_t)nx * SPB + v) * 64];
#endif
      }
      HX_SCHED_FENCE();
#ifndef KSG_SKIP_MFMA
      HX_UNROLL
      for (int p = 0; p < PLANES; ++p) acc[p] = hx_mfma_i32_32x32x32_i8(av[u], bv[p], acc[p]);
#else
      HX_UNROLL
      for (int p = 0; p < PLANES; ++p) acc[p].v[0] += bv[p].w[0] ^ av[u].w[0];
#endif
      HX_SCHED_FENCE();
    }
    HX_UNROLL
    for (int u = 0; u < SPB; ++u) av[u] = a1[u];
    __syncthreads();
  }
  if (!live) return;
  // everything the epilogue addresses derives from this lane value, made opaque HERE: otherwise the index, sum and
  // pointer loads of all 16 output rows are hoisted above the loop and held (or spilled) across it
  int lane_e = lane;
  HX_OPAQUE(lane_e);
  const uint32_t col = ct * KSM_CT + (lane_e & 31);
  if (col > n_out) return;
  const uint32_t half_b = 1u << (base_log - 1);
  const uint64_t corr = (uint64_t)half_b * colsum[col];
  HX_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int orow = (r & 3) + 8 * (r >> 2) + 4 * (lane_e >> 5);
    const uint32_t so = stile * 32 + orow;
    if (so >= num_samples) continue;
    const int64_t sum_a = (int64_t)suma[so];
    uint64_t v = 0;
    HX_UNROLL
    for (int p = 0; p < PLANES; ++p) v += (uint64_t)((int64_t)acc[p].v[r] + 128 * sum_a) << (8 * p);
    uint64_t o = corr - v;
    if (col == n_out) {
      const uint64_t b = lwe_in[(size_t)in_idx[so] * (n_in + 1) + n_in];
      o += sizeof(OutT) == 8 ? b : ((b >> 31) + 1) >> 1;
    }
    lwe_out[(size_t)out_idx[so] * (n_out + 1) + col] = (OutT)o;
  }
}

// A operands + digit sums of the large-batch path: buffers per (device, stream) — the keyswitch entry points of the
// reference carry no scratch argument.  Two buffers, so that nothing a captured graph refers to is ever freed, replaced
// or shared with live launches:
//   * `live`  serves launches outside stream capture; it grows by allocate-new / synchronise the stream / free-old;
//   * `cap`   serves launches recorded under stream capture.  Capture allocates nothing: the first captured launch that
//             finds a large enough `live` buffer takes it over (it becomes `cap`, later plain launches allocate a new
//             `live`); a captured launch that finds neither takes the one-launch kernel.  `cap` is only released with
//             the stream (cuda_destroy_stream), after every graph instantiated from the capture must be gone
//             (graphs).
struct KsdScratch {
  int device;
  hipStream_t stream;
  void *live;
  size_t live_bytes;
  void *cap;
  size_t cap_bytes;
};
static std::vector<KsdScratch> g_ksd;
static std::mutex g_ksd_mutex;
static void *ksd_scratch(int device, hipStream_t st, size_t bytes, bool capturing) {
  void *old = nullptr, *buf = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_ksd_mutex);
    KsdScratch *e = nullptr;
    for (KsdScratch &k : g_ksd)
      if (k.device == device && k.stream == st) e = &k;
    if (capturing) {
      if (e == nullptr) return nullptr;
      if (e->cap != nullptr) return e->cap_bytes >= bytes ? e->cap : nullptr;
      if (e->live == nullptr || e->live_bytes < bytes) return nullptr;
      e->cap = e->live;  // from now on this buffer belongs to the graphs captured on this stream
      e->cap_bytes = e->live_bytes;
      e->live = nullptr;
      e->live_bytes = 0;
      return e->cap;
    }
    if (e != nullptr && e->live_bytes >= bytes) return e->live;
    if (e == nullptr) {
      g_ksd.push_back(KsdScratch{device, st, nullptr, 0, nullptr, 0});
      e = &g_ksd.back();
    }
    old = e->live;
    buf = device_alloc_sync(bytes);
    e->live = buf;
    e->live_bytes = bytes;
  }
  if (old) {
    HX_CHECK(hipStreamSynchronize(st));  // launches that still read the old buffer (never a captured one: see above)
    device_free_sync(old);
  }
  return buf;
}
void ksd_release_stream(int device, hipStream_t st) {
  void *live = nullptr, *cap = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_ksd_mutex);
    for (size_t i = 0; i < g_ksd.size(); ++i)
      if (g_ksd[i].device == device && g_ksd[i].stream == st) {
        live = g_ksd[i].live;
        cap = g_ksd[i].cap;
        g_ksd.erase(g_ksd.begin() + i);
        break;
      }
  }
  if (live) device_free_sync(live);
  if (cap) device_free_sync(cap);
}

// Byte planes + column sums of a keyswitch key, built ONCE per key and kept until the key's device memory is
// released or overwritten.  The C ABI hands the key over as a plain device array at every call, so the cache is
// keyed by (device, key pointer, shape); every entry point of the boundary that frees or writes device memory
// (cuda_drop, cuda_memcpy_*_to_gpu / gpu_to_gpu, cuda_memset_async) calls ksm_invalidate_range first, so a
// recycled address can never serve stale planes.  A steady-state keyswitch call therefore allocates nothing and
// launches one kernel (graph-capture safe); only the first call with a new key allocates and lays the key out.
struct KsmEntry {
  int device;
  const void *ksk;
  size_t ksk_bytes;
  uint32_t n_in, n_out, level, level_pad, key_size;
  void *planes;      // [K/16][col tile][plane][col][16 B], then the column sums
  size_t plane_bytes;
  hipEvent_t ready;  // recorded behind the layout kernel: streams other than the building one wait for it
  hipStream_t builder;  // the stream the layout kernel ran on (its later work is ordered behind it anyway)
  bool done;            // the layout kernel is known to have completed: no more waits
  uint64_t last_use;
};
static std::vector<KsmEntry> g_ksm_cache;
static std::mutex g_ksm_mutex;
static uint64_t g_ksm_tick = 0;
constexpr size_t kKsmMaxEntries = 32;

// hipFree synchronises the device: never under g_ksm_mutex (other host threads launch keyswitches meanwhile) —
// entries are unlinked under the lock and released after it
static void ksm_release(const KsmEntry &e) {
  int cur = 0;
  HX_CHECK(hipGetDevice(&cur));
  HX_CHECK(hipSetDevice(e.device));
  device_free_sync(e.planes);  // synchronises the device: no kernel still reads the planes
  HX_CHECK(hipEventDestroy(e.ready));
  HX_CHECK(hipSetDevice(cur));
}

bool stream_is_capturing(hipStream_t st) {
#if defined(TFHE_HIPEMU)
  (void)st;
  return false;
#else
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &status) != hipSuccess) return false;
  return status != hipStreamCaptureStatusNone;
#endif
}

// device memory [p, p + bytes) of `device` is about to be freed or written
void ksm_invalidate_range(int device, const void *p, size_t bytes) {
  if (p == nullptr) return;
  std::vector<KsmEntry> dead;
  {
    std::lock_guard<std::mutex> lock(g_ksm_mutex);
    const char *lo = (const char *)p, *hi = lo + (bytes ? bytes : 1);
    for (size_t i = 0; i < g_ksm_cache.size();) {
      KsmEntry &e = g_ksm_cache[i];
      const char *klo = (const char *)e.ksk, *khi = klo + e.ksk_bytes;
      if (e.device == device && klo < hi && lo < khi) {
        dead.push_back(e);
        g_ksm_cache.erase(g_ksm_cache.begin() + i);
      } else {
        ++i;
      }
    }
  }
  for (const KsmEntry &e : dead) ksm_release(e);
}
size_t ksm_cache_entries() {
  std::lock_guard<std::mutex> lock(g_ksm_mutex);
  return g_ksm_cache.size();
}

template <typename OutT>
static bool keyswitch_mfma(hipStream_t st, OutT *lwe_out, const uint64_t *out_idx, const uint64_t *lwe_in,
                           const uint64_t *in_idx, const OutT *ksk, uint32_t n_in, uint32_t n_out,
                           uint32_t base_log, uint32_t level, uint32_t num_samples, const KsDigits *ready = nullptr) {
  uint32_t level_pad = 1;  // levels per mask word in the K dimension: the next power of two
  while (level_pad < level) level_pad <<= 1;
  const uint32_t K = n_in * level_pad;
  uint32_t log_k = 0;
  while (((uint64_t)1 << log_k) < K) ++log_k;
  // any batch size: a lone ciphertext occupies one row of one 32-row tile, and the launch still streams the key
  // through col_tiles workgroups (0.17 ms at the 2_2 sizes; the scalar kernels need 2.8 ms below 64 LWEs, 4 workgroups)
  if (level_pad > 16 || K % 32 != 0 || base_log > 6 || base_log + 7 + log_k > 31) return false;
  const uint32_t ncols = n_out + 1, col_tiles = (ncols + KSM_CT - 1) / KSM_CT;
  const size_t plane_bytes = (size_t)(K / 16) * col_tiles * sizeof(OutT) * KSM_CT * 16;
  const size_t need = plane_bytes + ((size_t)col_tiles * KSM_CT + KSM_FP) * sizeof(uint64_t);  // planes, column sums, fingerprint
  const size_t ksk_words = (size_t)n_in * level * ncols;
  int device = 0;
  HX_CHECK(hipGetDevice(&device));
  int8_t *planes = nullptr;
  uint64_t *colsum = nullptr;
  // Under stream capture the cache is read-only: building an entry allocates and frees (neither may be captured) and
  // would record `ready` inside the capture, where later launches of other streams could not wait for it.  A key
  // that is not warm yet takes the scalar kernel for the captured launch (same bits).
  const bool capturing = stream_is_capturing(st);
  std::vector<KsmEntry> evicted;
  {
    std::lock_guard<std::mutex> lock(g_ksm_mutex);
    KsmEntry *hit = nullptr;
    for (KsmEntry &e : g_ksm_cache)
      if (e.device == device && e.ksk == (const void *)ksk && e.n_in == n_in && e.n_out == n_out && e.level == level &&
          e.key_size == sizeof(OutT)) {
        hit = &e;
        break;
      }
    if (hit == nullptr && capturing) return false;
    if (hit == nullptr) {
      if (g_ksm_cache.size() >= kKsmMaxEntries) {  // least recently used key goes (released after the lock)
        size_t lru = 0;
        for (size_t i = 1; i < g_ksm_cache.size(); ++i)
          if (g_ksm_cache[i].last_use < g_ksm_cache[lru].last_use) lru = i;
        evicted.push_back(g_ksm_cache[lru]);
        g_ksm_cache.erase(g_ksm_cache.begin() + lru);
      }
      KsmEntry e{};
      e.device = device;
      e.ksk = ksk;
      e.ksk_bytes = (size_t)n_in * level * ncols * sizeof(OutT);
      e.n_in = n_in, e.n_out = n_out, e.level = level, e.level_pad = level_pad, e.key_size = sizeof(OutT);
      e.plane_bytes = plane_bytes;
      e.planes = device_alloc_sync(need);
      HX_CHECK(hipEventCreate(&e.ready));
      uint64_t *cs = (uint64_t *)((char *)e.planes + plane_bytes);
      HX_CHECK(hipMemsetAsync(cs, 0, (size_t)col_tiles * KSM_CT * sizeof(uint64_t), st));
      HX_LAUNCH((ksk_planes_kernel<OutT>), dim3((col_tiles * KSM_CT + 255) / 256, K / 16), dim3(256), 0, st,
                (int8_t *)e.planes, cs, ksk, K, ncols, col_tiles, level, level_pad);
      HX_LAUNCH((ksk_fingerprint_kernel<OutT>), dim3(1), dim3(KSM_FP), 0, st, cs + (size_t)col_tiles * KSM_CT, ksk,
                ksk_words);
      HX_CHECK(hipEventRecord(e.ready, st));
      e.builder = st;
      e.done = false;
      g_ksm_cache.push_back(e);
      hit = &g_ksm_cache.back();
    } else if (!hit->done && st != hit->builder) {
      // another stream: behind the layout kernel unless it is known to be over (a query, so that a launch under
      // stream capture does not pick up an event from outside the capture once the key is warm)
      if (hipEventQuery(hit->ready) == hipSuccess) hit->done = true;
      else if (capturing) return false;  // an event from outside the capture cannot be waited for inside it
      else HX_CHECK(hipStreamWaitEvent(st, hit->ready, 0));
    }
    hit->last_use = ++g_ksm_tick;
    planes = (int8_t *)hit->planes;
    colsum = (uint64_t *)((char *)hit->planes + hit->plane_bytes);
  }
  for (const KsmEntry &e : evicted) ksm_release(e);
  if (ready != nullptr && num_samples >= g_keyswitch_gemm_min.load() && g_keyswitch_split_digits.load()) {
    // the A operands were written by the bootstrap that produced lwe_in (PbsArgs::emit_a): the GEMM alone
    HX_PANIC_IF_FALSE(ready->steps == K / 32 && ready->base_log == base_log && ready->level == level,
                      "keyswitch: the digits at hand were made for another decomposition");
    const uint32_t tiles = (num_samples + 31) / 32;
    launch_ks_gemm<OutT>(dim3(col_tiles, (tiles + 3) / 4), st, K / 32, lwe_out, out_idx, lwe_in, in_idx, (const int8_t *)planes,
                         (const uint64_t *)colsum, (const int8_t *)ready->aplanes, (const int32_t *)ready->suma, n_in, n_out,
                         base_log, num_samples, col_tiles, K / 32, ksk, ksk_words);
    g_last_keyswitch_path.store(3);
    return true;
  }
  if (num_samples >= g_keyswitch_gemm_min.load() && g_keyswitch_split_digits.load()) {
    // large batch: the digits once (A operands in the stream's scratch), then the LDS-staged GEMM
    const uint32_t tiles = (num_samples + 31) / 32, steps = K / 32;
    const size_t a_bytes = (size_t)tiles * steps * 1024, need_s = a_bytes + (size_t)tiles * 32 * sizeof(int32_t);
    int8_t *scr = (int8_t *)ksd_scratch(device, st, need_s, capturing);
    if (scr != nullptr) {
      int32_t *suma = (int32_t *)(scr + a_bytes);
      HX_CHECK(hipMemsetAsync(suma, 0, (size_t)tiles * 32 * sizeof(int32_t), st));
#define KSD_LAUNCH(L)                                                                                              \
  do {                                                                                                               \
    if (level == L)                                                                                                  \
      HX_LAUNCH((ks_digits_kernel<L, false>), dim3(tiles, KSD_SPLIT), dim3(256), 0, st, scr, suma, lwe_in, in_idx, n_in, \
                base_log, level, num_samples);                                                                       \
    else                                                                                                             \
      HX_LAUNCH((ks_digits_kernel<L, true>), dim3(tiles, KSD_SPLIT), dim3(256), 0, st, scr, suma, lwe_in, in_idx, n_in,  \
                base_log, level, num_samples);                                                                       \
  } while (0)
      switch (level_pad) {
        case 1: KSD_LAUNCH(1); break;
        case 2: KSD_LAUNCH(2); break;
        case 4: KSD_LAUNCH(4); break;
        case 8: KSD_LAUNCH(8); break;
        default: KSD_LAUNCH(16); break;
      }
#undef KSD_LAUNCH
      launch_ks_gemm<OutT>(dim3(col_tiles, (tiles + 3) / 4), st, steps, lwe_out, out_idx, lwe_in, in_idx, (const int8_t *)planes,
                           (const uint64_t *)colsum, (const int8_t *)scr, (const int32_t *)suma, n_in, n_out, base_log,
                           num_samples, col_tiles, steps, ksk, ksk_words);
      g_last_keyswitch_path.store(2);
      return true;
    }
  }
  // up to 32 LWEs: one tile of samples, the four waves of a workgroup split K (needs steps = K / 32 divisible by 4)
  const bool split = num_samples <= 32 && (K / 32) % 4 == 0;
  // ... and K over KS_KPARTS workgroups per column tile when it divides: 29 workgroups alone stream the key at what
  // 29 CUs can pull (0.09 ms for the 121 MB of a 5-level key padded to 8; 0.052 ms at the 2_2 sizes), 232 at what the
  // memory system delivers (0.027-0.032 / 0.019-0.025 ms; measured 4 parts 0.032 / 0.022, 16 parts 0.034 / 0.026)
  // (33 .. 128 LWEs: one workgroup row of up to four tiles; the same sharing of K, over the grid's z dimension)
  // beyond 128 LWEs the grid has several rows of four tiles and proportionally fewer parts
  uint32_t want = g_keyswitch_kparts.load();
  for (uint32_t rows = (num_samples + 127) / 128; rows > 1 && want > 1; rows >>= 1) want >>= 1;
  const uint32_t kparts = (want > 1 && (K / 32) % (4 * want) == 0) ? want : 1u;
  if (kparts > 1)
    HX_LAUNCH((ks_zero_outputs_kernel<OutT>), dim3((ncols + 255) / 256, num_samples), dim3(256), 0, st, lwe_out, out_idx,
              n_out, num_samples);
  const dim3 grid(col_tiles, split ? kparts : (num_samples + 127) / 128, split ? 1 : kparts);
#define KSM_LAUNCH(L)                                                                                              \
  do {                                                                                                               \
    if (split && level == L)                                                                                         \
      HX_LAUNCH((ks_mfma_kernel<L, false, OutT, 4>), grid, dim3(256), 0, st, lwe_out, out_idx, lwe_in, in_idx,       \
                planes, colsum, n_in, n_out, base_log, num_samples, col_tiles, level, ksk, ksk_words);               \
    else if (split)                                                                                                  \
      HX_LAUNCH((ks_mfma_kernel<L, true, OutT, 4>), grid, dim3(256), 0, st, lwe_out, out_idx, lwe_in, in_idx,        \
                planes, colsum, n_in, n_out, base_log, num_samples, col_tiles, level, ksk, ksk_words);               \
    else if (level == L)                                                                                             \
      HX_LAUNCH((ks_mfma_kernel<L, false, OutT>), grid, dim3(256), 0, st, lwe_out, out_idx, lwe_in, in_idx, planes,   \
                colsum, n_in, n_out, base_log, num_samples, col_tiles, level, ksk, ksk_words);                       \
    else                                                                                                             \
      HX_LAUNCH((ks_mfma_kernel<L, true, OutT>), grid, dim3(256), 0, st, lwe_out, out_idx, lwe_in, in_idx, planes,    \
                colsum, n_in, n_out, base_log, num_samples, col_tiles, level, ksk, ksk_words);                       \
  } while (0)
  switch (level_pad) {
    case 1: KSM_LAUNCH(1); break;
    case 2: KSM_LAUNCH(2); break;
    case 4: KSM_LAUNCH(4); break;
    case 8: KSM_LAUNCH(8); break;
    default: KSM_LAUNCH(16); break;
  }
#undef KSM_LAUNCH
  g_last_keyswitch_path.store(1);
  return true;
}

// what a bootstrap can emit for the keyswitch that follows it (PbsArgs::emit_a): level_pad 4 or 8, 32-bit decomposition,
// and the shape the matrix-core path accepts
bool keyswitch_digits_emittable(uint32_t n_in, uint32_t base_log, uint32_t level, uint32_t *level_pad, uint32_t *steps) {
  uint32_t lp = 1;
  while (lp < level) lp <<= 1;
  const uint32_t K = n_in * lp;
  uint32_t log_k = 0;
  while (((uint64_t)1 << log_k) < K) ++log_k;
  if ((lp != 4 && lp != 8) || K % 32 != 0 || base_log > 6 || base_log + 7 + log_k > 31 || base_log * level > 30) return false;
  *level_pad = lp;
  *steps = K / 32;
  return g_keyswitch_use_mfma.load() && g_keyswitch_split_digits.load();
}

void launch_keyswitch(hipStream_t st, uint64_t *lwe_out, const uint64_t *out_idx, const uint64_t *lwe_in,
                      const uint64_t *in_idx, const uint64_t *ksk, uint32_t n_in, uint32_t n_out,
                      uint32_t base_log, uint32_t level, uint32_t num_samples, const KsDigits *ready) {
  HX_PANIC_IF_FALSE(base_log >= 1 && level >= 1 && base_log * level < 64,
                    "keyswitch: unsupported decomposition (base_log=%u, level=%u)", base_log, level);
  if (num_samples == 0) return;
  if (g_keyswitch_use_mfma.load() &&
      keyswitch_mfma(st, lwe_out, out_idx, lwe_in, in_idx, ksk, n_in, n_out, base_log, level, num_samples, ready))
    return;
  g_last_keyswitch_path.store(0);
  // the scalar kernels stage at most KS_MAXL levels per mask element (the matrix-core path above takes up to 16)
  HX_PANIC_IF_FALSE(level <= KS_MAXL, "keyswitch: level_count %u > %d is only supported by the matrix-core kernel (base_log <= 6)",
                    level, KS_MAXL);
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

void launch_keyswitch_64_32(hipStream_t st, uint32_t *lwe_out, const uint64_t *out_idx, const uint64_t *lwe_in,
                            const uint64_t *in_idx, const uint32_t *ksk, uint32_t n_in, uint32_t n_out,
                            uint32_t base_log, uint32_t level, uint32_t num_samples) {
  // lwe_keyswitch.rs:353-359: the decomposition must fit the OUTPUT width
  HX_PANIC_IF_FALSE(base_log >= 1 && level >= 1 && base_log * level <= 32,
                    "keyswitch 64->32: unsupported decomposition (base_log=%u, level=%u)", base_log, level);
  if (num_samples == 0) return;
  if (g_keyswitch_use_mfma.load() &&
      keyswitch_mfma(st, lwe_out, out_idx, lwe_in, in_idx, ksk, n_in, n_out, base_log, level, num_samples))
    return;
  HX_PANIC_IF_FALSE(level <= KS_MAXL, "keyswitch 64->32: level_count %u > %d is only supported by the matrix-core kernel",
                    level, KS_MAXL);
  const dim3 grid((n_out + 1 + KS_TPB - 1) / KS_TPB, (num_samples + KS_TB - 1) / KS_TB);
  const size_t smem = sizeof(uint32_t) * KS_IC * level * KS_TB;
  HX_LAUNCH(keyswitch_64_32_kernel, grid, dim3(KS_TPB), smem, st, lwe_out, out_idx, lwe_in, in_idx, ksk, n_in, n_out,
            base_log, level, num_samples);
}

}  // namespace tfhe_hip
