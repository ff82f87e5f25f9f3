// multibit.hip — multi-bit PBS (grouping factor g): n/g external products instead of n CMUXes.
//
// Semantics: cc/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:30-65 (modulus switch of
// the 2^g - 1 subset sums), :647-880 (deterministic blind rotation: acc <- LUT*X^-b, then per
// group dst = 0 + src (x) GGSW_comb), key layout of
// cc/algorithms/lwe_multi_bit_bootstrap_key_generation.rs:21-78.
// Like the CPU reference (:116-156) — and unlike the reference's GPU backend, which keeps the key in the standard
// domain and combines with integer monomial products before transforming — the key lives in the FOURIER domain on
// the device (converted once by cuda_convert_lwe_multi_bit_programmable_bootstrap_key_64_async) and the per-LWE
// keybundle  GGSW_comb = GGSW_0 + sum_s GGSW_s (.) FFT(X^{deg_s})  is a pointwise combine with the monomial
// factors of pbs_common.h: no rotations, no keybundle transforms, nothing parked in device memory by the
// throughput kernels.  Operation order fixed (pbs_common.h), so results are bit-exact against the oracle.
//
// pbs_multi_bit_kernel: one workgroup per LWE runs all groups in one launch, building every keybundle element in
// registers right where the multiply-accumulate consumes it.
#include <atomic>

#include "kernels.h"

namespace tfhe_hip {

// keybundle element of polynomial `poly` (index inside one GGSW) at storage slot `slot` / position `pos`
template <int N>
HX_DEV cplx keybundle_point(const cplx *__restrict__ gk, size_t ggsw_c, size_t poly, uint32_t slot, uint32_t pos,
                            uint32_t per, const uint32_t *deg, const double *__restrict__ mono) {
  constexpr int n = N / 2;
  cplx kb = gk[poly * n + slot];  // subset 0: not rotated
  for (uint32_t s = 1; s < per; ++s)
    kb = cmul_add(gk[(size_t)s * ggsw_c + poly * n + slot], monomial_factor<N>(mono, pos, deg[s]), kb);
  return kb;
}

// ACC_GLOBAL (N = 8192, 16384, as in pbs_fft_generic_kernel): the accumulator lives in the per-sample device
// buffer of the scratch; the workgroup runs on one CU, so its writes are visible to its reads after the barrier.
template <int N, int K1, bool ACC_GLOBAL = false>
__global__ void __launch_bounds__(GenericCfg<N>::TPB)
    pbs_multi_bit_kernel(PbsArgs a, uint32_t grouping, FftTables tb) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB, PER = n / TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  const int tid = threadIdx.x;
  const uint32_t sample = blockIdx.x;
  uint64_t *acc = ACC_GLOBAL ? a.acc_scratch + (size_t)sample * K1 * N : (uint64_t *)smem;  // K1*N torus words (src, then dst)
  const FBuf fbuf{(cplx *)(smem + (ACC_GLOBAL ? 0 : (size_t)K1 * N * 8))};                  // n complex points (padded)
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N;
  const cplx *bsk = (const cplx *)a.bsk;  // Fourier domain: [group][subset][level][row][col][slot]
  const uint32_t per = 1u << grouping, groups = a.n / grouping;
  const size_t kb_polys = (size_t)a.level * K1 * K1;
  const size_t ggsw_c = kb_polys * n;  // complex elements per GGSW

  // standard modulus switch of the body (multi-bit sets use no centered correction, :98-103)
  const uint32_t b_hat = (uint32_t)modulus_switch(lwe[a.n], LOG2N2);
  for (int p = 0; p < K1; ++p)
    for (uint32_t j = tid; j < (uint32_t)N; j += TPB) {
      bool neg;
      const uint32_t src = monomial_div_src(j, b_hat, N, neg);
      const uint64_t v = lut[p * N + src];
      acc[p * N + j] = neg ? (uint64_t)0 - v : v;
    }
  __syncthreads();

  for (uint32_t grp = 0; grp < groups; ++grp) {
    const cplx *gk = bsk + (size_t)grp * per * ggsw_c;
    uint32_t deg[16];
    if (a.mb_degrees != nullptr) {  // noise-test entry point: degrees switched ahead, [input][group][subset]
      const uint64_t *pre = a.mb_degrees + ((size_t)a.in_idx[sample] * groups + grp) * per;
      for (uint32_t s = 1; s < per; ++s) deg[s] = (uint32_t)pre[s];
    } else {
      multi_bit_degrees(lwe + (size_t)grp * grouping, grouping, LOG2N2, deg);
    }
    // ---- dst = 0 + src (x) keybundle   (ggsw.rs:483-602 with a zeroed output), keybundle built point by point
    cplx facc[K1][PER];
    bool first = true;
    for (uint32_t idx = 0; idx < a.level; ++idx) {
      for (int row = 0; row < K1; ++row) {
        for (int q = 0; q < PER; ++q) {
          const uint32_t j = tid + q * TPB;
          const int64_t d0 = decomp_digit(acc[row * N + j], a.base_log, a.level, idx);
          const int64_t d1 = decomp_digit(acc[row * N + j + n], a.base_log, a.level, idx);
          fbuf[j] = cplx{i64_to_f64(d0), i64_to_f64(d1)};
        }
        __syncthreads();
        lds_fft_forward<N, TPB>(fbuf, tb.fwd, tid);
        for (int c = 0; c < K1; ++c)
          for (int q = 0; q < PER; ++q) {
            const int pos = tid + q * TPB;
            const cplx y = keybundle_point<N>(gk, ggsw_c, ((size_t)idx * K1 + row) * K1 + c, bsk_slot<N, K1>(pos), pos,
                                              per, deg, tb.mono);
            facc[c][q] = first ? cmul_first(fbuf[pos], y) : cmul_add(fbuf[pos], y, facc[c][q]);
          }
        first = false;
        __syncthreads();
      }
    }
    for (int c = 0; c < K1; ++c) {
      for (int q = 0; q < PER; ++q) fbuf[tid + q * TPB] = facc[c][q];
      __syncthreads();
      lds_fft_inverse<N, TPB>(fbuf, tb.inv, tid);
      for (int q = 0; q < PER; ++q) {
        const int j = tid + q * TPB;
        const cplx y = fbuf[j];
        const double ur = tb.untw[2 * j], ui = tb.untw[2 * j + 1];
        acc[c * N + j] = from_torus(fma(-y.im, ui, y.re * ur));
        acc[c * N + j + n] = from_torus(fma(y.im, ur, y.re * ui));
      }
      __syncthreads();
    }
  }
  block_sample_extract<N, K1, TPB>(a, acc, sample, 0, false, tid);
}

// ------------------------------------------------------------------------- latency path (small batches)
// The keybundles depend on the LWE mask only, not on the accumulator, so they are all built up front by as many
// workgroups as there are (group, keybundle polynomial) pairs — the whole chip works for ONE ciphertext, which is
// what the reference's multi-block launch achieves (programmable_bootstrap_multibit.cuh:40-330) — and only the
// n/g external products remain sequential.  Same integer combine, same transforms, same product order as
// pbs_multi_bit_kernel: identical bits.  Groups are processed in chunks of at most `gcount` (the scratch is
// sized without knowing n, like the reference's lwe_chunk_size), the accumulator crossing chunks in `acc_g`.
// One workgroup per (group, keybundle polynomial, tile of MB_KB_TILE ciphertexts): every key element is loaded once
// and combined for the whole tile (the monomial factors differ per ciphertext, the key does not) — with one
// ciphertext per workgroup 32 ciphertexts re-read the 241 MB key 32 times from L2 (2.5 ms for a round of 32 blocks
// of the radix layer).  Same combine order per ciphertext: identical bits.
constexpr int MB_KB_TILE = 8;  // from 4 ciphertexts up; below, one ciphertext per workgroup (S = 1)
#ifndef MB_KB_SLOTS_FROM
#define MB_KB_SLOTS_FROM 17
#endif
template <int N, int K1, int S>
__global__ void __launch_bounds__(GenericCfg<N>::TPB)
    mb_keybundle_kernel(PbsArgs a, uint32_t grouping, cplx *kb_lat, FftTables tb, uint32_t g0, uint32_t gcount) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  uint32_t (*sdeg)[16] = (uint32_t (*)[16])smem;  // [S][16]
  const int tid = threadIdx.x;
  const uint32_t s0 = blockIdx.y * S;
  const uint32_t count = a.num_samples - s0 < (uint32_t)S ? a.num_samples - s0 : (uint32_t)S;
  const size_t kb_polys = (size_t)a.level * K1 * K1;
  const uint32_t gl = blockIdx.x / (uint32_t)kb_polys, poly = blockIdx.x % (uint32_t)kb_polys, grp = g0 + gl;
  const uint32_t per = 1u << grouping;
  const size_t ggsw_c = kb_polys * n;
  const cplx *gk = (const cplx *)a.bsk + (size_t)grp * per * ggsw_c;
  if (tid < S) {  // the subset degrees of ciphertext s0 + tid
    uint32_t deg[16];
    for (int q = 0; q < 16; ++q) deg[q] = 0;
    if ((uint32_t)tid < count) {
      const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[s0 + tid] * (a.n + 1);
      multi_bit_degrees(lwe + (size_t)grp * grouping, grouping, LOG2N2, deg);
    }
    for (int q = 0; q < 16; ++q) sdeg[tid][q] = deg[q];
  }
  __syncthreads();
  // pointwise combine; parked in transform-POSITION order (what the accumulate kernels index)
  for (uint32_t pos = tid; pos < (uint32_t)n; pos += TPB) {
    const uint32_t slot = bsk_slot<N, K1>(pos);
    cplx kb[S];
    const cplx k0 = gk[poly * n + slot];  // subset 0: not rotated
    HX_UNROLL
    for (int j = 0; j < S; ++j) kb[j] = k0;
    for (uint32_t sb = 1; sb < per; ++sb) {
      const cplx ks = gk[(size_t)sb * ggsw_c + poly * n + slot];
      HX_UNROLL
      for (int j = 0; j < S; ++j) kb[j] = cmul_add(ks, monomial_factor<N>(tb.mono, pos, sdeg[j][sb]), kb[j]);
    }
    HX_UNROLL
    for (int j = 0; j < S; ++j)
      if ((uint32_t)j < count) kb_lat[(((size_t)(s0 + j) * gcount + gl) * kb_polys + poly) * n + pos] = kb[j];
  }
}

// N = 2048, k = 1 (the key is stored in the throughput kernel's slot order, bsk_slot): the form above walks POSITIONS,
// so its key loads are 16 bytes at a stride of 1 KB, and it gathers one monomial base per (position, subset,
// ciphertext) — 480 dependent table loads per thread at S = 8: 0.26 ms for a round of 7-14 radix blocks, 1.1 ms for 32.
// Here a thread walks the SLOTS tid, tid + 256, ...: the key loads of a wave are contiguous, and its four slots are the
// positions hi*16 + lo0 + 4q of ONE hi = tid % 64 — one gathered base per (subset, ciphertext) serves all four, the
// 16th roots come from a 16-entry LDS table (M_d[p] = base(hi, d) * w16[(bitrev4(lo) d) mod 16], pbs_common.h).  The
// keybundles are parked in position order as before (the accumulate kernel reads 64 contiguous bytes per thread):
// each ciphertext's polynomial goes through a padded LDS buffer to turn slot order into position order.  Same factors,
// same multiply-add order per position: identical bits.
// SLOTS: the keybundles are parked in the key's own slot order (what the latency kernel reads, like a classic key:
// no transposition here); otherwise in position order (the generic product kernels)
template <int S, bool SLOTS>
__global__ void __launch_bounds__(256)
    mb_keybundle_2048_kernel(PbsArgs a, uint32_t grouping, cplx *kb_lat, FftTables tb, uint32_t g0, uint32_t gcount) {
  constexpr int N = 2048, n = N / 2, TPB = 256, LOG2N2 = 12;
  __shared__ uint32_t sdeg[S][16];
  __shared__ cplx sw16[16];
  __shared__ cplx xbuf[n + n / 16];
  const int tid = threadIdx.x;
  const uint32_t s0 = blockIdx.y * S;
  const uint32_t count = a.num_samples - s0 < (uint32_t)S ? a.num_samples - s0 : (uint32_t)S;
  const size_t kb_polys = (size_t)a.level * 4;
  const uint32_t gl = blockIdx.x / (uint32_t)kb_polys, poly = blockIdx.x % (uint32_t)kb_polys, grp = g0 + gl;
  const uint32_t per = 1u << grouping;
  const size_t ggsw_c = kb_polys * n;
  const cplx *gk = (const cplx *)a.bsk + (size_t)grp * per * ggsw_c + (size_t)poly * n + tid;
  if (tid < S) {  // the subset degrees of ciphertext s0 + tid
    uint32_t deg[16];
    for (int q = 0; q < 16; ++q) deg[q] = 0;
    if ((uint32_t)tid < count) {
      const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[s0 + tid] * (a.n + 1);
      multi_bit_degrees(lwe + (size_t)grp * grouping, grouping, LOG2N2, deg);
    }
    for (int q = 0; q < 16; ++q) sdeg[tid][q] = deg[q];
  }
  if (tid >= 64 && tid < 80) {
    const int t = tid - 64;
    sw16[t] = cplx{tb.mono[2 * (N / 8) * t], tb.mono[2 * (N / 8) * t + 1]};
  }
  __syncthreads();
  const uint32_t hi = (uint32_t)tid & 63u, lo0 = (uint32_t)tid >> 6;
  uint32_t br[4];  // bitrev4 of my four lo values
  HX_UNROLL
  for (int q = 0; q < 4; ++q) br[q] = __brev(lo0 + 4u * q) >> 28;
  cplx kb[4][S];
  HX_UNROLL
  for (int q = 0; q < 4; ++q) {
    const cplx k0 = gk[q * TPB];  // subset 0: not rotated
    HX_UNROLL
    for (int j = 0; j < S; ++j) kb[q][j] = k0;
  }
  for (uint32_t sb = 1; sb < per; ++sb) {
    cplx ks[4];
    HX_UNROLL
    for (int q = 0; q < 4; ++q) ks[q] = gk[(size_t)sb * ggsw_c + q * TPB];
    HX_UNROLL
    for (int j = 0; j < S; ++j) {
      const uint32_t deg = sdeg[j][sb];
      const uint32_t jb = monomial_base_index<N>(hi, deg);
      const cplx base{tb.mono[2 * jb], tb.mono[2 * jb + 1]};
      HX_UNROLL
      for (int q = 0; q < 4; ++q) kb[q][j] = cmul_add(ks[q], cmul_first(base, sw16[(br[q] * deg) & 15u]), kb[q][j]);
    }
  }
  if constexpr (SLOTS) {
    HX_UNROLL
    for (int j = 0; j < S; ++j)
      if ((uint32_t)j < count) {
        cplx *dst = kb_lat + (((size_t)(s0 + j) * gcount + gl) * kb_polys + poly) * n;
        HX_UNROLL
        for (int q = 0; q < 4; ++q) dst[tid + q * TPB] = kb[q][j];
      }
    return;
  }
  // slot order -> position order through LDS, one ciphertext at a time
  const FBuf xb{xbuf};
  HX_UNROLL
  for (int j = 0; j < S; ++j) {
    HX_UNROLL
    for (int q = 0; q < 4; ++q) xb[(int)(hi * 16u + lo0 + 4u * q)] = kb[q][j];
    __syncthreads();
    if ((uint32_t)j < count) {
      cplx *dst = kb_lat + (((size_t)(s0 + j) * gcount + gl) * kb_polys + poly) * n;
      HX_UNROLL
      for (int q = 0; q < 4; ++q) dst[tid + q * TPB] = xb[tid + q * TPB];
    }
    __syncthreads();
  }
}

template <int N, int K1>
__global__ void __launch_bounds__(GenericCfg<N>::TPB)
    mb_accumulate_kernel(PbsArgs a, const cplx *kb_lat, FftTables tb, uint64_t *acc_g, uint32_t gcount, uint32_t gpass,
                         int first, int last) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB, PER = n / TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  uint64_t *acc = (uint64_t *)smem;
  const FBuf fbuf{(cplx *)(smem + (size_t)K1 * N * 8)};
  const int tid = threadIdx.x;
  const uint32_t sample = blockIdx.x;
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N;
  const size_t kb_polys = (size_t)a.level * K1 * K1;
  uint64_t *mine = acc_g + (size_t)sample * K1 * N;
  if (first) {
    const uint32_t b_hat = (uint32_t)modulus_switch(lwe[a.n], LOG2N2);
    for (int p = 0; p < K1; ++p)
      for (uint32_t j = tid; j < (uint32_t)N; j += TPB) {
        bool neg;
        const uint32_t src = monomial_div_src(j, b_hat, N, neg);
        const uint64_t v = lut[p * N + src];
        acc[p * N + j] = neg ? (uint64_t)0 - v : v;
      }
  } else {
    for (uint32_t j = tid; j < (uint32_t)(K1 * N); j += TPB) acc[j] = mine[j];
  }
  __syncthreads();
  for (uint32_t gl = 0; gl < gpass; ++gl) {
    const cplx *kb = kb_lat + (((size_t)sample * gcount + gl) * kb_polys) * n;
    cplx facc[K1][PER];
    bool firstp = true;
    for (uint32_t idx = 0; idx < a.level; ++idx) {
      for (int row = 0; row < K1; ++row) {
        for (int q = 0; q < PER; ++q) {
          const uint32_t j = tid + q * TPB;
          const int64_t d0 = decomp_digit(acc[row * N + j], a.base_log, a.level, idx);
          const int64_t d1 = decomp_digit(acc[row * N + j + n], a.base_log, a.level, idx);
          fbuf[j] = cplx{i64_to_f64(d0), i64_to_f64(d1)};
        }
        __syncthreads();
        lds_fft_forward<N, TPB>(fbuf, tb.fwd, tid);
        const cplx *brow = kb + (((size_t)idx * K1 + row) * K1) * n;
        for (int c = 0; c < K1; ++c)
          for (int q = 0; q < PER; ++q) {
            const int pos = tid + q * TPB;
            const cplx y = brow[(size_t)c * n + pos];
            facc[c][q] = firstp ? cmul_first(fbuf[pos], y) : cmul_add(fbuf[pos], y, facc[c][q]);
          }
        firstp = false;
        __syncthreads();
      }
    }
    for (int c = 0; c < K1; ++c) {
      for (int q = 0; q < PER; ++q) fbuf[tid + q * TPB] = facc[c][q];
      __syncthreads();
      lds_fft_inverse<N, TPB>(fbuf, tb.inv, tid);
      for (int q = 0; q < PER; ++q) {
        const int j = tid + q * TPB;
        const cplx y = fbuf[j];
        const double ur = tb.untw[2 * j], ui = tb.untw[2 * j + 1];
        acc[c * N + j] = from_torus(fma(-y.im, ui, y.re * ur));
        acc[c * N + j + n] = from_torus(fma(y.im, ur, y.re * ui));
      }
      __syncthreads();
    }
  }
  if (last) {
    block_sample_extract<N, K1, TPB>(a, acc, sample, 0, false, tid);
  } else {
    for (uint32_t j = tid; j < (uint32_t)(K1 * N); j += TPB) mine[j] = acc[j];
  }
}

// The same products with one thread group per GLWE polynomial (as pbs_fft_par_kernel): the k+1 forward transforms
// of a level and the k+1 inverse transforms run side by side, half the barrier-separated stages for k = 1.
// Per output point the products are accumulated in the same order (level, then row): identical bits.
template <int N, int K1>
__global__ void __launch_bounds__(K1 *GenericCfg<N>::TPB)
    mb_accumulate_par_kernel(PbsArgs a, const cplx *kb_lat, FftTables tb, uint64_t *acc_g, uint32_t gcount,
                             uint32_t gpass, int first, int last) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB, TPBT = K1 * TPB, PER = n / TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  uint64_t *acc = (uint64_t *)smem;
  cplx *fbase = (cplx *)(smem + (size_t)K1 * N * 8);
  const int tid = threadIdx.x;
  const int grp = tid / TPB, lt = tid - grp * TPB;  // my row (forward) / column (inverse), thread inside it
  const uint32_t sample = blockIdx.x;
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N;
  const size_t kb_polys = (size_t)a.level * K1 * K1;
  uint64_t *mine = acc_g + (size_t)sample * K1 * N;
  const FBuf mybuf{fbase + (size_t)grp * fbuf_slots(N)};
  if (first) {
    const uint32_t b_hat = (uint32_t)modulus_switch(lwe[a.n], LOG2N2);
    for (uint32_t j = lt; j < (uint32_t)N; j += TPB) {
      bool neg;
      const uint32_t src = monomial_div_src(j, b_hat, N, neg);
      const uint64_t v = lut[grp * N + src];
      acc[grp * N + j] = neg ? (uint64_t)0 - v : v;
    }
  } else {
    for (uint32_t j = tid; j < (uint32_t)(K1 * N); j += TPBT) acc[j] = mine[j];
  }
  __syncthreads();
  for (uint32_t gl = 0; gl < gpass; ++gl) {
    const cplx *kb = kb_lat + (((size_t)sample * gcount + gl) * kb_polys) * n;
    cplx facc[PER];
    for (uint32_t idx = 0; idx < a.level; ++idx) {
      for (int q = 0; q < PER; ++q) {  // digits of my row of the accumulator itself
        const uint32_t j = lt + q * TPB;
        const int64_t d0 = decomp_digit(acc[grp * N + j], a.base_log, a.level, idx);
        const int64_t d1 = decomp_digit(acc[grp * N + j + n], a.base_log, a.level, idx);
        mybuf[j] = cplx{i64_to_f64(d0), i64_to_f64(d1)};
      }
      __syncthreads();
      lds_fft_forward<N, TPB>(mybuf, tb.fwd, lt);
      for (int row = 0; row < K1; ++row) {  // column `grp` of the external product
        const cplx *brow = kb + (((size_t)idx * K1 + row) * K1 + grp) * n;
        const FBuf f{fbase + (size_t)row * fbuf_slots(N)};
        for (int q = 0; q < PER; ++q) {
          const int pos = lt + q * TPB;
          const cplx y = brow[pos];
          facc[q] = (idx == 0 && row == 0) ? cmul_first(f[pos], y) : cmul_add(f[pos], y, facc[q]);
        }
      }
      __syncthreads();
    }
    for (int q = 0; q < PER; ++q) mybuf[lt + q * TPB] = facc[q];
    __syncthreads();
    lds_fft_inverse<N, TPB>(mybuf, tb.inv, lt);
    for (int q = 0; q < PER; ++q) {
      const int j = lt + q * TPB;
      const cplx y = mybuf[j];
      const double ur = tb.untw[2 * j], ui = tb.untw[2 * j + 1];
      acc[grp * N + j] = from_torus(fma(-y.im, ui, y.re * ur));
      acc[grp * N + j + n] = from_torus(fma(y.im, ur, y.re * ui));
    }
    __syncthreads();
  }
  if (last) {
    block_sample_extract<N, K1, TPBT>(a, acc, sample, 0, false, tid);
  } else {
    for (uint32_t j = tid; j < (uint32_t)(K1 * N); j += TPBT) mine[j] = acc[j];
  }
}

template <int N, int K1>
static void launch_mb_latency(hipStream_t st, const MultiBitArgs &m, const FftTables &tb, cplx *kb_lat,
                              uint32_t group_chunk, uint64_t *acc_g) {
  const PbsArgs &a = m.pbs;
  const uint32_t groups = a.n / m.grouping_factor, kb_polys = a.level * K1 * K1;
  const size_t smem_b = (size_t)K1 * N * 8 + fbuf_bytes(N);
  const size_t smem_p = (size_t)K1 * N * 8 + (size_t)K1 * fbuf_bytes(N);
  const bool par = K1 == 2 && !g_ntt_kernel_serial;  // same rule as the classic generic kernels (hip_backend_set_ntt_kernel)
  if (par)
    hx_set_dynamic_smem_once<mb_accumulate_par_kernel<N, K1>>(smem_p);
  else
    hx_set_dynamic_smem_once<mb_accumulate_kernel<N, K1>>(smem_b);
  for (uint32_t g0 = 0; g0 < groups; g0 += group_chunk) {
    const uint32_t gpass = groups - g0 < group_chunk ? groups - g0 : group_chunk;
    const bool block_products = N == 2048 && K1 == 2 && a.level <= 8 && !g_ntt_kernel_serial && !a.mb_generic_products;
    // From MB_KB_SLOTS_FROM ciphertexts on the keybundles stay in the key's slot order (the latency kernel reads them
    // like a classic key): the keybundle kernel drops its slot -> position transposition, 4-10 % of a round of 32-256
    // blocks; below, position order — one 64-byte run per thread and row is worth 3 % to a PBS that runs alone
    const bool slots = block_products && a.num_samples >= (uint32_t)MB_KB_SLOTS_FROM;
    if (N == 2048 && K1 == 2 && a.num_samples < 4) {
      if (slots)
        HX_LAUNCH((mb_keybundle_2048_kernel<1, true>), dim3(gpass * kb_polys, a.num_samples), dim3(256), 0, st, a,
                  m.grouping_factor, kb_lat, tb, g0, group_chunk);
      else
        HX_LAUNCH((mb_keybundle_2048_kernel<1, false>), dim3(gpass * kb_polys, a.num_samples), dim3(256), 0, st, a,
                  m.grouping_factor, kb_lat, tb, g0, group_chunk);
    } else if (N == 2048 && K1 == 2) {
      const dim3 grid(gpass * kb_polys, (a.num_samples + MB_KB_TILE - 1) / MB_KB_TILE);
      if (slots)
        HX_LAUNCH((mb_keybundle_2048_kernel<MB_KB_TILE, true>), grid, dim3(256), 0, st, a, m.grouping_factor, kb_lat, tb,
                  g0, group_chunk);
      else
        HX_LAUNCH((mb_keybundle_2048_kernel<MB_KB_TILE, false>), grid, dim3(256), 0, st, a, m.grouping_factor, kb_lat,
                  tb, g0, group_chunk);
    }
    else if (a.num_samples < 4)
      HX_LAUNCH((mb_keybundle_kernel<N, K1, 1>), dim3(gpass * kb_polys, a.num_samples), dim3(GenericCfg<N>::TPB),
                16 * sizeof(uint32_t), st, a, m.grouping_factor, kb_lat, tb, g0, group_chunk);
    else
      HX_LAUNCH((mb_keybundle_kernel<N, K1, MB_KB_TILE>), dim3(gpass * kb_polys, (a.num_samples + MB_KB_TILE - 1) / MB_KB_TILE),
                dim3(GenericCfg<N>::TPB), MB_KB_TILE * 16 * sizeof(uint32_t), st, a, m.grouping_factor, kb_lat, tb, g0,
                group_chunk);
    if (block_products)
      // the latency kernel's structure (registers + wave-local exchanges, 4 barriers per product)
      launch_mb_accumulate_block(st, a, tb, (const cplx *)kb_lat, acc_g, group_chunk, gpass, (int)(g0 == 0),
                                 (int)(g0 + gpass == groups), (int)slots);
    else if (par)
      HX_LAUNCH((mb_accumulate_par_kernel<N, K1>), dim3(a.num_samples), dim3(K1 * GenericCfg<N>::TPB), smem_p, st, a,
                (const cplx *)kb_lat, tb, acc_g, group_chunk, gpass, (int)(g0 == 0), (int)(g0 + gpass == groups));
    else
      HX_LAUNCH((mb_accumulate_kernel<N, K1>), dim3(a.num_samples), dim3(GenericCfg<N>::TPB), smem_b, st, a,
                (const cplx *)kb_lat, tb, acc_g, group_chunk, gpass, (int)(g0 == 0), (int)(g0 + gpass == groups));
  }
}

template <int N, int K1>
static void launch_mb(hipStream_t st, const MultiBitArgs &m, const FftTables &tb) {
  const size_t smem = (size_t)K1 * N * 8 + fbuf_bytes(N);
  hx_set_dynamic_smem_once<pbs_multi_bit_kernel<N, K1>>(smem);
  HX_LAUNCH((pbs_multi_bit_kernel<N, K1>), dim3(m.pbs.num_samples), dim3(GenericCfg<N>::TPB), smem, st, m.pbs,
            m.grouping_factor, tb);
}
template <int N>
static void launch_mb_big(hipStream_t st, const MultiBitArgs &m, const FftTables &tb) {  // N >= 8192, k = 1
  HX_PANIC_IF_FALSE(m.pbs.acc_scratch != nullptr, "multi-bit PBS scratch of a polynomial_size >= 8192 set has no accumulator buffer");
  const size_t smem = fbuf_bytes(N);
  hx_set_dynamic_smem_once<pbs_multi_bit_kernel<N, 2, true>>(smem);
  HX_LAUNCH((pbs_multi_bit_kernel<N, 2, true>), dim3(m.pbs.num_samples), dim3(GenericCfg<N>::TPB), smem, st, m.pbs,
            m.grouping_factor, tb);
}

void launch_pbs_multi_bit(hipStream_t st, uint32_t N, uint32_t glwe_dim, const MultiBitArgs &m0, const FftTables &tb,
                          uint64_t *acc_scratch) {
  MultiBitArgs m = m0;
  m.pbs.acc_scratch = acc_scratch;
  const uint32_t k1 = glwe_dim + 1;
  bool ok = true;
  switch (N) {
    case 8192: if (k1 == 2) launch_mb_big<8192>(st, m, tb); else ok = false; break;
    case 16384: if (k1 == 2) launch_mb_big<16384>(st, m, tb); else ok = false; break;
    case 256: if (k1 == 2) launch_mb<256, 2>(st, m, tb); else if (k1 == 3) launch_mb<256, 3>(st, m, tb); else if (k1 == 4) launch_mb<256, 4>(st, m, tb); else ok = false; break;
    case 512: if (k1 == 2) launch_mb<512, 2>(st, m, tb); else if (k1 == 3) launch_mb<512, 3>(st, m, tb); else if (k1 == 4) launch_mb<512, 4>(st, m, tb); else ok = false; break;
    case 1024: if (k1 == 2) launch_mb<1024, 2>(st, m, tb); else if (k1 == 3) launch_mb<1024, 3>(st, m, tb); else if (k1 == 4) launch_mb<1024, 4>(st, m, tb); else ok = false; break;
    case 2048: if (k1 == 2) launch_mb<2048, 2>(st, m, tb); else if (k1 == 3) launch_mb<2048, 3>(st, m, tb); else ok = false; break;
    case 4096: if (k1 == 2) launch_mb<4096, 2>(st, m, tb); else ok = false; break;
    default: ok = false;
  }
  if (!ok) HX_PANIC("unsupported (polynomial_size=%u, glwe_dimension=%u) for the multi-bit PBS", N, glwe_dim);
}

void launch_pbs_multi_bit_latency(hipStream_t st, uint32_t N, uint32_t glwe_dim, const MultiBitArgs &m,
                                  const FftTables &tb, cplx *kb_lat, uint32_t group_chunk, uint64_t *acc_g) {
  const uint32_t k1 = glwe_dim + 1;
  bool ok = true;
  switch (N) {
    case 256: if (k1 == 2) launch_mb_latency<256, 2>(st, m, tb, kb_lat, group_chunk, acc_g); else if (k1 == 3) launch_mb_latency<256, 3>(st, m, tb, kb_lat, group_chunk, acc_g); else if (k1 == 4) launch_mb_latency<256, 4>(st, m, tb, kb_lat, group_chunk, acc_g); else ok = false; break;
    case 512: if (k1 == 2) launch_mb_latency<512, 2>(st, m, tb, kb_lat, group_chunk, acc_g); else if (k1 == 3) launch_mb_latency<512, 3>(st, m, tb, kb_lat, group_chunk, acc_g); else if (k1 == 4) launch_mb_latency<512, 4>(st, m, tb, kb_lat, group_chunk, acc_g); else ok = false; break;
    case 1024: if (k1 == 2) launch_mb_latency<1024, 2>(st, m, tb, kb_lat, group_chunk, acc_g); else if (k1 == 3) launch_mb_latency<1024, 3>(st, m, tb, kb_lat, group_chunk, acc_g); else if (k1 == 4) launch_mb_latency<1024, 4>(st, m, tb, kb_lat, group_chunk, acc_g); else ok = false; break;
    case 2048: if (k1 == 2) launch_mb_latency<2048, 2>(st, m, tb, kb_lat, group_chunk, acc_g); else if (k1 == 3) launch_mb_latency<2048, 3>(st, m, tb, kb_lat, group_chunk, acc_g); else ok = false; break;
    case 4096: if (k1 == 2) launch_mb_latency<4096, 2>(st, m, tb, kb_lat, group_chunk, acc_g); else ok = false; break;
    default: ok = false;
  }
  if (!ok) HX_PANIC("unsupported (polynomial_size=%u, glwe_dimension=%u) for the multi-bit PBS", N, glwe_dim);
}

}  // namespace tfhe_hip
