// pbs_generic.hip — reference-shaped PBS kernels for every supported (N, k, l): one
// workgroup per LWE, accumulator resident in LDS for all n CMUX iterations, one transform
// buffer in LDS, Fourier/NTT-domain output accumulators in registers.
//
// Path restated: cc/fft_impl/fft64/crypto/bootstrap.rs:294-380,480-520 (f64 engine) and
// cc/algorithms/lwe_programmable_bootstrapping/ntt64_bnf_pbs.rs:208-280,541-705 (NTT engine);
// replaces backends/tfhe-cuda-backend/cuda/src/pbs/programmable_bootstrap_classic.cuh:783-1033
// (two launches per iteration there; the whole loop is one launch here).
//
// These kernels favour generality; the throughput kernel for the headline parameter set lives
// in pbs_fft_wave.hip and produces bit-identical results (same butterfly dataflow).
#include "pbs_common.h"
#include "kernels.h"

namespace tfhe_hip {
bool g_ntt_kernel_serial = false;

template <int N>
HX_DEV uint64_t rot_sub(const uint64_t *poly, uint32_t j, uint32_t a_hat) {
  bool neg;
  const uint32_t src = monomial_mul_src(j, a_hat, N, neg);
  const uint64_t s = poly[src];
  return (neg ? (uint64_t)0 - s : s) - poly[j];
}

// digit `idx` of x as an f64: one-level sets read it off the high dword, and every digit of a base below
// 2^31 converts from 32 bits (one instruction instead of the 64-bit conversion sequence)
HX_DEV double digit_f64(uint64_t x, uint32_t base_log, uint32_t level, uint32_t idx) {
  if (level == 1 && base_log <= 30) return (double)decomp_digit_l1_hi((uint32_t)(x >> 32), base_log);
  const int64_t d = decomp_digit(x, base_log, level, idx);
  return base_log <= 31 ? (double)(int32_t)d : i64_to_f64(d);
}

// ------------------------------------------------------------------------- f64 engine
// ACC_GLOBAL (N = 8192, 16384: programmable_bootstrap_classic.cuh supports rings up to 2^14): the accumulator
// is kept in a per-sample device buffer instead of LDS; a workgroup runs on one CU, so its own writes are
// visible to its later reads through that CU's L1 after the workgroup barrier.
template <int N, int K1, bool ACC_GLOBAL = false>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) pbs_fft_generic_kernel(PbsArgs a, FftTables tb) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB, PER = n / TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  const int tid = threadIdx.x;
  const uint32_t sample = blockIdx.x;
  uint64_t *acc = ACC_GLOBAL ? a.acc_scratch + (size_t)sample * K1 * N : (uint64_t *)smem;  // K1*N torus words
  const FBuf fbuf{(cplx *)(smem + (ACC_GLOBAL ? 0 : (size_t)K1 * N * 8))};                  // n complex points, padded
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N;
  const cplx *bsk = (const cplx *)a.bsk;

  const uint32_t b_hat = block_body_modulus_switch<TPB>(lwe, a.n, LOG2N2, a.ms_type, (uint64_t *)fbuf.p, tid);
  // acc <- LUT * X^{-b_hat}
  for (int p = 0; p < K1; ++p)
    for (uint32_t j = tid; j < (uint32_t)N; j += TPB) {
      bool neg;
      const uint32_t src = monomial_div_src(j, b_hat, N, neg);
      const uint64_t v = lut[p * N + src];
      acc[p * N + j] = neg ? (uint64_t)0 - v : v;
    }
  __syncthreads();

  for (uint32_t i = 0; i < a.n; ++i) {
    const uint32_t a_hat = (uint32_t)modulus_switch(lwe[i], LOG2N2);
    if (a_hat == 0) continue;  // uniform across the workgroup (bootstrap.rs:334)
    cplx facc[K1][PER];
    bool first = true;
    for (uint32_t idx = 0; idx < a.level; ++idx) {
      for (int row = 0; row < K1; ++row) {
        // ct1 = acc*X^a_hat - acc, decomposed on the fly; fold N reals into n complex
        for (int q = 0; q < PER; ++q) {
          const uint32_t j = tid + q * TPB;
          // (digit_f64 of the kernel below costs this one 50 registers and a workgroup per CU: measured slower)
          const int64_t d0 = decomp_digit(rot_sub<N>(acc + row * N, j, a_hat), a.base_log, a.level, idx);
          const int64_t d1 = decomp_digit(rot_sub<N>(acc + row * N, j + n, a_hat), a.base_log, a.level, idx);
          fbuf[j] = cplx{i64_to_f64(d0), i64_to_f64(d1)};
        }
        __syncthreads();
        lds_fft_forward<N, TPB>(fbuf, tb.fwd, tid);
        const cplx *brow = bsk + ((((size_t)i * a.level + idx) * K1 + row) * K1) * n;
        for (int c = 0; c < K1; ++c)
          for (int q = 0; q < PER; ++q) {
            const int pos = tid + q * TPB;
            const cplx y = brow[(size_t)c * n + bsk_slot<N, K1>(pos)];
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
        const double tr = fma(-y.im, ui, y.re * ur);
        const double ti = fma(y.im, ur, y.re * ui);
        acc[c * N + j] += from_torus(tr);
        acc[c * N + j + n] += from_torus(ti);
      }
      __syncthreads();
    }
  }
  block_sample_extract<N, K1, TPB>(a, acc, sample, 0, false, tid);
}

// Same engine, one thread group per GLWE polynomial: the k+1 forward transforms of a level (one per row) and
// the k+1 inverse transforms (one per column) run side by side, so a CMUX has (k+1)x fewer barrier-separated
// stages.  Per output point the products are accumulated in the same order (level, then row): identical bits.
template <int N, int K1>
__global__ void __launch_bounds__(K1 *GenericCfg<N>::TPB) pbs_fft_par_kernel(PbsArgs a, FftTables tb) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB, TPBT = K1 * TPB, PER = n / TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  uint64_t *acc = (uint64_t *)smem;                  // K1*N torus words
  cplx *fbase = (cplx *)(smem + (size_t)K1 * N * 8);  // K1 padded transform buffers of n complex points
  const int tid = threadIdx.x;
  const int grp = tid / TPB, lt = tid - grp * TPB;   // my row (forward) / column (inverse), thread inside it
  const uint32_t sample = blockIdx.x;
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N;
  const cplx *bsk = (const cplx *)a.bsk;
  const FBuf mybuf{fbase + (size_t)grp * fbuf_slots(N)};

  // body modulus switch; TPBT need not be a power of two (k = 2), so the reduction is a plain sum
  uint64_t corr = 0;
  if (a.ms_type == 1) {
    uint64_t *scratch = (uint64_t *)fbase;
    uint64_t sh = 0;
    int64_t sd = 0;
    for (uint32_t i = tid; i < a.n; i += TPBT) {
      uint64_t h;
      int64_t d;
      centered_ms_terms(lwe[i], LOG2N2, h, d);
      sh += h;
      sd += d;
    }
    scratch[tid] = sh;
    scratch[TPBT + tid] = (uint64_t)sd;
    __syncthreads();
    uint64_t th = 0, td = 0;
    for (int l = 0; l < TPBT; ++l) {
      th += scratch[l];
      td += scratch[TPBT + l];
    }
    __syncthreads();
    corr = centered_ms_finish(th, (int64_t)td, LOG2N2);
  }
  const uint32_t b_hat = (uint32_t)modulus_switch(lwe[a.n] + corr, LOG2N2);
  for (uint32_t j = lt; j < (uint32_t)N; j += TPB) {  // acc <- LUT * X^{-b_hat}
    bool neg;
    const uint32_t src = monomial_div_src(j, b_hat, N, neg);
    const uint64_t v = lut[grp * N + src];
    acc[grp * N + j] = neg ? (uint64_t)0 - v : v;
  }
  __syncthreads();

  for (uint32_t i = 0; i < a.n; ++i) {
    const uint32_t a_hat = (uint32_t)modulus_switch(lwe[i], LOG2N2);
    if (a_hat == 0) continue;  // uniform across the workgroup (bootstrap.rs:334)
    cplx facc[PER];
    for (uint32_t idx = 0; idx < a.level; ++idx) {
      for (int q = 0; q < PER; ++q) {  // digits of my row
        const uint32_t j = lt + q * TPB;
        mybuf[j] = cplx{digit_f64(rot_sub<N>(acc + grp * N, j, a_hat), a.base_log, a.level, idx),
                        digit_f64(rot_sub<N>(acc + grp * N, j + n, a_hat), a.base_log, a.level, idx)};
      }
      __syncthreads();
      lds_fft_forward<N, TPB>(mybuf, tb.fwd, lt);
      for (int row = 0; row < K1; ++row) {  // column `grp` of the external product
        const cplx *brow = bsk + ((((size_t)i * a.level + idx) * K1 + row) * K1 + grp) * n;
        const FBuf f{fbase + (size_t)row * fbuf_slots(N)};
        for (int q = 0; q < PER; ++q) {
          const int pos = lt + q * TPB;
          const cplx y = brow[bsk_slot<N, K1>(pos)];
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
      acc[grp * N + j] += from_torus(fma(-y.im, ui, y.re * ur));
      acc[grp * N + j + n] += from_torus(fma(y.im, ur, y.re * ui));
    }
    __syncthreads();
  }
  block_sample_extract<N, K1, TPBT>(a, acc, sample, 0, false, tid);
}

// ------------------------------------------------------------------------- exact engine
// External products by exact negacyclic convolution mod 2^64 on the standard-domain key
// (cc/algorithms/lwe_programmable_bootstrapping/karatsuba_pbs.rs:199-413; any exact product gives
// the same bits).  O(N^2) per polynomial product: a verification engine — it lets the GPU path
// reproduce the reference's own golden vectors (apps/test-vectors, *_karatsuba files) bit for bit.
template <int N, int K1>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) pbs_exact_generic_kernel(PbsArgs a) {
  constexpr int TPB = GenericCfg<N>::TPB, PER = N / TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  uint64_t *acc = (uint64_t *)smem;          // K1*N
  int64_t *dig = (int64_t *)(acc + (size_t)K1 * N);  // N digits of one (level, row)
  const int tid = threadIdx.x;
  const uint32_t sample = blockIdx.x;
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N;
  const uint64_t *bsk = (const uint64_t *)a.bsk;

  const uint32_t b_hat = block_body_modulus_switch<TPB>(lwe, a.n, LOG2N2, a.ms_type, (uint64_t *)dig, tid);
  for (int p = 0; p < K1; ++p)
    for (uint32_t j = tid; j < (uint32_t)N; j += TPB) {
      bool neg;
      const uint32_t src = monomial_div_src(j, b_hat, N, neg);
      const uint64_t v = lut[p * N + src];
      acc[p * N + j] = neg ? (uint64_t)0 - v : v;
    }
  __syncthreads();

  for (uint32_t i = 0; i < a.n; ++i) {
    const uint32_t a_hat = (uint32_t)modulus_switch(lwe[i], LOG2N2);
    if (a_hat == 0) continue;
    uint64_t oacc[K1][PER];
    for (int c = 0; c < K1; ++c)
      for (int q = 0; q < PER; ++q) oacc[c][q] = 0;
    for (uint32_t idx = 0; idx < a.level; ++idx) {
      for (int row = 0; row < K1; ++row) {
        for (int q = 0; q < PER; ++q) {
          const uint32_t j = tid + q * TPB;
          dig[j] = decomp_digit(rot_sub<N>(acc + row * N, j, a_hat), a.base_log, a.level, idx);
        }
        __syncthreads();
        const uint64_t *brow = bsk + ((((size_t)i * a.level + idx) * K1 + row) * K1) * N;
        for (int c = 0; c < K1; ++c)
          for (int q = 0; q < PER; ++q) {
            const int m = tid + q * TPB;
            uint64_t sum = 0;
            for (int j = 0; j <= m; ++j) sum += (uint64_t)dig[j] * brow[(size_t)c * N + (m - j)];
            for (int j = m + 1; j < N; ++j) sum -= (uint64_t)dig[j] * brow[(size_t)c * N + (N + m - j)];
            oacc[c][q] += sum;
          }
        __syncthreads();
      }
    }
    for (int c = 0; c < K1; ++c)
      for (int q = 0; q < PER; ++q) acc[c * N + tid + q * TPB] += oacc[c][q];
    __syncthreads();
  }
  block_sample_extract<N, K1, TPB>(a, acc, sample, 0, false, tid);
}

// ------------------------------------------------------------------------- NTT engine
template <int N, int K1>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) pbs_ntt_generic_kernel(PbsArgs a, NttTables tb) {
  constexpr int TPB = GenericCfg<N>::TPB, PER = N / TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  uint64_t *acc = (uint64_t *)smem;    // K1*N
  uint64_t *nbuf = acc + (size_t)K1 * N;  // N
  const int tid = threadIdx.x;
  const uint32_t sample = blockIdx.x;
  if (a.only_flagged != nullptr) {  // the recomputation behind a split-key launch: its flagged ciphertexts only
    if (a.only_flagged[sample] == 0u) return;
    if (tid == 0 && a.recomputed != nullptr) atomicAdd(a.recomputed, 1u);
  }
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N;
  const uint64_t *bsk = (const uint64_t *)a.bsk;

  const uint32_t b_hat = block_body_modulus_switch<TPB>(lwe, a.n, LOG2N2, a.ms_type, nbuf, tid);
  for (int p = 0; p < K1; ++p)
    for (uint32_t j = tid; j < (uint32_t)N; j += TPB) acc[p * N + j] = lut[p * N + j];
  __syncthreads();

  for (uint32_t i = 0; i < a.n; ++i) {
    const uint32_t a_hat = (uint32_t)modulus_switch(lwe[i], LOG2N2);
    if (a_hat == 0) continue;
    uint64_t nacc[K1][PER];
    for (int c = 0; c < K1; ++c)
      for (int q = 0; q < PER; ++q) nacc[c][q] = 0;
    for (uint32_t idx = 0; idx < a.level; ++idx) {
      for (int row = 0; row < K1; ++row) {
        for (int q = 0; q < PER; ++q) {
          const uint32_t j = tid + q * TPB;
          const int64_t d = decomp_digit(rot_sub<N>(acc + row * N, j, a_hat), a.base_log, a.level, idx);
          nbuf[j] = d < 0 ? (uint64_t)d + GL_P : (uint64_t)d;  // ntt64.rs:199-220
        }
        __syncthreads();
        lds_ntt_forward<N, TPB>(nbuf, tb.tw, tid);
        const uint64_t *brow = bsk + ((((size_t)i * a.level + idx) * K1 + row) * K1) * N;
        for (int c = 0; c < K1; ++c)
          for (int q = 0; q < PER; ++q) {
            const int pos = tid + q * TPB;
            nacc[c][q] = gl_add_lazy(nacc[c][q], gl_mul(brow[(size_t)c * N + pos], nbuf[pos]));
          }
        __syncthreads();
      }
    }
    for (int c = 0; c < K1; ++c) {
      for (int q = 0; q < PER; ++q) nbuf[tid + q * TPB] = gl_mul(nacc[c][q], tb.n_inv);  // normalize
      __syncthreads();
      lds_ntt_inverse<N, TPB>(nbuf, tb.itw, tid);
      for (int q = 0; q < PER; ++q) {
        const int j = tid + q * TPB;
        acc[c * N + j] += gl_modswitch_to_pow2(nbuf[j]);
      }
      __syncthreads();
    }
  }
  // rotation by -b_hat is applied last on this path (ntt64_bnf_pbs.rs:262-271)
  block_sample_extract<N, K1, TPB>(a, acc, sample, b_hat, true, tid);
}

// Same engine, one thread group per GLWE polynomial: the k+1 forward transforms of a level (one per row) and
// the k+1 inverse transforms (one per column) run side by side, which halves (k = 1) the number of
// barrier-separated stages per CMUX.  Exact arithmetic mod p: identical bits to the kernel above.
template <int N, int K1>
__global__ void __launch_bounds__(K1 *GenericCfg<N>::TPB) pbs_ntt_par_kernel(PbsArgs a, NttTables tb) {
  constexpr int TPB = GenericCfg<N>::TPB, TPBT = K1 * TPB, PER = N / TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  uint64_t *acc = (uint64_t *)smem;          // K1*N torus words
  uint64_t *nbuf = acc + (size_t)K1 * N;     // K1*N field elements (one transform buffer per group)
  const int tid = threadIdx.x;
  const int grp = tid / TPB, lt = tid - grp * TPB;  // my row (forward) / column (inverse), thread inside it
  const uint32_t sample = blockIdx.x;
  if (a.only_flagged != nullptr) {  // the recomputation behind a split-key launch: its flagged ciphertexts only
    if (a.only_flagged[sample] == 0u) return;
    if (tid == 0 && a.recomputed != nullptr) atomicAdd(a.recomputed, 1u);
  }
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N;
  const uint64_t *bsk = (const uint64_t *)a.bsk;
  uint64_t *mybuf = nbuf + (size_t)grp * N;

  // body modulus switch; TPBT need not be a power of two (k = 2), so the reduction is a plain sum
  uint64_t corr = 0;
  if (a.ms_type == 1) {
    uint64_t sh = 0;
    int64_t sd = 0;
    for (uint32_t i = tid; i < a.n; i += TPBT) {
      uint64_t h;
      int64_t d;
      centered_ms_terms(lwe[i], LOG2N2, h, d);
      sh += h;
      sd += d;
    }
    nbuf[tid] = sh;
    nbuf[TPBT + tid] = (uint64_t)sd;
    __syncthreads();
    uint64_t th = 0, td = 0;
    for (int l = 0; l < TPBT; ++l) {
      th += nbuf[l];
      td += nbuf[TPBT + l];
    }
    __syncthreads();
    corr = centered_ms_finish(th, (int64_t)td, LOG2N2);
  }
  const uint32_t b_hat = (uint32_t)modulus_switch(lwe[a.n] + corr, LOG2N2);
  for (uint32_t j = lt; j < (uint32_t)N; j += TPB) acc[grp * N + j] = lut[grp * N + j];
  __syncthreads();

  for (uint32_t i = 0; i < a.n; ++i) {
    const uint32_t a_hat = (uint32_t)modulus_switch(lwe[i], LOG2N2);
    if (a_hat == 0) continue;
    uint64_t nacc[PER];
    for (int q = 0; q < PER; ++q) nacc[q] = 0;
    for (uint32_t idx = 0; idx < a.level; ++idx) {
      for (int q = 0; q < PER; ++q) {
        const uint32_t j = lt + q * TPB;
        const int64_t d = decomp_digit(rot_sub<N>(acc + grp * N, j, a_hat), a.base_log, a.level, idx);
        mybuf[j] = d < 0 ? (uint64_t)d + GL_P : (uint64_t)d;  // ntt64.rs:199-220
      }
      __syncthreads();
      lds_ntt_forward<N, TPB>(mybuf, tb.tw, lt);
      for (int row = 0; row < K1; ++row) {  // column `grp` of the external product
        const uint64_t *brow = bsk + ((((size_t)i * a.level + idx) * K1 + row) * K1 + grp) * N;
        const uint64_t *f = nbuf + (size_t)row * N;
        for (int q = 0; q < PER; ++q) {
          const int pos = lt + q * TPB;
          nacc[q] = gl_add_lazy(nacc[q], gl_mul(brow[pos], f[pos]));
        }
      }
      __syncthreads();
    }
    for (int q = 0; q < PER; ++q) mybuf[lt + q * TPB] = gl_mul(nacc[q], tb.n_inv);  // normalize
    __syncthreads();
    lds_ntt_inverse<N, TPB>(mybuf, tb.itw, lt);
    for (int q = 0; q < PER; ++q) {
      const int j = lt + q * TPB;
      acc[grp * N + j] += gl_modswitch_to_pow2(mybuf[j]);
    }
    __syncthreads();
  }
  // rotation by -b_hat is applied last on this path (ntt64_bnf_pbs.rs:262-271)
  block_sample_extract<N, K1, TPBT>(a, acc, sample, b_hat, true, tid);
}

// ------------------------------------------------------------------------- BSK conversion
// one workgroup per polynomial: torus -> f64 tree order / Goldilocks NTT domain
// (cc/algorithms/lwe_bootstrap_key_conversion.rs:20-150, 367-434)
template <int N>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) bsk_to_fourier_kernel(const uint64_t *src, cplx *dst, FftTables tb,
                                                                           int slot_order) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB;
  HX_DYN_SMEM(smem);
  const FBuf fbuf{(cplx *)smem};
  const int tid = threadIdx.x;
  const uint64_t *p = src + (size_t)blockIdx.x * N;
  for (int j = tid; j < n; j += TPB)
    fbuf[j] = cplx{i64_to_f64((int64_t)p[j]) * 5.421010862427522e-20, i64_to_f64((int64_t)p[j + n]) * 5.421010862427522e-20};
  __syncthreads();
  lds_fft_forward<N, TPB>(fbuf, tb.fwd, tid);
  cplx *o = dst + (size_t)blockIdx.x * n;
  // slot_order: 0 tree order, 1 the N = 2048 throughput kernel's, 2 the N = 1024 one's (bsk_slot)
  for (int j = tid; j < n; j += TPB)
    o[slot_order == 1 ? bsk_slot<2048, 2>(j) : slot_order == 2 ? bsk_slot<1024, 2>(j) : j] = fbuf[j];
}

// Split-key form of the exact engine (pbs_fft_wave.hip, LIMBS mode): key word x -> k = round(x P / 2^64)
// (ntt64.rs:144-160), negated and halved modulo p (the engine recombines 2 S into minus the accumulator: arith.h GL_SPLIT_*), centred into (-P/2, P/2], cut into NTT_SPLIT_LIMBS balanced 16-bit limbs
// kc = sum_m c_m 2^(16 m), c_m in [-2^15, 2^15] — and the polynomial of every limb transformed as INTEGERS (no torus
// scaling).  Workgroup (p, q): limb index q (0 = most significant) of source polynomial p = (i*2 + row)*2 + col
// goes to destination polynomial (i*LIMBS + q)*4 + row*2 + col, in the throughput kernel's slot order.
template <int N>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) bsk_to_split_kernel(const uint64_t *src, cplx *dst, FftTables tb) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB, L = NTT_SPLIT_LIMBS;
  HX_DYN_SMEM(smem);
  const FBuf fbuf{(cplx *)smem};
  const int tid = threadIdx.x;
  const uint64_t *p = src + (size_t)blockIdx.x * N;
  const int m = L - 1 - (int)blockIdx.y;  // limb exponent: value c_m 2^(16 m)
  auto limb = [&](uint64_t x) {
    // -k / 2 mod p: the engine's Horner states carry 2 S, and its registers hold MINUS the accumulator (arith.h GL_SPLIT_*)
    const uint64_t k = gl_modswitch_from_pow2(x), v = gl_half(k ? GL_P - k : 0);
    int64_t kc = v > (GL_P >> 1) ? (int64_t)(v - GL_P) : (int64_t)v;
    int64_t c = 0;
    for (int q = 0; q <= m; ++q) {
      c = q == L - 1 ? kc : (int64_t)(int16_t)(uint16_t)kc;  // the top limb takes what is left (|.| <= 2^15)
      kc = (kc - c) >> 16;
    }
    return (double)(int32_t)c;
  };
  for (int j = tid; j < n; j += TPB) fbuf[j] = cplx{limb(p[j]), limb(p[j + n])};
  __syncthreads();
  lds_fft_forward<N, TPB>(fbuf, tb.fwd, tid);
  const size_t i = blockIdx.x >> 2, rc = blockIdx.x & 3;
  cplx *o = dst + ((i * L + blockIdx.y) * 4 + rc) * n;
  for (int j = tid; j < n; j += TPB) o[bsk_slot<2048, 2>(j)] = fbuf[j];
}

template <int N>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) bsk_to_ntt_kernel(const uint64_t *src, uint64_t *dst, NttTables tb) {
  constexpr int TPB = GenericCfg<N>::TPB;
  HX_DYN_SMEM(smem);
  uint64_t *nbuf = (uint64_t *)smem;
  const int tid = threadIdx.x;
  const uint64_t *p = src + (size_t)blockIdx.x * N;
  for (int j = tid; j < N; j += TPB) nbuf[j] = gl_modswitch_from_pow2(p[j]);
  __syncthreads();
  lds_ntt_forward<N, TPB>(nbuf, tb.tw, tid);
  uint64_t *o = dst + (size_t)blockIdx.x * N;
  for (int j = tid; j < N; j += TPB) o[j] = nbuf[j];
}

// ------------------------------------------------------------------------- launchers
template <int N>
static void launch_fft_big(hipStream_t st, const PbsArgs &a, const FftTables &tb) {  // N >= 8192, k = 1
  HX_PANIC_IF_FALSE(a.acc_scratch != nullptr, "PBS scratch of a polynomial_size >= 8192 set has no accumulator buffer");
  const size_t smem = fbuf_bytes(N);
  hx_set_dynamic_smem_once<pbs_fft_generic_kernel<N, 2, true>>(smem);
  HX_LAUNCH((pbs_fft_generic_kernel<N, 2, true>), dim3(a.num_samples), dim3(GenericCfg<N>::TPB), smem, st, a, tb);
}
template <int N, int K1>
static void launch_fft(hipStream_t st, const PbsArgs &a, const FftTables &tb) {
  // one group per polynomial pays off for k = 1 (43.5k PBS/s at 2_2); with three groups (k = 2, N = 1024) the
  // larger workgroup costs more occupancy than the shorter barrier chain returns (45.9k vs 59.0k): single group
  if (g_ntt_kernel_serial || K1 != 2) {
    const size_t smem = (size_t)K1 * N * 8 + fbuf_bytes(N);
    hx_set_dynamic_smem_once<pbs_fft_generic_kernel<N, K1>>(smem);
    HX_LAUNCH((pbs_fft_generic_kernel<N, K1>), dim3(a.num_samples), dim3(GenericCfg<N>::TPB), smem, st, a, tb);
    return;
  }
  const size_t smem = (size_t)K1 * N * 8 + (size_t)K1 * fbuf_bytes(N);
  hx_set_dynamic_smem_once<pbs_fft_par_kernel<N, K1>>(smem);
  HX_LAUNCH((pbs_fft_par_kernel<N, K1>), dim3(a.num_samples), dim3(K1 * GenericCfg<N>::TPB), smem, st, a, tb);
}
template <int N, int K1>
static void launch_ntt(hipStream_t st, const PbsArgs &a, const NttTables &tb) {
  if (g_ntt_kernel_serial || K1 != 2) {  // same rule as the f64 engine above
    const size_t smem = (size_t)(K1 + 1) * N * 8;
    hx_set_dynamic_smem_once<pbs_ntt_generic_kernel<N, K1>>(smem);
    HX_LAUNCH((pbs_ntt_generic_kernel<N, K1>), dim3(a.num_samples), dim3(GenericCfg<N>::TPB), smem, st, a, tb);
    return;
  }
  const size_t smem = (size_t)2 * K1 * N * 8;
  hx_set_dynamic_smem_once<pbs_ntt_par_kernel<N, K1>>(smem);
  HX_LAUNCH((pbs_ntt_par_kernel<N, K1>), dim3(a.num_samples), dim3(K1 * GenericCfg<N>::TPB), smem, st, a, tb);
}

#define HX_DISPATCH_NK(FN, ...)                                                              \
  do {                                                                                       \
    const uint32_t k1_ = glwe_dim + 1;                                                       \
    bool ok_ = true;                                                                         \
    switch (N) {                                                                             \
      case 256: if (k1_ == 2) FN<256, 2>(__VA_ARGS__); else if (k1_ == 3) FN<256, 3>(__VA_ARGS__); else if (k1_ == 4) FN<256, 4>(__VA_ARGS__); else ok_ = false; break; \
      case 512: if (k1_ == 2) FN<512, 2>(__VA_ARGS__); else if (k1_ == 3) FN<512, 3>(__VA_ARGS__); else if (k1_ == 4) FN<512, 4>(__VA_ARGS__); else ok_ = false; break; \
      case 1024: if (k1_ == 2) FN<1024, 2>(__VA_ARGS__); else if (k1_ == 3) FN<1024, 3>(__VA_ARGS__); else if (k1_ == 4) FN<1024, 4>(__VA_ARGS__); else ok_ = false; break; \
      case 2048: if (k1_ == 2) FN<2048, 2>(__VA_ARGS__); else if (k1_ == 3) FN<2048, 3>(__VA_ARGS__); else ok_ = false; break; \
      case 4096: if (k1_ == 2) FN<4096, 2>(__VA_ARGS__); else ok_ = false; break;             \
      default: ok_ = false;                                                                  \
    }                                                                                        \
    if (!ok_) HX_PANIC("unsupported (polynomial_size=%u, glwe_dimension=%u) for the MI355X PBS", N, glwe_dim); \
  } while (0)

void launch_pbs_fft_generic(hipStream_t st, uint32_t N, uint32_t glwe_dim, const PbsArgs &a, const FftTables &tb) {
  if (N > 4096) {  // f64 engine only, k = 1 (the reference's 3_3 / 4_4 sets)
    HX_PANIC_IF_FALSE(glwe_dim == 1 && (N == 8192 || N == 16384),
                      "unsupported (polynomial_size=%u, glwe_dimension=%u) for the MI355X PBS", N, glwe_dim);
    if (N == 8192) launch_fft_big<8192>(st, a, tb); else launch_fft_big<16384>(st, a, tb);
    return;
  }
  HX_DISPATCH_NK(launch_fft, st, a, tb);
}
template <int N, int K1>
static void launch_exact(hipStream_t st, const PbsArgs &a) {
  const size_t smem = (size_t)(K1 + 1) * N * 8;
  hx_set_dynamic_smem_once<pbs_exact_generic_kernel<N, K1>>(smem);
  HX_LAUNCH((pbs_exact_generic_kernel<N, K1>), dim3(a.num_samples), dim3(GenericCfg<N>::TPB), smem, st, a);
}
void launch_pbs_exact_generic(hipStream_t st, uint32_t N, uint32_t glwe_dim, const PbsArgs &a) {
  HX_DISPATCH_NK(launch_exact, st, a);
}
void launch_pbs_ntt_generic(hipStream_t st, uint32_t N, uint32_t glwe_dim, const PbsArgs &a, const NttTables &tb) {
  HX_DISPATCH_NK(launch_ntt, st, a, tb);
}

template <int N> static void launch_conv_f(hipStream_t st, const uint64_t *src, void *dst, size_t polys, const FftTables &tb, int slot_order) {
  if (fbuf_bytes(N) > 48 * 1024)
    hx_set_dynamic_smem_once<bsk_to_fourier_kernel<N>>(fbuf_bytes(N));
  HX_LAUNCH((bsk_to_fourier_kernel<N>), dim3((unsigned)polys), dim3(GenericCfg<N>::TPB), fbuf_bytes(N), st, src, (cplx *)dst, tb, slot_order);
}
template <int N> static void launch_conv_n(hipStream_t st, const uint64_t *src, void *dst, size_t polys, const NttTables &tb) {
  HX_LAUNCH((bsk_to_ntt_kernel<N>), dim3((unsigned)polys), dim3(GenericCfg<N>::TPB), (size_t)N * 8, st, src, (uint64_t *)dst, tb);
}
#define HX_DISPATCH_N(FN, ...)                                              \
  switch (N) {                                                              \
    case 256: FN<256>(__VA_ARGS__); break;                                  \
    case 512: FN<512>(__VA_ARGS__); break;                                  \
    case 1024: FN<1024>(__VA_ARGS__); break;                                \
    case 2048: FN<2048>(__VA_ARGS__); break;                                \
    case 4096: FN<4096>(__VA_ARGS__); break;                                \
    default: HX_PANIC("unsupported polynomial_size=%u", N);                 \
  }
void launch_bsk_to_fourier(hipStream_t st, uint32_t N, uint32_t glwe_dim, const uint64_t *src_dev, void *dst, size_t polys, const FftTables &tb) {
  // must agree with bsk_slot<N, K1>
  const int slot_order = (N == 2048 && glwe_dim == 1) ? 1 : (N == 1024 && (glwe_dim == 1 || glwe_dim == 2)) ? 2 : 0;
  if (N == 8192) return launch_conv_f<8192>(st, src_dev, dst, polys, tb, slot_order);
  if (N == 16384) return launch_conv_f<16384>(st, src_dev, dst, polys, tb, slot_order);
  HX_DISPATCH_N(launch_conv_f, st, src_dev, dst, polys, tb, slot_order);
}
void launch_bsk_to_split(hipStream_t st, uint32_t N, const uint64_t *src_dev, void *dst, size_t polys, const FftTables &tb) {
  HX_PANIC_IF_FALSE(N == 2048, "split-key exact engine: polynomial_size %u not supported (2048)", N);
  HX_LAUNCH((bsk_to_split_kernel<2048>), dim3((unsigned)polys, NTT_SPLIT_LIMBS), dim3(GenericCfg<2048>::TPB),
            fbuf_bytes(2048), st, src_dev, (cplx *)dst, tb);
}
void launch_bsk_to_ntt(hipStream_t st, uint32_t N, const uint64_t *src_dev, void *dst, size_t polys, const NttTables &tb) {
  HX_DISPATCH_N(launch_conv_n, st, src_dev, dst, polys, tb);
}

}  // namespace tfhe_hip
