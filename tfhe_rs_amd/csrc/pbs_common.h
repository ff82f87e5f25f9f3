// pbs_common.h — pieces shared by the PBS kernels: launch arguments, the LDS-resident
// reference transforms (one workgroup per polynomial, barrier per stage), prologue/epilogue.
#pragma once
#include "arith.h"
#include "tables.h"

namespace tfhe_hip {

// Arguments of one batched PBS launch.  Index conventions are those of the interface we
// replace: lwe_in / lwe_out / lut are flat lists, *_idx are u64 arrays on the device
// (backends/tfhe-cuda-backend/cuda/src/pbs/programmable_bootstrap_classic.cuh:821-826);
// many-LUT output t lives at offset t*num_samples*(k*N+1) and extracts coefficient
// t*lut_stride (:990-1001).
struct PbsArgs {
  uint64_t *lwe_out;
  const uint64_t *out_idx;
  const uint64_t *lut;
  const uint64_t *lut_idx;
  const uint64_t *lwe_in;
  const uint64_t *in_idx;
  const void *bsk;
  uint32_t n;          // input (small) LWE dimension
  uint32_t base_log;
  uint32_t level;
  uint32_t num_samples;
  uint32_t num_many_lut;
  uint32_t lut_stride;
  uint32_t ms_type;    // PBS_MS_REDUCTION_T: 0 none, 1 centered
  // multi-bit PBS only (0 otherwise): grouping factor; bsk is then the Fourier-domain multi-bit key
  uint32_t grouping = 0;
  // N >= 8192 only: (k+1) N torus words per sample — the accumulator of those rings does not fit in LDS next
  // to the transform buffer and lives in device memory (L2-resident between the iterations of a workgroup)
  uint64_t *acc_scratch = nullptr;
  // exact engine, split-key form: set to 1 by a lane whose f64 product was not within 1/4 of an integer (the round-off
  // check of an FFT-based exact multiplication); read back by hip_programmable_bootstrap_ntt64_split_roundoff_status
  uint32_t *roundoff_flag = nullptr;
  // ... and, per sample of the launch, a word that the same lane sets to 1 (null: not recorded).  The launch that follows on
  // the same stream — the integer Goldilocks kernel with `only_flagged` pointing at these words and the NTT-domain key —
  // recomputes exactly those ciphertexts (a workgroup whose word is 0 returns at once) and counts them in `recomputed`:
  // the split-key entry point is exact whatever the data (ntt64_bnf_pbs.rs:208-280 is what both compute)
  uint32_t *bad_samples = nullptr;
  const uint32_t *only_flagged = nullptr;
  uint32_t *recomputed = nullptr;
  // multi-bit throughput kernel only: 8 progress counters (one per XCD, 128 bytes apart), zeroed by the launch —
  // the workgroups of an XCD stay within a few groups of each other so that a group's key is fetched from HBM
  // once per XCD instead of once per workgroup
  uint32_t *pace = nullptr;
  // multi-bit PBS of the reference's noise tests only (cuda_multi_bit_programmable_bootstrap_noise_tests_64_async): the
  // subset degrees of every group, already modulus switched by cuda_modulus_switch_multi_bit_64_async, as 2^g words per
  // group ([group][subset]) per input — the keybundle reads them instead of switching the mask sums itself
  // (programmable_bootstrap_multibit.cuh:85-107)
  const uint64_t *mb_degrees = nullptr;
  // Throughput kernel for N = 2048, k = 1 only (null otherwise): the sample extraction ALSO writes the keyswitch
  // operands of its output — for every mask word the shifted digits d + B/2 of the keyswitch that will read this
  // ciphertext next, as bytes in the A-operand layout of ks_gemm_kernel ([tile of 32 samples][step of 32 k][lane =
  // (k half, row)][16 bytes], row = this launch's sample index), and per sample the sum of those bytes — so that the
  // next keyswitch skips its digit pass (keyswitch.hip).  emit_level_pad = 4 or 8, emit_base_log * emit_level <= 30.
  int8_t *emit_a = nullptr;
  int32_t *emit_suma = nullptr;
  uint32_t emit_base_log = 0, emit_level = 0, emit_level_pad = 0, emit_steps = 0;
  // host side only (kernel selection of THIS call; comparison choices of hip_backend_set_fft_kernel): carried in the
  // arguments, not in globals, so that concurrent host threads on different streams cannot change each other's kernel
  bool mb_no_share = false;          // choice 7: every wave pair loads its own key
  bool mb_no_octet = false;          // choice 8: quads of waves share the key loads even where the whole workgroup could
  bool mb_generic_products = false;  // choice 6: latency path with the products on the generic kernels
};

constexpr int ilog2_c(int x) { return x <= 1 ? 0 : 1 + ilog2_c(x / 2); }
template <int N> struct GenericCfg {
  static constexpr int TPB = N > 8192 ? 1024 : N > 4096 ? 512 : (N / 4 < 256) ? N / 4 : 256;
};

// Storage slot of transform position `pos` inside one Fourier-domain key polynomial.  The
// key buffer is opaque to callers (gpu/entities/lwe_bootstrap_key.rs:57-104), so for the
// headline ring (N = 2048, k = 1) it is stored in the order the throughput kernel reads it:
// lane L of a wave owns positions L*16..L*16+15 and element r of every lane is contiguous.
template <int N, int K1>
HX_DEV int bsk_slot(int pos) {
  if constexpr (N == 2048 && K1 == 2) return (pos & 15) * 64 + (pos >> 4);
  if constexpr (N == 1024 && (K1 == 2 || K1 == 3)) return (pos & 7) * 64 + (pos >> 3);  // pbs_fft_wave3.hip, layout LC
  return pos;
}

// ---------------------------------------------------------------- multi-bit: Fourier-domain keybundle
// The CPU reference keeps the multi-bit key in the Fourier domain and combines the 2^g GGSWs of a group there
// (cc/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:116-156):
//     GGSW_comb = GGSW_0 + sum_{s >= 1} GGSW_s (.) FFT(X^{deg_s})
// In the transform order of DESIGN.md §4 (position p <-> zeta^(1 + 4 bitrev p)) the monomial's value at
// position p is the product of one table entry per 16 positions and a 16th root of unity — which is how a lane
// that owns 16 consecutive positions gets its 16 factors from one gathered table entry.  Spec (bit-exact
// across the oracle and every kernel): M_d[p] = cmul_first(mono[jb], mono[jw]),
//     jb = ((1 + 4 bitrev_{L-4}(p >> 4)) d) mod 2N,  jw = (N/8) ((bitrev_4(p & 15) d) mod 16),  L = log2(N/2);
// KB <- K_0, then KB <- cmul_add(K_s, M_{deg_s}, KB) for s = 1 .. 2^g - 1 in this order.
template <int N>
HX_DEV uint32_t monomial_base_index(uint32_t hi /* p >> 4 */, uint32_t deg) {
  constexpr int L4 = ilog2_c(N / 2) - 4;
  const uint32_t br = L4 ? (__brev(hi) >> (32 - (L4 ? L4 : 1))) : 0u;
  return ((1u + 4u * br) * deg) & (2u * N - 1u);
}
template <int N>
HX_DEV uint32_t monomial_root_index(uint32_t lo /* p & 15 */, uint32_t deg) {
  return (uint32_t)(N / 8) * (((__brev(lo) >> 28) * deg) & 15u);
}
template <int N>
HX_DEV cplx monomial_factor(const double *__restrict__ mono, uint32_t pos, uint32_t deg) {
  const uint32_t jb = monomial_base_index<N>(pos >> 4, deg), jw = monomial_root_index<N>(pos & 15u, deg);
  return cmul_first(cplx{mono[2 * jb], mono[2 * jb + 1]}, cplx{mono[2 * jw], mono[2 * jw + 1]});
}
// subset degrees of one group (:30-65): subset s selects mask element m when bit (g-1-m) of s is set; the
// sum wraps BEFORE the modulus switch
HX_DEV void multi_bit_degrees(const uint64_t *__restrict__ group_mask, uint32_t g, uint32_t log2_2n, uint32_t *deg) {
  const uint32_t per = 1u << g;
  for (uint32_t s = 1; s < per; ++s) {
    uint64_t sum = 0;
    for (uint32_t m = 0; m < g; ++m)
      if ((s >> (g - 1 - m)) & 1) sum += group_mask[m];
    deg[s] = (uint32_t)modulus_switch(sum, log2_2n);
  }
}

// ---------------------------------------------------------------- LDS transforms (generic)
// Transform buffer of n = N/2 complex points with one spare 16-byte slot after every 16: the late
// (forward) / early (inverse) passes touch points 4 or 16 apart, which unpadded is a 4-way bank conflict
// on every access (39 % of the LDS time of the N = 1024 kernel before the padding).
struct FBuf {
  cplx *p;
  HX_DEV cplx &operator[](int q) const { return p[q + (q >> 4)]; }
};
constexpr size_t fbuf_slots(int N) { return (size_t)(N / 2 + N / 32); }
constexpr size_t fbuf_bytes(int N) { return fbuf_slots(N) * 16; }
// forward: DESIGN.md §4 merged-twist tree, in place over buf[0..n).  Two radix-2 stages per barrier: a
// thread takes the 4 points {P, P + m/4, P + m/2, P + 3m/4} of a group through stage m and stage m/2 (the
// same butterflies with the same twiddles as the stage-per-barrier form, hence the same bits); an odd
// stage count ends with one radix-2 stage.
template <int N, int TPB>
HX_DEV void lds_fft_forward(FBuf buf, const double *__restrict__ fwd, int tid) {
  constexpr int n = N / 2;
  int m = n, cnt = 1;
  for (; m >= 4; m >>= 2, cnt <<= 2) {
    const int quarter = m >> 2;
    for (int u = tid; u < n / 4; u += TPB) {
      const int g = u / quarter, j = u - g * quarter;
      const int p0 = g * m + j, p1 = p0 + quarter, p2 = p1 + quarter, p3 = p2 + quarter;
      const cplx wa{fwd[2 * (cnt + g)], fwd[2 * (cnt + g) + 1]};
      const cplx wb0{fwd[2 * (2 * cnt + 2 * g)], fwd[2 * (2 * cnt + 2 * g) + 1]};
      const cplx wb1{fwd[2 * (2 * cnt + 2 * g + 1)], fwd[2 * (2 * cnt + 2 * g + 1) + 1]};
      cplx x0 = buf[p0], x1 = buf[p1], x2 = buf[p2], x3 = buf[p3];
      bfly(x0, x2, wa);
      bfly(x1, x3, wa);
      bfly(x0, x1, wb0);
      bfly(x2, x3, wb1);
      buf[p0] = x0;
      buf[p1] = x1;
      buf[p2] = x2;
      buf[p3] = x3;
    }
    __syncthreads();
  }
  if (m == 2) {  // odd number of stages: the last one alone
    for (int b = tid; b < n / 2; b += TPB) {
      const int p0 = 2 * b, p1 = p0 + 1;
      const cplx s{fwd[2 * (cnt + b)], fwd[2 * (cnt + b) + 1]};
      cplx x = buf[p0], y = buf[p1];
      bfly(x, y, s);
      buf[p0] = x;
      buf[p1] = y;
    }
    __syncthreads();
  }
}
// one radix-2 stage of the backward transform on (x, y) = (position j of its group, j + half):
// stages half = 1, 2 have trivial twiddles (1, -i) and use plain additions
HX_DEV void inv_bfly(cplx &x, cplx &y, int half, int j, const double *__restrict__ inv) {
  if (half == 1 || (half == 2 && j == 0)) {
    const cplx o1{x.re + y.re, x.im + y.im}, o2{x.re - y.re, x.im - y.im};
    x = o1;
    y = o2;
  } else if (half == 2) {  // w = -i
    const cplx o1{x.re + y.im, x.im - y.re}, o2{x.re - y.im, x.im + y.re};
    x = o1;
    y = o2;
  } else {
    bfly(x, y, cplx{inv[2 * (half + j)], inv[2 * (half + j) + 1]});
  }
}
// backward (unnormalised DIT over the tree order), two stages (half, 2 half) per barrier
template <int N, int TPB>
HX_DEV void lds_fft_inverse(FBuf buf, const double *__restrict__ inv, int tid) {
  constexpr int n = N / 2;
  int half = 1;
  for (; 4 * half <= n; half <<= 2) {
    for (int u = tid; u < n / 4; u += TPB) {
      const int q = u / half, j = u - q * half;
      const int p0 = q * 4 * half + j, p1 = p0 + half, p2 = p1 + half, p3 = p2 + half;
      cplx x0 = buf[p0], x1 = buf[p1], x2 = buf[p2], x3 = buf[p3];
      inv_bfly(x0, x1, half, j, inv);
      inv_bfly(x2, x3, half, j, inv);
      inv_bfly(x0, x2, 2 * half, j, inv);
      inv_bfly(x1, x3, 2 * half, j + half, inv);
      buf[p0] = x0;
      buf[p1] = x1;
      buf[p2] = x2;
      buf[p3] = x3;
    }
    __syncthreads();
  }
  if (2 * half <= n) {  // odd number of stages: the last one alone
    for (int b = tid; b < n / 2; b += TPB) {
      cplx x = buf[b], y = buf[b + half];
      inv_bfly(x, y, half, b, inv);
      buf[b] = x;
      buf[b + half] = y;
    }
    __syncthreads();
  }
}
// Goldilocks negacyclic NTT, Cooley–Tukey / Gentleman–Sande (tfhe-ntt generic_solinas.rs:449-514).
// Stage loops are unrolled (strides and table offsets become immediates).  Forward: values stay lazy
// between stages (every addition has the canonical product as its second operand) and are made
// canonical once at the end; inverse: sums are made canonical, differences go lazily into the product.
HX_DEV void ntt_ct(uint64_t &x, uint64_t &y, uint64_t w) {  // Cooley–Tukey butterfly, lazy in, lazy out
  const uint64_t zw = gl_mul(y, w);
  const uint64_t a = x;
  x = gl_add_lazy(a, zw);
  y = gl_sub_lazy(a, zw);
}
HX_DEV void ntt_gs(uint64_t &x, uint64_t &y, uint64_t w) {  // Gentleman–Sande butterfly, canonical in and out
  const uint64_t a = x, c = y;
  x = gl_canon(gl_add_lazy(a, c));
  y = gl_mul(gl_sub_lazy(a, c), w);
}
template <int N, int TPB>
HX_DEV void lds_ntt_forward(uint64_t *buf, const uint64_t *__restrict__ tw, int tid) {
  constexpr int LOGN = ilog2_c(N), UNITS = (N / 4 + TPB - 1) / TPB, PER2 = (N / 2 + TPB - 1) / TPB;
  HX_UNROLL
  for (int s = 0; s + 1 < LOGN; s += 2) {  // stages s and s + 1 on the 4 points of one thread
    HX_OPAQUE(tid);  // addresses are recomputed per pass instead of being kept (or spilled) across the caller's loop
    const int t = N >> (s + 1), t2 = t >> 1, m = 1 << s, lt2 = LOGN - 2 - s;
    HX_UNROLL
    for (int q = 0; q < UNITS; ++q) {
      const int u = tid + q * TPB;
      if (N / 4 % TPB == 0 || u < N / 4) {
        const int g = u >> lt2, j = u & (t2 - 1);
        const int p0 = 2 * g * t + j, p1 = p0 + t2, p2 = p0 + t, p3 = p2 + t2;
        uint64_t x0 = buf[p0], x1 = buf[p1], x2 = buf[p2], x3 = buf[p3];
        const uint64_t wa = tw[m + g], wb0 = tw[2 * m + 2 * g], wb1 = tw[2 * m + 2 * g + 1];
        ntt_ct(x0, x2, wa);
        ntt_ct(x1, x3, wa);
        ntt_ct(x0, x1, wb0);
        ntt_ct(x2, x3, wb1);
        buf[p0] = x0;
        buf[p1] = x1;
        buf[p2] = x2;
        buf[p3] = x3;
      }
      HX_SCHED_FENCE();  // one unit in flight: bounds the registers the unrolled body may take
    }
    __syncthreads();
  }
  if (LOGN & 1) {  // last stage alone: t = 1, group g = pair index
    HX_OPAQUE(tid);
    HX_UNROLL
    for (int q = 0; q < PER2; ++q) {
      const int b = tid + q * TPB;
      if (N / 2 % TPB == 0 || b < N / 2) {
        uint64_t x = buf[2 * b], y = buf[2 * b + 1];
        ntt_ct(x, y, tw[N / 2 + b]);
        buf[2 * b] = x;
        buf[2 * b + 1] = y;
      }
      if (q & 1) HX_SCHED_FENCE();
    }
    __syncthreads();
  }
  for (int j = tid; j < N; j += TPB) buf[j] = gl_canon(buf[j]);
  __syncthreads();
}
template <int N, int TPB>
HX_DEV void lds_ntt_inverse(uint64_t *buf, const uint64_t *__restrict__ itw, int tid) {
  constexpr int LOGN = ilog2_c(N), UNITS = (N / 4 + TPB - 1) / TPB, PER2 = (N / 2 + TPB - 1) / TPB;
  HX_UNROLL
  for (int s = 0; s + 1 < LOGN; s += 2) {
    HX_OPAQUE(tid);
    const int t = 1 << s, m = N >> (s + 1);
    HX_UNROLL
    for (int q = 0; q < UNITS; ++q) {
      const int u = tid + q * TPB;
      if (N / 4 % TPB == 0 || u < N / 4) {
        const int G = u >> s, j = u & (t - 1);
        const int p0 = 4 * G * t + j, p1 = p0 + t, p2 = p1 + t, p3 = p2 + t;
        uint64_t x0 = buf[p0], x1 = buf[p1], x2 = buf[p2], x3 = buf[p3];
        const uint64_t wa0 = itw[m + 2 * G], wa1 = itw[m + 2 * G + 1], wb = itw[(m >> 1) + G];
        ntt_gs(x0, x1, wa0);
        ntt_gs(x2, x3, wa1);
        ntt_gs(x0, x2, wb);
        ntt_gs(x1, x3, wb);
        buf[p0] = x0;
        buf[p1] = x1;
        buf[p2] = x2;
        buf[p3] = x3;
      }
      HX_SCHED_FENCE();
    }
    __syncthreads();
  }
  if (LOGN & 1) {  // last stage alone: t = N / 2, one group, twiddle itw[1]
    HX_OPAQUE(tid);
    const uint64_t w = itw[1];
    HX_UNROLL
    for (int q = 0; q < PER2; ++q) {
      const int b = tid + q * TPB;
      if (N / 2 % TPB == 0 || b < N / 2) {
        uint64_t x = buf[b], y = buf[b + N / 2];
        ntt_gs(x, y, w);
        buf[b] = x;
        buf[b + N / 2] = y;
      }
      if (q & 1) HX_SCHED_FENCE();
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- prologue / epilogue
// b_hat = ms(body + correction); the centered correction is an exact integer reduction
// over the mask (cc/algorithms/modulus_switch.rs:57-103).  scratch: 2*TPB u64 in LDS.
template <int TPB>
HX_DEV uint32_t block_body_modulus_switch(const uint64_t *__restrict__ lwe, uint32_t n, uint32_t log_modulus,
                                          uint32_t ms_type, uint64_t *scratch, int tid) {
  uint64_t corr = 0;
  if (ms_type == 1) {
    uint64_t sh = 0;
    int64_t sd = 0;
    for (uint32_t i = tid; i < n; i += TPB) {
      uint64_t h;
      int64_t d;
      centered_ms_terms(lwe[i], log_modulus, h, d);
      sh += h;
      sd += d;
    }
    scratch[tid] = sh;
    scratch[TPB + tid] = (uint64_t)sd;
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
      if (tid < s) {
        scratch[tid] += scratch[tid + s];
        scratch[TPB + tid] += scratch[TPB + tid + s];
      }
      __syncthreads();
    }
    corr = centered_ms_finish(scratch[0], (int64_t)scratch[TPB], log_modulus);
    __syncthreads();
  }
  return (uint32_t)modulus_switch(lwe[n] + corr, log_modulus);
}

// sample extraction of coefficient nth, optionally through a final division by X^{post_div}
// (the NTT-bnf order rotates by -b_hat last).  cc/algorithms/glwe_sample_extraction.rs:119-146
template <int N, int K1, int TPB>
HX_DEV void block_sample_extract(const PbsArgs &a, const uint64_t *acc, uint32_t sample, uint32_t post_div,
                                 bool use_post_div, int tid) {
  constexpr int k = K1 - 1;
  const size_t out_sz = (size_t)k * N + 1;
  auto coeff = [&](int p, uint32_t j) -> uint64_t {
    if (!use_post_div) return acc[p * N + j];
    bool neg;
    const uint32_t src = monomial_div_src(j, post_div, N, neg);
    const uint64_t v = acc[p * N + src];
    return neg ? (uint64_t)0 - v : v;
  };
  for (uint32_t t = 0; t < a.num_many_lut; ++t) {
    const uint32_t nth = t * a.lut_stride;
    uint64_t *out = a.lwe_out + (size_t)t * a.num_samples * out_sz + (size_t)a.out_idx[sample] * out_sz;
    for (int p = 0; p < k; ++p)
      for (uint32_t j = tid; j < (uint32_t)N; j += TPB)
        out[(size_t)p * N + j] = (j <= nth) ? coeff(p, nth - j) : (uint64_t)0 - coeff(p, N + nth - j);
    if (tid == 0) out[(size_t)k * N] = coeff(k, nth);
  }
}

}  // namespace tfhe_hip
