// pbs_ntt_crt.hip — the bit-exact 64-bit-prime NTT engine (tfhe-ntt semantics) computed on the FP64 pipes.
//
// cc/algorithms/lwe_programmable_bootstrapping/ntt64_bnf_pbs.rs:208-280,541-705 define the result: per CMUX the
// negacyclic products  sum_rows digit_poly (*) key_poly  are taken modulo the Goldilocks prime P = 2^64 - 2^32 + 1
// (key values k = round(x P / 2^64), digits as signed integers), switched back to 2^64 and added to the
// accumulator.  The value mod P of a product is a property of the integers involved, not of the transform used:
//
//     R = sum_j d_j kc_j   over Z,   |d_j| <= B/2,  kc_j the key value centred into (-P/2, P/2)
//
// satisfies |R| <= (k+1) l N (B/2) (P/2) < 2^97 for the sets this engine accepts, so R is determined by its residues
// modulo two primes p1 p2 > 2^98.99 and  R mod P  is the reference's value.  Both primes are below 2^49.5, and a
// 64-bit modular multiplication costs 27 integer instructions on this machine where a 50-bit one costs six FP64
// instructions (exact product by fma, quotient by a rounded multiplication, remainder by fma) — the engine the
// MI355X is built around.  `pbs_ntt_par_kernel` (pbs_generic.hip) remains the integer-Goldilocks form of the same
// function; both give the oracle's bits (tests/test_backend_parity.py).
//
// Exactness (every double below holds an integer of magnitude < 2^53, every operation on them is exact):
//   mulmod(a, w), |a| <= 2^52, |w| <= p/2:  h = fl(a w), l = a w - h (fma, exact), q = rint(fl(h / p)) differs from
//     a w / p by at most 0.5 + 2^52 (p/2)/p * 3 * 2^-53 = 1.25, h - q p is an integer below 2^52 (fma, exact), so the
//     result r = (h - q p) + l = a w - q p  is exact and |r| <= 1.25 p.
//   reduce(x), |x| <= 2^53: x - p rint(x / p), exact, |r| <= (p + 1) / 2.
//   forward (Cooley-Tukey): a level adds at most 1.25 p to a value; values are reduced after every 4 levels:
//     magnitudes stay below 0.5 p + 5 p = 5.5 p < 2^52 (p < 2^49.5).
//   inverse (Gentleman-Sande, two levels per pass): inputs of a pass <= 1.25 p, sums <= 5 p, products <= 1.25 p; the
//     two pure sums of a 4-point unit are reduced at the end of the pass.
#include "hx.h"
#include "arith.h"
#include "kernels.h"
#include "pbs_common.h"
#include "tables.h"

namespace tfhe_hip {
namespace crt {

template <int Q> HX_DEV double prime() { return Q ? CRT_P2 : CRT_P1; }
template <int Q> HX_DEV double prime_inv() { return Q ? 1.0 / CRT_P2 : 1.0 / CRT_P1; }  // correctly rounded constants

template <int Q>
HX_DEV double mulmod(double a, double w) {
  const double h = a * w;
  const double l = fma(a, w, -h);
  const double q = __builtin_rint(h * prime_inv<Q>());
  return fma(-q, prime<Q>(), h) + l;
}
template <int Q>
HX_DEV double reduce(double x) {
  const double q = __builtin_rint(x * prime_inv<Q>());
  return fma(-q, prime<Q>(), x);
}
template <int Q>
HX_DEV void ct(double &x, double &y, double w) {  // Cooley-Tukey butterfly
  const double m = mulmod<Q>(y, w);
  const double a = x;
  x = a + m;
  y = a - m;
}
template <int Q>
HX_DEV void gs(double &x, double &y, double w) {  // Gentleman-Sande butterfly
  const double a = x, c = y;
  x = a + c;
  y = mulmod<Q>(a - c, w);
}

// same schedule as lds_ntt_forward (pbs_common.h): stages s, s+1 on the 4 points of one thread per barrier
template <int N, int TPB, int Q>
HX_DEV void lds_forward(double *buf, const double *__restrict__ tw, int tid) {
  constexpr int LOGN = ilog2_c(N), UNITS = (N / 4 + TPB - 1) / TPB, PER2 = (N / 2 + TPB - 1) / TPB;
  HX_UNROLL
  for (int s = 0; s + 1 < LOGN; s += 2) {
    HX_OPAQUE(tid);
    const int t = N >> (s + 1), t2 = t >> 1, m = 1 << s, lt2 = LOGN - 2 - s;
    const bool red = ((s + 2) % 4 == 0) && (s + 2 < LOGN);  // 4 levels done since the last reduction
    HX_UNROLL
    for (int q = 0; q < UNITS; ++q) {
      const int u = tid + q * TPB;
      if (N / 4 % TPB == 0 || u < N / 4) {
        const int g = u >> lt2, j = u & (t2 - 1);
        const int p0 = 2 * g * t + j, p1 = p0 + t2, p2 = p0 + t, p3 = p2 + t2;
        double x0 = buf[p0], x1 = buf[p1], x2 = buf[p2], x3 = buf[p3];
        const double wa = tw[m + g], wb0 = tw[2 * m + 2 * g], wb1 = tw[2 * m + 2 * g + 1];
        ct<Q>(x0, x2, wa);
        ct<Q>(x1, x3, wa);
        ct<Q>(x0, x1, wb0);
        ct<Q>(x2, x3, wb1);
        if (red) {
          x0 = reduce<Q>(x0);
          x1 = reduce<Q>(x1);
          x2 = reduce<Q>(x2);
          x3 = reduce<Q>(x3);
        }
        buf[p0] = x0;
        buf[p1] = x1;
        buf[p2] = x2;
        buf[p3] = x3;
      }
      HX_SCHED_FENCE();
    }
    __syncthreads();
  }
  if (LOGN & 1) {  // last stage alone: t = 1, group g = pair index
    HX_OPAQUE(tid);
    HX_UNROLL
    for (int q = 0; q < PER2; ++q) {
      const int b = tid + q * TPB;
      if (N / 2 % TPB == 0 || b < N / 2) {
        double x = buf[2 * b], y = buf[2 * b + 1];
        ct<Q>(x, y, tw[N / 2 + b]);
        buf[2 * b] = x;
        buf[2 * b + 1] = y;
      }
      if (q & 1) HX_SCHED_FENCE();
    }
    __syncthreads();
  }
}
// inputs <= 1.25 p (reduced products); outputs <= 5 p, the caller reduces them
template <int N, int TPB, int Q>
HX_DEV void lds_inverse(double *buf, const double *__restrict__ itw, int tid) {
  constexpr int LOGN = ilog2_c(N), UNITS = (N / 4 + TPB - 1) / TPB, PER2 = (N / 2 + TPB - 1) / TPB;
  HX_UNROLL
  for (int s = 0; s + 1 < LOGN; s += 2) {
    HX_OPAQUE(tid);
    const int t = 1 << s, m = N >> (s + 1);
    const bool last = !(LOGN & 1) && s + 2 >= LOGN;  // nothing follows: the caller reduces
    HX_UNROLL
    for (int q = 0; q < UNITS; ++q) {
      const int u = tid + q * TPB;
      if (N / 4 % TPB == 0 || u < N / 4) {
        const int G = u >> s, j = u & (t - 1);
        const int p0 = 4 * G * t + j, p1 = p0 + t, p2 = p1 + t, p3 = p2 + t;
        double x0 = buf[p0], x1 = buf[p1], x2 = buf[p2], x3 = buf[p3];
        const double wa0 = itw[m + 2 * G], wa1 = itw[m + 2 * G + 1], wb = itw[(m >> 1) + G];
        gs<Q>(x0, x1, wa0);
        gs<Q>(x2, x3, wa1);
        gs<Q>(x0, x2, wb);
        gs<Q>(x1, x3, wb);
        if (!last) {  // the two pure sums of the unit (x2, x3 are products already)
          x0 = reduce<Q>(x0);
          x1 = reduce<Q>(x1);
        }
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
    const double w = itw[1];
    HX_UNROLL
    for (int q = 0; q < PER2; ++q) {
      const int b = tid + q * TPB;
      if (N / 2 % TPB == 0 || b < N / 2) {
        double x = buf[b], y = buf[b + N / 2];
        gs<Q>(x, y, w);
        buf[b] = x;
        buf[b + N / 2] = y;
      }
      if (q & 1) HX_SCHED_FENCE();
    }
    __syncthreads();
  }
}

// R = r1 + p1 t,  t = (r2 - r1) p1^-1 mod p2 centred  (Garner), then R mod P in the Goldilocks field and the switch to
// 2^64 of cc/commons/math/ntt/ntt64.rs:162-177.  r1, r2 centred residues (|r| <= (p+1)/2).
HX_DEV uint64_t crt_to_torus(double r1, double r2) {
  const double t = reduce<1>(mulmod<1>(r2 - r1, CRT_P1_INV_MOD_P2));
  const int64_t ti = (int64_t)t, ri = (int64_t)r1;  // |.| < 2^50
  const uint64_t tg = ti < 0 ? (uint64_t)ti + GL_P : (uint64_t)ti;
  const uint64_t rg = ri < 0 ? (uint64_t)ri + GL_P : (uint64_t)ri;
  const uint64_t v = gl_add(gl_mul(tg, (uint64_t)CRT_P1_U64), rg);
  return gl_modswitch_to_pow2(v);
}

template <int N>
HX_DEV uint64_t rot_sub(const uint64_t *poly, uint32_t j, uint32_t a_hat) {
  bool neg;
  const uint32_t src = monomial_mul_src(j, a_hat, N, neg);
  const uint64_t s = poly[src];
  return (neg ? (uint64_t)0 - s : s) - poly[j];
}

// One workgroup per LWE, one thread group per GLWE polynomial (as pbs_ntt_par_kernel): the k+1 forward transforms
// of a level and the k+1 inverse transforms run side by side; the two primes one after the other through the same
// LDS buffer.  Key: [i][level][row][col][prime][N] doubles, centred residues of (key value) * N^-1 in the transform
// domain (bsk_to_crt_kernel).
template <int N, int K1>
__global__ void __launch_bounds__(K1 *GenericCfg<N>::TPB) pbs_ntt_crt_kernel(PbsArgs a, CrtTables tb) {
  constexpr int TPB = GenericCfg<N>::TPB, TPBT = K1 * TPB, PER = N / TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  uint64_t *acc = (uint64_t *)smem;                    // K1*N torus words
  double *nbuf = (double *)(acc + (size_t)K1 * N);     // K1*N residues (one transform buffer per group)
  const int tid = threadIdx.x;
  const int grp = tid / TPB, lt = tid - grp * TPB;  // my row (forward) / column (inverse), thread inside it
  const uint32_t sample = blockIdx.x;
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N;
  const double *bsk = (const double *)a.bsk;
  double *mybuf = nbuf + (size_t)grp * N;

  // body modulus switch; TPBT need not be a power of two (k = 2), so the reduction is a plain sum
  uint64_t corr = 0;
  if (a.ms_type == 1) {
    uint64_t *red = (uint64_t *)nbuf;
    uint64_t sh = 0;
    int64_t sd = 0;
    for (uint32_t i = tid; i < a.n; i += TPBT) {
      uint64_t h;
      int64_t d;
      centered_ms_terms(lwe[i], LOG2N2, h, d);
      sh += h;
      sd += d;
    }
    red[tid] = sh;
    red[TPBT + tid] = (uint64_t)sd;
    __syncthreads();
    uint64_t th = 0, td = 0;
    for (int l = 0; l < TPBT; ++l) {
      th += red[l];
      td += red[TPBT + l];
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
    double n0[PER], n1[PER];  // column `grp` of the external product, residues mod p1 / p2
    HX_UNROLL
    for (int q = 0; q < PER; ++q) n0[q] = n1[q] = 0.0;
    for (uint32_t idx = 0; idx < a.level; ++idx) {
      double dig[PER];  // the digits of my row at this level: the same integers modulo both primes
      HX_UNROLL
      for (int q = 0; q < PER; ++q) {
        const uint32_t j = lt + q * TPB;
        const int64_t d = decomp_digit(rot_sub<N>(acc + grp * N, j, a_hat), a.base_log, a.level, idx);
        dig[q] = a.base_log <= 31 ? (double)(int32_t)d : i64_to_f64(d);
      }
      const double *klev = bsk + (((size_t)i * a.level + idx) * K1 * K1) * 2 * N;
      // ---- prime 1
      HX_UNROLL
      for (int q = 0; q < PER; ++q) mybuf[lt + q * TPB] = dig[q];
      __syncthreads();
      lds_forward<N, TPB, 0>(mybuf, tb.tw[0], lt);
      for (int row = 0; row < K1; ++row) {
        const double *brow = klev + ((size_t)(row * K1 + grp) * 2 + 0) * N;
        const double *f = nbuf + (size_t)row * N;
        HX_UNROLL
        for (int q = 0; q < PER; ++q) {
          const int pos = lt + q * TPB;
          n0[q] += mulmod<0>(f[pos], brow[pos]);
        }
      }
      HX_UNROLL
      for (int q = 0; q < PER; ++q) n0[q] = reduce<0>(n0[q]);  // at most 0.5 p + (k+1) 1.25 p before
      __syncthreads();
      // ---- prime 2
      HX_UNROLL
      for (int q = 0; q < PER; ++q) mybuf[lt + q * TPB] = dig[q];
      __syncthreads();
      lds_forward<N, TPB, 1>(mybuf, tb.tw[1], lt);
      for (int row = 0; row < K1; ++row) {
        const double *brow = klev + ((size_t)(row * K1 + grp) * 2 + 1) * N;
        const double *f = nbuf + (size_t)row * N;
        HX_UNROLL
        for (int q = 0; q < PER; ++q) {
          const int pos = lt + q * TPB;
          n1[q] += mulmod<1>(f[pos], brow[pos]);
        }
      }
      HX_UNROLL
      for (int q = 0; q < PER; ++q) n1[q] = reduce<1>(n1[q]);
      __syncthreads();
    }
    HX_UNROLL
    for (int q = 0; q < PER; ++q) mybuf[lt + q * TPB] = n0[q];
    __syncthreads();
    lds_inverse<N, TPB, 0>(mybuf, tb.itw[0], lt);
    HX_UNROLL
    for (int q = 0; q < PER; ++q) n0[q] = reduce<0>(mybuf[lt + q * TPB]);
    __syncthreads();
    HX_UNROLL
    for (int q = 0; q < PER; ++q) mybuf[lt + q * TPB] = n1[q];
    __syncthreads();
    lds_inverse<N, TPB, 1>(mybuf, tb.itw[1], lt);
    HX_UNROLL
    for (int q = 0; q < PER; ++q) {
      const int j = lt + q * TPB;
      acc[grp * N + j] += crt_to_torus(n0[q], reduce<1>(mybuf[j]));
    }
    __syncthreads();
  }
  // rotation by -b_hat is applied last on this path (ntt64_bnf_pbs.rs:262-271)
  block_sample_extract<N, K1, TPBT>(a, acc, sample, b_hat, true, tid);
}

// ------------------------------------------------------------------------- key conversion
// cc/algorithms/lwe_bootstrap_key_conversion.rs:367-434 gives the key value k = round(x P / 2^64); this engine
// keeps its centred form modulo p1 and p2 in the transform domain, times N^-1 (the inverse transform's scaling)
template <int N>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) bsk_to_crt_kernel(const uint64_t *src, double *dst, CrtTables tb) {
  constexpr int TPB = GenericCfg<N>::TPB;
  HX_DYN_SMEM(smem);
  double *buf = (double *)smem;
  const int tid = threadIdx.x;
  const uint64_t *p = src + (size_t)blockIdx.x * N;
  double *o = dst + (size_t)blockIdx.x * 2 * N;
  for (int q = 0; q < 2; ++q) {
    const int64_t pq = q ? (int64_t)CRT_P2_U64 : (int64_t)CRT_P1_U64;
    for (int j = tid; j < N; j += TPB) {
      const uint64_t kv = gl_modswitch_from_pow2(p[j]);
      const int64_t kc = kv > (GL_P >> 1) ? (int64_t)(kv - GL_P) : (int64_t)kv;  // (-P/2, P/2)
      int64_t r = kc % pq;                                                          // (-p, p)
      if (r > pq / 2) r -= pq;
      if (r < -(pq / 2)) r += pq;
      buf[j] = (double)r;
    }
    __syncthreads();
    if (q == 0) lds_forward<N, TPB, 0>(buf, tb.tw[0], tid);
    else lds_forward<N, TPB, 1>(buf, tb.tw[1], tid);
    for (int j = tid; j < N; j += TPB)
      o[(size_t)q * N + j] = q ? reduce<1>(mulmod<1>(buf[j], tb.n_inv[1])) : reduce<0>(mulmod<0>(buf[j], tb.n_inv[0]));
    __syncthreads();
  }
}

template <int N, int K1>
static void launch_crt(hipStream_t st, const PbsArgs &a, const CrtTables &tb) {
  const size_t smem = (size_t)K1 * N * 8 * 2;
  hx_set_dynamic_smem_once<pbs_ntt_crt_kernel<N, K1>>(smem);
  HX_LAUNCH((pbs_ntt_crt_kernel<N, K1>), dim3(a.num_samples), dim3(K1 * GenericCfg<N>::TPB), smem, st, a, tb);
}
template <int N>
static void launch_conv(hipStream_t st, const uint64_t *src, void *dst, size_t polys, const CrtTables &tb) {
  HX_LAUNCH((bsk_to_crt_kernel<N>), dim3((unsigned)polys), dim3(GenericCfg<N>::TPB), (size_t)N * 8, st, src, (double *)dst, tb);
}

}  // namespace crt

// |R| <= (k+1) l N 2^(base_log-1) 2^63 must stay below p1 p2 / 2 (2^97.99...): log2((k+1) l N) + base_log <= 35,
// on the (N, k) the kernel is instantiated for
bool pbs_ntt_crt_supported(uint32_t N, uint32_t glwe_dim, uint32_t level, uint32_t base_log) {
  if (!(N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096)) return false;
  const uint32_t k1 = glwe_dim + 1;
  if (k1 < 2 || k1 > (N <= 1024 ? 4u : N == 2048 ? 3u : 2u)) return false;
  const uint64_t terms = (uint64_t)k1 * level * N;
  uint32_t lg = 0;
  while (((uint64_t)1 << lg) < terms) ++lg;  // ceil(log2(terms))
  return base_log >= 1 && lg + base_log <= 35;
}

#define HX_DISPATCH_NK_CRT(FN, ...)                                                                                     \
  do {                                                                                                                \
    const uint32_t k1_ = glwe_dim + 1;                                                                                \
    switch (N) {                                                                                                      \
      case 256: if (k1_ == 2) FN<256, 2>(__VA_ARGS__); else if (k1_ == 3) FN<256, 3>(__VA_ARGS__); else FN<256, 4>(__VA_ARGS__); break; \
      case 512: if (k1_ == 2) FN<512, 2>(__VA_ARGS__); else if (k1_ == 3) FN<512, 3>(__VA_ARGS__); else FN<512, 4>(__VA_ARGS__); break; \
      case 1024: if (k1_ == 2) FN<1024, 2>(__VA_ARGS__); else if (k1_ == 3) FN<1024, 3>(__VA_ARGS__); else FN<1024, 4>(__VA_ARGS__); break; \
      case 2048: if (k1_ == 2) FN<2048, 2>(__VA_ARGS__); else FN<2048, 3>(__VA_ARGS__); break;                         \
      default: FN<4096, 2>(__VA_ARGS__);                                                                              \
    }                                                                                                                 \
  } while (0)

void launch_pbs_ntt_crt(hipStream_t st, uint32_t N, uint32_t glwe_dim, const PbsArgs &a, const CrtTables &tb) {
  HX_PANIC_IF_FALSE(pbs_ntt_crt_supported(N, glwe_dim, a.level, a.base_log),
                    "parameter set outside the two-prime NTT engine (polynomial_size=%u, glwe_dimension=%u, level=%u, "
                    "base_log=%u)", N, glwe_dim, a.level, a.base_log);
  HX_DISPATCH_NK_CRT(crt::launch_crt, st, a, tb);
}
void launch_bsk_to_crt(hipStream_t st, uint32_t N, const uint64_t *src_dev, void *dst, size_t polys, const CrtTables &tb) {
  switch (N) {
    case 256: crt::launch_conv<256>(st, src_dev, dst, polys, tb); break;
    case 512: crt::launch_conv<512>(st, src_dev, dst, polys, tb); break;
    case 1024: crt::launch_conv<1024>(st, src_dev, dst, polys, tb); break;
    case 2048: crt::launch_conv<2048>(st, src_dev, dst, polys, tb); break;
    case 4096: crt::launch_conv<4096>(st, src_dev, dst, polys, tb); break;
    default: HX_PANIC("unsupported polynomial_size=%u for the two-prime NTT engine", N);
  }
}

}  // namespace tfhe_hip
