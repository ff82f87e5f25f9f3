// pbs_ref64.hip — the reference's OWN f64 arithmetic on the MI355X: a verification engine (like the exact
// engine) that runs the blind rotation with tfhe-fft's radix-4 decimation-in-frequency Stockham transform and
// the reference's x86 conversion / multiply-accumulate forms, operation for operation, so that its outputs
// are the bytes of the reference's f64 golden vectors (apps/test-vectors/data/*/lwe_after_{id,spec}_pbs.cbor,
// generated with `experimental-force_fft_algo_dif4`, apps/test-vectors/Cargo.toml:11).
//
// What it establishes: IEEE f64 arithmetic on gfx950 (v_fma_f64, v_mul_f64, v_add_f64, v_rndne_f64) gives the
// reference CPU's bits when the operations are issued in the reference's order; the production kernels
// (pbs_fft_wave*.hip, pbs_fft_block.hip, pbs_generic.hip) differ from this engine only in the ORDER of their
// butterflies and in algebraically neutral fusions, and are compared with it by phase.
//
// Restated (paths in the tfhe-rs tree):
//   tfhe-fft/src/dif4.rs:111-163,185-236, dif2.rs:100-140   radix-4 / radix-2 Stockham passes
//   tfhe-fft/src/x86.rs:47-55                                complex multiply  fmaddsub(aa, xy, bb*yx)
//   tfhe/src/core_crypto/fft_impl/fft64/math/fft/mod.rs:201-222      convert_forward_torus (no FMA)
//   .../fft/x86.rs:414-500                                   convert_forward_integer (fmsub / fmadd)
//   .../fft/x86.rs:743-790,893-960                           convert_add_backward_torus
//   .../fft64/crypto/ggsw.rs:483-697                         add_external_product_assign / update_with_fmadd
//   .../fft64/crypto/bootstrap.rs:294-365                    blind_rotate_assign
// One workgroup per LWE, everything in LDS, a barrier per pass: built for agreement, not for speed.
#include "kernels.h"

namespace tfhe_hip {

HX_DEV cplx ref_mul(const cplx a, const cplx b) {  // FftSimd::mul / pulp mul_c64s
  return cplx{fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)};
}
HX_DEV cplx ref_mul_add(const cplx a, const cplx b, const cplx c) {  // pulp mul_add_c64s
  return cplx{fma(a.re, b.re, -fma(a.im, b.im, -c.re)), fma(a.re, b.im, fma(a.im, b.re, c.im))};
}
HX_DEV cplx ref_mul_j(bool fwd, const cplx z) { return fwd ? cplx{-z.im, z.re} : cplx{z.im, -z.re}; }
HX_DEV cplx cadd(const cplx a, const cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
HX_DEV cplx csub(const cplx a, const cplx b) { return cplx{a.re - b.re, a.im - b.im}; }

// ordered radix-4 DIF Stockham transform of n points: data in `x`, scratch `y`, result in `x`.
// tw: [w_init (n) | w (n)] of RefTables (forward) or its conjugate table (inverse)
template <int N, int TPB>
HX_DEV void ref_transform(cplx *x, cplx *y, const cplx *__restrict__ tw, bool fwd, int tid) {
  constexpr int n = N / 2;
  const cplx *w = tw + n;
  cplx *src = x, *dst = y;
  int s = 1, m = n;
  for (; m > 4; m >>= 2, s <<= 2) {
    for (int u = tid; u < n / 4; u += TPB) {
      const int q = u / s, j = u - q * s;
      const cplx a = src[u], b = src[n / 4 + u], c = src[n / 2 + u], d = src[3 * (n / 4) + u];
      const cplx w1 = w[4 * q * s + 1], w2 = w[4 * q * s + 2], w3 = w[4 * q * s + 3];
      const cplx apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), jbmd = ref_mul_j(fwd, csub(b, d));
      cplx *o = dst + (size_t)q * 4 * s + j;
      o[0] = cadd(apc, bpd);
      o[s] = ref_mul(w1, csub(amc, jbmd));
      o[2 * s] = ref_mul(w2, csub(apc, bpd));
      o[3 * s] = ref_mul(w3, cadd(amc, jbmd));
    }
    __syncthreads();
    cplx *t = src;
    src = dst;
    dst = t;
  }
  // tail: reads `src`, the result must land in `x`
  if (m == 4) {
    cplx r[(n / 4 + TPB - 1) / TPB][4];
    int c0 = 0;
    for (int u = tid; u < n / 4; u += TPB, ++c0) {
      const cplx a = src[u], b = src[n / 4 + u], c = src[n / 2 + u], d = src[3 * (n / 4) + u];
      const cplx apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), jbmd = ref_mul_j(fwd, csub(b, d));
      r[c0][0] = cadd(apc, bpd);
      r[c0][1] = csub(amc, jbmd);
      r[c0][2] = csub(apc, bpd);
      r[c0][3] = cadd(amc, jbmd);
    }
    __syncthreads();
    c0 = 0;
    for (int u = tid; u < n / 4; u += TPB, ++c0) {
      x[u] = r[c0][0];
      x[n / 4 + u] = r[c0][1];
      x[n / 2 + u] = r[c0][2];
      x[3 * (n / 4) + u] = r[c0][3];
    }
  } else {  // m == 2
    cplx r[(n / 2 + TPB - 1) / TPB][2];
    int c0 = 0;
    for (int u = tid; u < n / 2; u += TPB, ++c0) {
      const cplx a = src[u], b = src[n / 2 + u];
      r[c0][0] = cadd(a, b);
      r[c0][1] = csub(a, b);
    }
    __syncthreads();
    c0 = 0;
    for (int u = tid; u < n / 2; u += TPB, ++c0) {
      x[u] = r[c0][0];
      x[n / 2 + u] = r[c0][1];
    }
  }
  __syncthreads();
}

// key conversion (par_convert_standard_lwe_bootstrap_key_to_fourier): one workgroup per polynomial
template <int N>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) bsk_to_ref64_kernel(const uint64_t *src, cplx *dst, RefTables tb) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB;
  HX_DYN_SMEM(smem);
  cplx *x = (cplx *)smem, *y = x + n;
  const int tid = threadIdx.x;
  const uint64_t *poly = src + (size_t)blockIdx.x * N;
  for (int i = tid; i < n; i += TPB) {
    const double re = (double)(int64_t)poly[i] * 5.421010862427522e-20, im = (double)(int64_t)poly[i + n] * 5.421010862427522e-20;
    const double wr = tb.twist[2 * i], wi = tb.twist[2 * i + 1];
    x[i] = cplx{re * wr - im * wi, re * wi + im * wr};  // num_complex Mul: no fused operation
  }
  __syncthreads();
  ref_transform<N, TPB>(x, y, (const cplx *)tb.w, true, tid);
  for (int i = tid; i < n; i += TPB) dst[(size_t)blockIdx.x * n + i] = x[i];
}

template <int N, int K1>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) pbs_ref64_kernel(PbsArgs a, RefTables tb) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB, PER = n / TPB, LOG2N2 = ilog2_c(2 * N);
  HX_DYN_SMEM(smem);
  uint64_t *acc = (uint64_t *)smem;                 // K1*N torus words
  cplx *x = (cplx *)(smem + (size_t)K1 * N * 8);    // transform buffer and its Stockham scratch
  cplx *y = x + n;
  const int tid = threadIdx.x;
  const uint32_t sample = blockIdx.x;
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N;
  const cplx *bsk = (const cplx *)a.bsk;  // [i][level][row][col][n], transform output order (natural)

  const uint32_t b_hat = block_body_modulus_switch<TPB>(lwe, a.n, LOG2N2, a.ms_type, (uint64_t *)x, tid);
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
    cplx facc[K1][PER];
    bool uninit = true;
    for (uint32_t idx = 0; idx < a.level; ++idx)
      for (int row = 0; row < K1; ++row) {
        // digits of ct1 = acc*X^a_hat - acc, to f64, times the twist (fmsub / fmadd, fft/x86.rs:471-472)
        for (int q = 0; q < PER; ++q) {
          const uint32_t j = tid + q * TPB;
          bool neg;
          uint32_t src = monomial_mul_src(j, a_hat, N, neg);
          uint64_t s0 = acc[row * N + src];
          const uint64_t c0 = (neg ? (uint64_t)0 - s0 : s0) - acc[row * N + j];
          src = monomial_mul_src(j + n, a_hat, N, neg);
          s0 = acc[row * N + src];
          const uint64_t c1 = (neg ? (uint64_t)0 - s0 : s0) - acc[row * N + j + n];
          const double re = i64_to_f64(decomp_digit(c0, a.base_log, a.level, idx));
          const double im = i64_to_f64(decomp_digit(c1, a.base_log, a.level, idx));
          const double wr = tb.twist[2 * j], wi = tb.twist[2 * j + 1];
          x[j] = cplx{fma(re, wr, -(im * wi)), fma(re, wi, im * wr)};
        }
        __syncthreads();
        ref_transform<N, TPB>(x, y, (const cplx *)tb.w, true, tid);
        const cplx *brow = bsk + ((((size_t)i * a.level + idx) * K1 + row) * K1) * n;
        for (int c = 0; c < K1; ++c)
          for (int q = 0; q < PER; ++q) {
            const int pos = tid + q * TPB;
            const cplx g = brow[(size_t)c * n + pos];
            facc[c][q] = uninit ? ref_mul(g, x[pos]) : ref_mul_add(g, x[pos], facc[c][q]);
          }
        uninit = false;
        __syncthreads();
      }
    for (int c = 0; c < K1; ++c) {
      for (int q = 0; q < PER; ++q) x[tid + q * TPB] = facc[c][q];
      __syncthreads();
      ref_transform<N, TPB>(x, y, (const cplx *)tb.w_inv, false, tid);
      const double normalization = 1.0 / (double)n;
      for (int q = 0; q < PER; ++q) {
        const int j = tid + q * TPB;
        const double w_re = normalization * tb.twist[2 * j], w_im = normalization * tb.twist[2 * j + 1];
        const double mul_re = fma(x[j].re, w_re, x[j].im * w_im);
        const double mul_im = fma(-x[j].re, w_im, x[j].im * w_re);
        const double fr = rint((mul_re - rint(mul_re)) * 18446744073709551616.0);
        const double fi = rint((mul_im - rint(mul_im)) * 18446744073709551616.0);
        acc[c * N + j] += (uint64_t)f64_to_i64_sat(fr);
        acc[c * N + j + n] += (uint64_t)f64_to_i64_sat(fi);
      }
      __syncthreads();
    }
  }
  block_sample_extract<N, K1, TPB>(a, acc, sample, 0, false, tid);
}

template <int N, int K1>
static void launch_ref(hipStream_t st, const PbsArgs &a, const RefTables &tb) {
  const size_t smem = (size_t)K1 * N * 8 + (size_t)N * 16;
  hx_set_dynamic_smem_once<pbs_ref64_kernel<N, K1>>(smem);
  HX_LAUNCH((pbs_ref64_kernel<N, K1>), dim3(a.num_samples), dim3(GenericCfg<N>::TPB), smem, st, a, tb);
}
template <int N>
static void launch_conv_ref(hipStream_t st, const uint64_t *src, void *dst, size_t polys, const RefTables &tb) {
  hx_set_dynamic_smem_once<bsk_to_ref64_kernel<N>>((size_t)N * 16);
  HX_LAUNCH((bsk_to_ref64_kernel<N>), dim3((unsigned)polys), dim3(GenericCfg<N>::TPB), (size_t)N * 16, st, src, (cplx *)dst, tb);
}

void launch_pbs_ref64(hipStream_t st, uint32_t N, uint32_t glwe_dim, const PbsArgs &a, const RefTables &tb) {
  HX_PANIC_IF_FALSE(glwe_dim == 1, "reference-order f64 engine: glwe_dimension 1 only (got %u)", glwe_dim);
  switch (N) {
    case 256: launch_ref<256, 2>(st, a, tb); break;
    case 512: launch_ref<512, 2>(st, a, tb); break;
    case 1024: launch_ref<1024, 2>(st, a, tb); break;
    case 2048: launch_ref<2048, 2>(st, a, tb); break;
    default: HX_PANIC("reference-order f64 engine: unsupported polynomial_size=%u", N);
  }
}
void launch_bsk_to_ref64(hipStream_t st, uint32_t N, const uint64_t *src_dev, void *dst, size_t polys, const RefTables &tb) {
  switch (N) {
    case 256: launch_conv_ref<256>(st, src_dev, dst, polys, tb); break;
    case 512: launch_conv_ref<512>(st, src_dev, dst, polys, tb); break;
    case 1024: launch_conv_ref<1024>(st, src_dev, dst, polys, tb); break;
    case 2048: launch_conv_ref<2048>(st, src_dev, dst, polys, tb); break;
    default: HX_PANIC("reference-order f64 engine: unsupported polynomial_size=%u", N);
  }
}

}  // namespace tfhe_hip
