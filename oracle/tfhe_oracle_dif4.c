/*
 * tfhe_oracle_dif4.c — restatement of the reference's OWN f64 transform and conversions, in the one
 * configuration whose outputs the reference publishes digests of: the golden vectors of apps/test-vectors are
 * generated with the feature `experimental-force_fft_algo_dif4` (apps/test-vectors/Cargo.toml:11), which pins
 * tfhe-fft to its radix-4 decimation-in-frequency Stockham plan (fft/mod.rs:178-192).  TEST INFRASTRUCTURE ONLY.
 *
 * Purpose: pin the SEMANTICS of the f64 engine — twist factors, conversion and rounding rules, multiply-
 * accumulate forms, blind-rotation order — to bytes the reference itself produced
 * (apps/test-vectors/data/{toy_params,valid_params_128}/{glwe_after_*_br,lwe_after_*_pbs}.cbor, whose SHA-256
 * are in apps/test-vectors/checksums.sha256).  The GPU engine keeps its own (fixed, documented) butterfly order
 * and is compared with THIS path by phase, and with the fixed-order restatement of tfhe_oracle.c bit for bit.
 *
 * Restated here, with the operation order of the x86 (AVX2+FMA / AVX-512) paths the vectors were made on —
 * V3 and V4 perform the same IEEE operations per element, so the result does not depend on the vector width:
 *   tfhe-fft/src/fft_simd.rs:239-295   sincospi64 (twiddle generator), :297-321 init_wt
 *   tfhe-fft/src/dif4.rs:111-163       stockham_core_generic (radix-4 pass), :185-236 last butterfly
 *   tfhe-fft/src/dif2.rs:100-140       size-2 tail for odd log2
 *   tfhe-fft/src/x86.rs:47-55,121-129  complex multiply = fmaddsub(aa, xy, bb*yx)
 *   tfhe/.../fft/mod.rs:63-74          twisties (f64::sin_cos of i*pi/(2n))
 *   tfhe/.../fft/mod.rs:201-222        convert_forward_torus (scalar, no FMA): key conversion
 *   tfhe/.../fft/x86.rs:414-500        convert_forward_integer (fmsub / fmadd)
 *   tfhe/.../fft/x86.rs:743-790,893-960 convert_add_backward_torus (fmadd / fnmadd, nearest-even rounds)
 *   tfhe/.../crypto/ggsw.rs:483-697    add_external_product_assign, update_with_fmadd (pulp mul_c64s /
 *                                      mul_add_c64s: fmaddsub(aa, xy, fmaddsub(bb, yx, c)))
 *   tfhe/.../crypto/bootstrap.rs:294-365 blind_rotate_assign
 */
#include "tfhe_oracle.h"
#include "tfhe_oracle_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double re, im; } c64;

/* tfhe-fft/src/fft_simd.rs:239-295 (https://stackoverflow.com/a/42792940) */
static void sincospi64(double a, double *s_out, double *c_out) {
  double az = a * 0.0;
  a = fabs(a) < 9007199254740992.0 ? a : az;
  double r = round(a + a); /* Rust f64::round: half away from zero, as C round() */
  int64_t i = (int64_t)r;
  double t = fma(-0.5, r, a);
  double s = t * t;
  r = -1.0369917389758117e-4;
  r = fma(r, s, 1.9294935641298806e-3);
  r = fma(r, s, -2.5806887942825395e-2);
  r = fma(r, s, 2.3533063028328211e-1);
  r = fma(r, s, -1.3352627688538006e+0);
  r = fma(r, s, 4.0587121264167623e+0);
  r = fma(r, s, -4.9348022005446790e+0);
  double c = fma(r, s, 1.0000000000000000e+0);
  r = 4.6151442520157035e-4;
  r = fma(r, s, -7.3700183130883555e-3);
  r = fma(r, s, 8.2145868949323936e-2);
  r = fma(r, s, -5.9926452893214921e-1);
  r = fma(r, s, 2.5501640398732688e+0);
  r = fma(r, s, -5.1677127800499516e+0);
  s = s * t;
  r *= s;
  s = fma(t, 3.14159265358979323846264338327950288, r);
  if (i & 2) { s = 0.0 - s; c = 0.0 - c; }
  if (i & 1) { double tt = 0.0 - s; s = c; c = tt; }
  if (a == floor(a)) s = az;
  *s_out = s; *c_out = c;
}

typedef struct {
  uint32_t N, n;
  double *tw_re, *tw_im; /* twisties */
  c64 *w, *w_inv;        /* 2n each: [w_init (n) | w (n)] */
} dif4_plan;
static dif4_plan g_plans[8];
static int g_nplans = 0;

static const dif4_plan *get_plan(uint32_t N) {
  const dif4_plan *res = NULL;
#pragma omp critical(orc_dif4_plan)
  {
    for (int i = 0; i < g_nplans; ++i) if (g_plans[i].N == N) res = &g_plans[i];
    if (!res) {
      dif4_plan *p = &g_plans[g_nplans];
      uint32_t n = N / 2;
      p->N = N; p->n = n;
      p->tw_re = (double *)malloc(sizeof(double) * n);
      p->tw_im = (double *)malloc(sizeof(double) * n);
      double unit = 3.14159265358979323846264338327950288 / (2.0 * (double)n); /* mod.rs:68 */
      for (uint32_t i = 0; i < n; ++i) { p->tw_im[i] = sin((double)i * unit); p->tw_re[i] = cos((double)i * unit); }
      p->w = (c64 *)malloc(sizeof(c64) * 2 * n);
      p->w_inv = (c64 *)malloc(sizeof(c64) * 2 * n);
      for (uint32_t i = 0; i < 2 * n; ++i) { p->w[i].re = p->w[i].im = NAN; p->w_inv[i] = p->w[i]; }
      /* init_wt(r = 4, n) */
      uint32_t nr = n / 4;
      double theta = -2.0 / (double)n;
      for (uint32_t q = 0; q < nr; ++q)
        for (uint32_t k = 1; k < 4; ++k) {
          double s, c;
          sincospi64(theta * (double)(k * q), &s, &c);
          c64 z = {c, s}, zc = {c, -s};
          p->w[q + k * nr] = z; p->w[n + 4 * q + k] = z;
          p->w_inv[q + k * nr] = zc; p->w_inv[n + 4 * q + k] = zc;
        }
      ++g_nplans;
      res = p;
    }
  }
  return res;
}

static inline c64 cadd(c64 a, c64 b) { c64 r = {a.re + b.re, a.im + b.im}; return r; }
static inline c64 csub(c64 a, c64 b) { c64 r = {a.re - b.re, a.im - b.im}; return r; }
/* FftSimd::mul(a, b) = fmaddsub(aa, xy, bb*yx): re = a.re*b.re - (a.im*b.im), im = a.re*b.im + (a.im*b.re) */
static inline c64 cmul_fft(c64 a, c64 b) {
  c64 r = {fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)};
  return r;
}
/* mul_j: fwd -> swap_re_im(conj(z)) = (-im, re); inverse -> conj(swap_re_im(z)) = (im, -re)  (sign flips by xor) */
static inline c64 mul_j(int fwd, c64 z) {
  c64 r;
  if (fwd) { r.re = -z.im; r.im = z.re; } else { r.re = z.im; r.im = -z.re; }
  return r;
}

/* ordered radix-4 DIF Stockham transform of n = 2^m complex points, in place in buf (scratch: n points) */
static void dif4_transform(c64 *buf, c64 *scratch, const c64 *tw /* 2n: w_init | w */, uint32_t n, int fwd) {
  const c64 *w = tw + n;
  c64 *x = buf, *y = scratch;
  uint32_t s = 1, m = n; /* m = size of the sub-transforms still to do */
  while (m > 4) {
    const c64 *x0 = x, *x1 = x + n / 4, *x2 = x + n / 2, *x3 = x + 3 * (n / 4);
    for (uint32_t q = 0; q < n / (4 * s); ++q) {
      const c64 w1 = w[4 * q * s + 1], w2 = w[4 * q * s + 2], w3 = w[4 * q * s + 3];
      c64 *y0 = y + (size_t)q * 4 * s, *y1 = y0 + s, *y2 = y1 + s, *y3 = y2 + s;
      for (uint32_t j = 0; j < s; ++j) {
        c64 a = x0[q * s + j], b = x1[q * s + j], c = x2[q * s + j], d = x3[q * s + j];
        c64 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), jbmd = mul_j(fwd, csub(b, d));
        y0[j] = cadd(apc, bpd);
        y1[j] = cmul_fft(w1, csub(amc, jbmd));
        y2[j] = cmul_fft(w2, csub(apc, bpd));
        y3[j] = cmul_fft(w3, cadd(amc, jbmd));
      }
    }
    c64 *t = x; x = y; y = t;
    s *= 4; m /= 4;
  }
  /* tail: data is in x; the result must land in buf */
  if (m == 4) {
    c64 *x0 = x, *x1 = x + n / 4, *x2 = x + n / 2, *x3 = x + 3 * (n / 4);
    c64 *o0 = buf, *o1 = buf + n / 4, *o2 = buf + n / 2, *o3 = buf + 3 * (n / 4);
    for (uint32_t j = 0; j < n / 4; ++j) {
      c64 a = x0[j], b = x1[j], c = x2[j], d = x3[j];
      c64 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), jbmd = mul_j(fwd, csub(b, d));
      o0[j] = cadd(apc, bpd); o1[j] = csub(amc, jbmd); o2[j] = csub(apc, bpd); o3[j] = cadd(amc, jbmd);
    }
  } else { /* m == 2 */
    c64 *x0 = x, *x1 = x + n / 2, *o0 = buf, *o1 = buf + n / 2;
    for (uint32_t j = 0; j < n / 2; ++j) {
      c64 a = x0[j], b = x1[j];
      o0[j] = cadd(a, b); o1[j] = csub(a, b);
    }
  }
}

void orc_dif4_fft(double *buf /* n complex, in place */, uint32_t N, int fwd) {
  const dif4_plan *p = get_plan(N);
  c64 *scratch = (c64 *)malloc(sizeof(c64) * p->n);
  dif4_transform((c64 *)buf, scratch, fwd ? p->w : p->w_inv, p->n, fwd);
  free(scratch);
}

/* fft/mod.rs:201-222 + plan.fwd : standard polynomial (torus) -> Fourier, the key conversion path */
static void forward_as_torus(c64 *out, c64 *scratch, const uint64_t *poly, const dif4_plan *p) {
  const double norm = 5.421010862427522e-20; /* 2^-64 */
  uint32_t n = p->n;
  for (uint32_t i = 0; i < n; ++i) {
    double re = (double)(int64_t)poly[i] * norm, im = (double)(int64_t)poly[i + n] * norm;
    out[i].re = re * p->tw_re[i] - im * p->tw_im[i];   /* num_complex Mul, no contraction */
    out[i].im = re * p->tw_im[i] + im * p->tw_re[i];
  }
  dif4_transform(out, scratch, p->w, n, 1);
}

/* fft/x86.rs:414-500 + plan.fwd : decomposition digits -> Fourier */
static void forward_as_integer(c64 *out, c64 *scratch, const int64_t *digits, const dif4_plan *p) {
  uint32_t n = p->n;
  for (uint32_t i = 0; i < n; ++i) {
    double re = (double)digits[i], im = (double)digits[i + n];
    out[i].re = fma(re, p->tw_re[i], -(im * p->tw_im[i]));
    out[i].im = fma(re, p->tw_im[i], im * p->tw_re[i]);
  }
  dif4_transform(out, scratch, p->w, n, 1);
}

/* plan.inv + fft/x86.rs:743-790,893-960 : Fourier -> torus, added to poly */
static void add_backward_as_torus(uint64_t *poly, c64 *fourier, c64 *scratch, const dif4_plan *p) {
  uint32_t n = p->n;
  dif4_transform(fourier, scratch, p->w_inv, n, 0);
  const double normalization = 1.0 / (double)n, scaling = 18446744073709551616.0;
  for (uint32_t i = 0; i < n; ++i) {
    double w_re = normalization * p->tw_re[i], w_im = normalization * p->tw_im[i];
    double mul_re = fma(fourier[i].re, w_re, fourier[i].im * w_im);
    double mul_im = fma(-fourier[i].re, w_im, fourier[i].im * w_re);
    double fr = mul_re - nearbyint(mul_re), fi = mul_im - nearbyint(mul_im);
    fr = nearbyint(fr * scaling); fi = nearbyint(fi * scaling);
    poly[i] += (uint64_t)orc_f64_to_i64_sat(fr);
    poly[i + n] += (uint64_t)orc_f64_to_i64_sat(fi);
  }
}

/* cc/algorithms/lwe_bootstrap_key_conversion.rs (par_convert_standard_lwe_bootstrap_key_to_fourier) */
void orc_dif4_convert_bsk(double *bsk_f, const uint64_t *bsk_std, uint32_t n, uint32_t k, uint32_t N,
                          uint32_t level) {
  const dif4_plan *p = get_plan(N);
  size_t polys = (size_t)n * level * (k + 1) * (k + 1);
#pragma omp parallel
  {
    c64 *scratch = (c64 *)malloc(sizeof(c64) * p->n);
#pragma omp for schedule(static)
    for (size_t q = 0; q < polys; ++q) forward_as_torus((c64 *)(bsk_f + q * N), scratch, bsk_std + q * N, p);
    free(scratch);
  }
}

/* bootstrap.rs:294-365.  msed: the n+1 modulus-switched values (mask then body).  In pulp's complex
 * multiply(-add) (pulp 0.22.3, Cargo.lock) the KEY element is the operand whose parts are broadcast
 * (`lhs` of update_with_fmadd, ggsw.rs:652-676): with the roles swapped the imaginary part rounds differently
 * and the digests no longer match. */
void orc_dif4_blind_rotate(uint64_t *acc, const uint64_t *lut, const uint64_t *msed, const double *bsk_f,
                           uint32_t n, uint32_t k, uint32_t N, uint32_t base_log, uint32_t level) {
  const dif4_plan *p = get_plan(N);
  size_t gl = (size_t)(k + 1) * N, ggsw_sz = (size_t)level * (k + 1) * gl;
  uint32_t nn = N / 2;
  uint64_t *ct1 = (uint64_t *)malloc(sizeof(uint64_t) * gl * 2), *states = ct1 + gl;
  int64_t *digits = (int64_t *)malloc(sizeof(int64_t) * N);
  c64 *fourier = (c64 *)malloc(sizeof(c64) * nn * (k + 3)), *scratch = fourier + nn, *outb = scratch + nn;
  for (uint32_t q = 0; q <= k; ++q) orc_monomial_div(acc + (size_t)q * N, lut + (size_t)q * N, N, msed[n]);
  for (uint32_t i = 0; i < n; ++i) {
    if (msed[i] == 0) continue;
    for (uint32_t q = 0; q <= k; ++q) orc_monomial_mul_and_sub(ct1 + (size_t)q * N, acc + (size_t)q * N, N, msed[i]);
    /* add_external_product_assign(acc, ggsw_i, ct1) */
    const double *ggsw = bsk_f + (size_t)i * ggsw_sz;
    for (size_t j = 0; j < gl; ++j) states[j] = orc_decomp_init_state(ct1[j], base_log, level);
    int uninit = 1;
    for (uint32_t idx = 0; idx < level; ++idx)
      for (uint32_t row = 0; row <= k; ++row) {
        for (uint32_t j = 0; j < N; ++j)
          digits[j] = (int64_t)orc_decompose_one_level(base_log, &states[(size_t)row * N + j]);
        forward_as_integer(fourier, scratch, digits, p);
        const c64 *grow = (const c64 *)(ggsw + ((size_t)idx * (k + 1) + row) * gl);
        for (uint32_t c = 0; c <= k; ++c) {
          c64 *o = outb + (size_t)c * nn;
          const c64 *g = grow + (size_t)c * nn;
          for (uint32_t j = 0; j < nn; ++j) {
            c64 a = g[j], b = fourier[j];
            if (uninit) { /* mul_c64s: fmaddsub(aa, xy, bb*yx) */
              o[j].re = fma(a.re, b.re, -(a.im * b.im));
              o[j].im = fma(a.re, b.im, a.im * b.re);
            } else {      /* mul_add_c64s: fmaddsub(aa, xy, fmaddsub(bb, yx, c)) */
              o[j].re = fma(a.re, b.re, -fma(a.im, b.im, -o[j].re));
              o[j].im = fma(a.re, b.im, fma(a.im, b.re, o[j].im));
            }
          }
        }
        uninit = 0;
      }
    for (uint32_t c = 0; c <= k; ++c) add_backward_as_torus(acc + (size_t)c * N, outb + (size_t)c * nn, scratch, p);
  }
  free(ct1); free(digits); free(fourier);
}
