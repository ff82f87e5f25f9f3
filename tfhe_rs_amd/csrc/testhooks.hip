// testhooks.hip — exposes single device functions and transforms so the parity tests can
// compare them with the oracle value by value (hip_test_* in include/tfhe_hip_backend.h).
#include "kernels.h"

namespace tfhe_hip {

// op: 0 modulus_switch(x, p0)            1 decomp_init_state(x, base_log=p0, level=p1)
//     2 decomp_digit(x, p0, p1, idx) for idx = 0..p1-1 (out has count*p1 entries)
//     3 from_torus(bits-as-f64)          4 f64_to_i64_sat(bits-as-f64)
//     5 i64_to_f64(x) (out = f64 bits)   6 gl_modswitch_from_pow2   7 gl_modswitch_to_pow2
//     8 gl_mul(in[2i], in[2i+1])         9 gl_add   10 gl_sub
//     11 decomp_digit_l1_hi(hi32(x), base_log=p0)  (single-level fast path of the wave kernel)
__global__ void test_arith_kernel(uint32_t op, const uint64_t *in, uint64_t *out, uint32_t count, uint32_t p0,
                                  uint32_t p1) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  union { uint64_t u; double d; } cv;
  switch (op) {
    case 0: out[i] = modulus_switch(in[i], p0); break;
    case 1: out[i] = decomp_init_state(in[i], p0, p1); break;
    case 2: for (uint32_t idx = 0; idx < p1; ++idx) out[(size_t)i * p1 + idx] = (uint64_t)decomp_digit(in[i], p0, p1, idx); break;
    case 3: cv.u = in[i]; out[i] = from_torus(cv.d); break;
    case 4: cv.u = in[i]; out[i] = (uint64_t)f64_to_i64_sat(cv.d); break;
    case 5: cv.d = i64_to_f64((int64_t)in[i]); out[i] = cv.u; break;
    case 6: out[i] = gl_modswitch_from_pow2(in[i]); break;
    case 7: out[i] = gl_modswitch_to_pow2(in[i]); break;
    case 8: out[i] = gl_mul(in[2 * i], in[2 * i + 1]); break;
    case 9: out[i] = gl_add(in[2 * i], in[2 * i + 1]); break;
    case 10: out[i] = gl_sub(in[2 * i], in[2 * i + 1]); break;
    case 11: out[i] = (uint64_t)(int64_t)decomp_digit_l1_hi((uint32_t)(in[i] >> 32), p0); break;
    case 12: out[i] = gl_modswitch_to_pow2_lazy(in[i]); break;                 // any in[i] (lazy value)
    case 13: out[i] = gl_horner16(in[2 * i], in[2 * i + 1]); break;            // lazy result: compare mod p
    case 14: out[i] = gl_acc_modswitch_to_pow2_lazy(in[2 * i], in[2 * i + 1]); break;  // acc + modswitch(lazy v)
    default: out[i] = 0;
  }
}

void launch_test_arith(hipStream_t st, uint32_t op, const uint64_t *in, uint64_t *out, uint32_t count, uint32_t p0,
                       uint32_t p1) {
  if (!count) return;
  HX_LAUNCH(test_arith_kernel, dim3((count + 255) / 256), dim3(256), 0, st, op, in, out, count, p0, p1);
}

// op: 0 forward(int digits i64[N]) -> f64[N]      1 forward(torus u64[N]) -> f64[N] (BSK conversion)
//     2 backward-add: in = f64[N] fourier followed by u64[N] poly ; out = u64[N]
//     3 ntt forward u64[N] (values < p)            4 ntt normalize+inverse u64[N]
template <int N>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) test_transform_kernel(uint32_t op, const void *in, void *out,
                                                                           FftTables ft, NttTables nt) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB;
  HX_DYN_SMEM(smem);
  const int tid = threadIdx.x;
  if (op <= 2) {
    const FBuf fbuf{(cplx *)smem};
    if (op == 0) {
      const int64_t *d = (const int64_t *)in;
      for (int j = tid; j < n; j += TPB) fbuf[j] = cplx{i64_to_f64(d[j]), i64_to_f64(d[j + n])};
    } else if (op == 1) {
      const uint64_t *p = (const uint64_t *)in;
      for (int j = tid; j < n; j += TPB)
        fbuf[j] = cplx{i64_to_f64((int64_t)p[j]) * 5.421010862427522e-20, i64_to_f64((int64_t)p[j + n]) * 5.421010862427522e-20};
    } else {
      const cplx *f = (const cplx *)in;
      for (int j = tid; j < n; j += TPB) fbuf[j] = f[j];
    }
    __syncthreads();
    if (op <= 1) {
      lds_fft_forward<N, TPB>(fbuf, ft.fwd, tid);
      cplx *o = (cplx *)out;
      for (int j = tid; j < n; j += TPB) o[j] = fbuf[j];
    } else {
      lds_fft_inverse<N, TPB>(fbuf, ft.inv, tid);
      const uint64_t *poly = (const uint64_t *)((const cplx *)in + n);
      uint64_t *o = (uint64_t *)out;
      for (int j = tid; j < n; j += TPB) {
        const cplx y = fbuf[j];
        const double ur = ft.untw[2 * j], ui = ft.untw[2 * j + 1];
        o[j] = poly[j] + from_torus(fma(-y.im, ui, y.re * ur));
        o[j + n] = poly[j + n] + from_torus(fma(y.im, ur, y.re * ui));
      }
    }
  } else {
    uint64_t *nbuf = (uint64_t *)smem;
    const uint64_t *p = (const uint64_t *)in;
    for (int j = tid; j < N; j += TPB) nbuf[j] = (op == 4) ? gl_mul(p[j], nt.n_inv) : p[j];
    __syncthreads();
    if (op == 3) lds_ntt_forward<N, TPB>(nbuf, nt.tw, tid);
    else lds_ntt_inverse<N, TPB>(nbuf, nt.itw, tid);
    uint64_t *o = (uint64_t *)out;
    for (int j = tid; j < N; j += TPB) o[j] = nbuf[j];
  }
}

template <int N>
static void launch_tt(hipStream_t st, uint32_t op, const void *in, void *out, const FftTables &ft, const NttTables &nt) {
  HX_LAUNCH((test_transform_kernel<N>), dim3(1), dim3(GenericCfg<N>::TPB), fbuf_bytes(N), st, op, in, out, ft, nt);
}

void launch_test_transform(hipStream_t st, uint32_t op, uint32_t N, const void *in, void *out, uint32_t gpu_index) {
  const FftTables ft = get_fft_tables(gpu_index, st, N);
  const NttTables nt = get_ntt_tables(gpu_index, st, N);
  switch (N) {
    case 256: launch_tt<256>(st, op, in, out, ft, nt); break;
    case 512: launch_tt<512>(st, op, in, out, ft, nt); break;
    case 1024: launch_tt<1024>(st, op, in, out, ft, nt); break;
    case 2048: launch_tt<2048>(st, op, in, out, ft, nt); break;
    case 4096: launch_tt<4096>(st, op, in, out, ft, nt); break;
    default: HX_PANIC("unsupported polynomial_size=%u", N);
  }
}

}  // namespace tfhe_hip
