// fourier.hip — the transform of the path as launches of its own: the entry points the reference's backend tests drive
// (backends/tfhe-cuda-backend/cuda/include/pbs/programmable_bootstrap.h:8-45; tests_and_benchmarks/tests/test_fft.cpp,
// test_forward_fft16x4x16.cpp, test_fft16x4x16.cpp; tfhe/src/core_crypto/gpu/algorithms/test/fft/mod.rs:268-294 with its
// golden spectrum).  One workgroup per polynomial on the generic LDS transform (pbs_common.h: the SPEC's merged-twist
// decimation tree forward, radix-2 DIT backward), the polynomial "compressed" as the reference passes it:
// complex[i] = (p[i], p[i + N/2]).
//
// Orders.  The tree order of the SPEC's forward transform IS the native order of the reference's classic transform
// (NSMFFT_direct): natural frequency f sits at index bitreverse((n - f) mod n), n = N/2 — the permutation the reference's own
// test applies between its two transforms (test_forward_fft16x4x16.cpp:24-31), and the one that maps this transform onto the
// reference's golden spectrum (natural order: F[f] = sum_k (p[k] + i p[k + n]) e^{i pi k / N} e^{-2 pi i k f / n}).
#include "kernels.h"
#include "pbs_common.h"
#include "tables.h"

namespace tfhe_hip {

enum FourierOp : int {
  FOURIER_FORWARD_TREE = 0,     // cuda_forward_fft_classic_async: spectrum in tree (= NSMFFT_direct native) order
  FOURIER_FORWARD_NATURAL = 1,  // cuda_forward_fft16x4x16_async: spectrum in natural frequency order
  FOURIER_BACKWARD_NATURAL = 2, // cuda_backward_fft16x4x16_async: natural-order spectrum -> time domain, untwisted, NOT scaled by 1/n
  FOURIER_MUL = 3,              // cuda_fourier_polynomial_mul*_async: negacyclic product of two polynomials
};

template <int N>
__global__ void __launch_bounds__(GenericCfg<N>::TPB) fourier_kernel(int op, cplx *in1, const cplx *in2, cplx *out, FftTables tb) {
  constexpr int n = N / 2, TPB = GenericCfg<N>::TPB, LOGN = __builtin_ctz((unsigned)n);
  HX_DYN_SMEM(smem);
  const FBuf fbuf{(cplx *)smem};
  const int tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * n;
  auto tree_index = [&](int f) { return (int)(__brev((unsigned)((n - f) & (n - 1))) >> (32 - LOGN)); };
  if (op == FOURIER_BACKWARD_NATURAL) {
    for (int f = tid; f < n; f += TPB) fbuf[tree_index(f)] = in1[base + f];
  } else {
    for (int j = tid; j < n; j += TPB) fbuf[j] = in1[base + j];
  }
  __syncthreads();
  if (op == FOURIER_FORWARD_TREE || op == FOURIER_FORWARD_NATURAL) {
    lds_fft_forward<N, TPB>(fbuf, tb.fwd, tid);
    if (op == FOURIER_FORWARD_TREE) {
      for (int j = tid; j < n; j += TPB) out[base + j] = fbuf[j];
    } else {
      for (int f = tid; f < n; f += TPB) out[base + f] = fbuf[tree_index(f)];
    }
    return;
  }
  if (op == FOURIER_MUL) {
    // as the reference's batch_polynomial_mul (cuda/src/fft/bnsmfft.cuh:695-768): the first operand's spectrum goes back
    // into input1 ("d_input1 can be modified inside the function"), the second is transformed in LDS and multiplied there
    lds_fft_forward<N, TPB>(fbuf, tb.fwd, tid);
    for (int j = tid; j < n; j += TPB) in1[base + j] = fbuf[j];
    __syncthreads();
    for (int j = tid; j < n; j += TPB) fbuf[j] = in2[base + j];
    __syncthreads();
    lds_fft_forward<N, TPB>(fbuf, tb.fwd, tid);
    for (int j = tid; j < n; j += TPB) fbuf[j] = cmul_first(fbuf[j], in1[base + j]);
    __syncthreads();
  }
  lds_fft_inverse<N, TPB>(fbuf, tb.inv, tid);
  // untwist (the table carries 1/n: undone, exactly, for the pure inverse transform — the reference's leaves the scaling to
  // the bootstrapping key, bnsmfft.cuh:996-1003)
  const double scale = op == FOURIER_BACKWARD_NATURAL ? (double)n : 1.0;
  for (int j = tid; j < n; j += TPB) {
    const cplx y = fbuf[j];
    const double ur = tb.untw[2 * j] * scale, ui = tb.untw[2 * j + 1] * scale;
    out[base + j] = cplx{fma(-y.im, ui, y.re * ur), fma(y.im, ur, y.re * ui)};
  }
}

template <int N>
static void launch_fourier_n(hipStream_t st, int op, cplx *in1, const cplx *in2, cplx *out, uint32_t total, const FftTables &tb) {
  hx_set_dynamic_smem_once<fourier_kernel<N>>(fbuf_bytes(N));
  HX_LAUNCH((fourier_kernel<N>), dim3(total), dim3(GenericCfg<N>::TPB), fbuf_bytes(N), st, op, in1, in2, out, tb);
}

// false: a polynomial size the transform does not exist for (the reference's switch falls through without a launch)
bool launch_fourier(hipStream_t st, uint32_t gpu_index, int op, void *in1, const void *in2, void *out, uint32_t N, uint32_t total) {
  if (!total) return true;
  switch (N) {
    case 256: case 512: case 1024: case 2048: case 4096: case 8192: case 16384: break;
    default: return false;
  }
  const FftTables tb = get_fft_tables(gpu_index, st, N);
  cplx *a = (cplx *)in1, *o = (cplx *)out;
  const cplx *b = (const cplx *)in2;
  switch (N) {
    case 256: launch_fourier_n<256>(st, op, a, b, o, total, tb); break;
    case 512: launch_fourier_n<512>(st, op, a, b, o, total, tb); break;
    case 1024: launch_fourier_n<1024>(st, op, a, b, o, total, tb); break;
    case 2048: launch_fourier_n<2048>(st, op, a, b, o, total, tb); break;
    case 4096: launch_fourier_n<4096>(st, op, a, b, o, total, tb); break;
    case 8192: launch_fourier_n<8192>(st, op, a, b, o, total, tb); break;
    default: launch_fourier_n<16384>(st, op, a, b, o, total, tb); break;
  }
  return true;
}

}  // namespace tfhe_hip
