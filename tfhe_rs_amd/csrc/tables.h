// tables.h — per-device constant tables (f64 transform twiddles, Goldilocks NTT twiddles).
#pragma once
#include "hx.h"

namespace tfhe_hip {

// f64 transform tables for polynomial size N (n = N/2 complex points), DESIGN.md §4:
//   fwd[(1<<d)+g]  = exp(i*pi*(1+4*bitrev_d(g)) / 2^(d+2))       d = 0..log2(n)-1
//   inv[half+j]    = exp(-2*pi*i*j / (2*half))                   half = 1,2,..,n/2 ; j < half
//   untw[j]        = conj(exp(i*pi*j/N)) / n
// each entry (re, im) as two doubles; index 0 of fwd/inv unused.
//   mono[j]        = exp(i*pi*j/N), j < 2N (octant-symmetric): the transform of a monomial X^d at position p is
//                    mono[((1+4*bitrev_{L-4}(p>>4))*d) mod 2N] * mono[(N/8)*((bitrev_4(p&15)*d) mod 16)]  (multi-bit PBS)
//   mono_lane      (N = 2048 and requested, else null): the same base factors laid out for a wave — entry [d][h] =
//                    mono[((1 + 4 bitrev_6(h)) d) mod 2N], d < 2N, h < 64: the 64 lanes of a wave read ONE 1 KB run for a
//                    degree d instead of 64 scattered 16-byte entries (4 MB; same values)
struct FftTables {
  const double *fwd;
  const double *inv;
  const double *untw;
  const double *mono;
  const double *mono_lane;
};

// Tables of the reference-order f64 engine (pbs_ref64.hip): what tfhe-fft / tfhe build for a radix-4 DIF plan of
// n = N/2 points — twist[i] = (cos, sin)(i*pi/(2n)) from the host libm like Twisties::new (fft/mod.rs:63-74);
// w = [w_init | w] of init_wt(4, n) with tfhe-fft's sincospi64 (fft_simd.rs:239-321), w_inv its conjugate
struct RefTables {
  const double *twist;
  const double *w;
  const double *w_inv;
};

// Goldilocks tables: tw[m+g] = psi^bitrev(m+g), itw likewise for psi^-1; n_inv = N^-1 mod p
struct NttTables {
  const uint64_t *tw;
  const uint64_t *itw;
  uint64_t n_inv;
};


// Lazily built, cached per (device, N); `stream` orders the upload before first use.
// with_mono_lane: the caller runs multi-bit kernels (mono_lane is built on first such request, 4 MB at N = 2048)
FftTables get_fft_tables(uint32_t gpu_index, hipStream_t stream, uint32_t N, bool with_mono_lane = false);
NttTables get_ntt_tables(uint32_t gpu_index, hipStream_t stream, uint32_t N);
RefTables get_ref_tables(uint32_t gpu_index, hipStream_t stream, uint32_t N);

// host-side generators (also exported through the test ABI so the tables can be compared
// with the oracle's independently computed ones)
void fill_fft_tables_host(uint32_t N, double *fwd, double *inv, double *untw);
void fill_monomial_table_host(uint32_t N, double *mono /* 4N doubles */);
void fill_ntt_tables_host(uint32_t N, uint64_t *tw, uint64_t *itw, uint64_t *n_inv);

}  // namespace tfhe_hip
