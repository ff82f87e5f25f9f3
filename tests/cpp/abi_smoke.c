/* abi_smoke.c — drives the backend from COMPILED code through nothing but the C ABI of
 * include/tfhe_hip_backend.h (plain pointers and sizes), the way the reference's Rust FFI does
 * (tfhe/src/core_crypto/gpu/ffi.rs:21-92: scratch -> launch -> cleanup on one stream).
 * Test infrastructure: tests/test_c_host.py writes the fixture (keys, inputs, the oracle's expected bits),
 * builds this file with gcc and runs it against the library given on the command line.
 *
 * fixture (little-endian u64 words): n k N base_log level ms_type B | bsk[n*(k+1)^2*level*N] |
 *                                    lut[(k+1)*N] | lwe_in[B*(n+1)] | expected[B*(k*N+1)]               */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/tfhe_hip_backend.h"

#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); \
  if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s <library.so> <fixture.bin>\n", argv[0]); return 2; }
  void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  SYM(cuda_create_stream_ffi) SYM(cuda_destroy_stream) SYM(cuda_synchronize_stream) SYM(cuda_malloc) SYM(cuda_drop)
  SYM(cuda_memcpy_async_to_gpu) SYM(cuda_memcpy_async_to_cpu) SYM(cuda_convert_lwe_programmable_bootstrap_key_64_async)
  SYM(scratch_cuda_programmable_bootstrap_64_async) SYM(cuda_programmable_bootstrap_64_async)
  SYM(cleanup_cuda_programmable_bootstrap_64)

  FILE *f = fopen(argv[2], "rb");
  if (!f) { perror("fixture"); return 2; }
  uint64_t hdr[7];
  if (fread(hdr, 8, 7, f) != 7) return 2;
  const uint32_t n = hdr[0], k = hdr[1], N = hdr[2], base_log = hdr[3], level = hdr[4], ms_type = hdr[5], B = hdr[6];
  const size_t bsk_w = (size_t)n * (k + 1) * (k + 1) * level * N, lut_w = (size_t)(k + 1) * N;
  const size_t in_w = (size_t)B * (n + 1), out_w = (size_t)B * ((size_t)k * N + 1);
  uint64_t *bsk = malloc(bsk_w * 8), *lut = malloc(lut_w * 8), *in = malloc(in_w * 8), *want = malloc(out_w * 8),
           *got = malloc(out_w * 8), *idx = malloc((size_t)B * 8), *zero = calloc(B, 8);
  if (fread(bsk, 8, bsk_w, f) != bsk_w || fread(lut, 8, lut_w, f) != lut_w || fread(in, 8, in_w, f) != in_w ||
      fread(want, 8, out_w, f) != out_w) { fprintf(stderr, "short fixture\n"); return 2; }
  fclose(f);
  for (uint32_t i = 0; i < B; ++i) idx[i] = i;

  void *s = p_cuda_create_stream_ffi(0);
  void *d_bsk = p_cuda_malloc(bsk_w * 8, 0), *d_lut = p_cuda_malloc(lut_w * 8, 0), *d_in = p_cuda_malloc(in_w * 8, 0);
  void *d_out = p_cuda_malloc(out_w * 8, 0), *d_idx = p_cuda_malloc((size_t)B * 8, 0), *d_zero = p_cuda_malloc((size_t)B * 8, 0);
  p_cuda_convert_lwe_programmable_bootstrap_key_64_async(s, 0, d_bsk, bsk, n, k, level, N);
  p_cuda_memcpy_async_to_gpu(d_lut, lut, lut_w * 8, s, 0);
  p_cuda_memcpy_async_to_gpu(d_in, in, in_w * 8, s, 0);
  p_cuda_memcpy_async_to_gpu(d_idx, idx, (size_t)B * 8, s, 0);
  p_cuda_memcpy_async_to_gpu(d_zero, zero, (size_t)B * 8, s, 0);
  int8_t *buf = NULL;
  p_scratch_cuda_programmable_bootstrap_64_async(s, 0, &buf, n, k, N, level, B, true, (enum PBS_MS_REDUCTION_T)ms_type);
  p_cuda_programmable_bootstrap_64_async(s, 0, d_out, d_idx, d_lut, d_zero, d_in, d_idx, d_bsk, buf, n, k, N, base_log,
                                         level, B, 1, 0);
  p_cuda_memcpy_async_to_cpu(got, d_out, out_w * 8, s, 0);
  p_cuda_synchronize_stream(s, 0);
  p_cleanup_cuda_programmable_bootstrap_64(s, 0, &buf);
  p_cuda_drop(d_bsk, 0); p_cuda_drop(d_lut, 0); p_cuda_drop(d_in, 0); p_cuda_drop(d_out, 0); p_cuda_drop(d_idx, 0);
  p_cuda_drop(d_zero, 0);
  p_cuda_destroy_stream(s, 0);
  size_t bad = 0;
  for (size_t i = 0; i < out_w; ++i) bad += got[i] != want[i];
  printf("%s: %u PBS, %zu of %zu output words differ from the expected bits\n", bad ? "FAIL" : "OK", B, bad, out_w);
  return bad ? 1 : 0;
}
