// microbench4.hip — round 5: issue cost of the 64-bit integer forms the split-key engine's Horner step is made of, and
// whether the integer instructions of one wave issue under the f64 instructions of the other wave of its SIMD.
// Two waves per SIMD on every CU (512 threads per block, 256 blocks); cycles per wave-instruction at the clock the run held
// (s_memtime around the loop).  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench4.hip -o tools/bin/microbench4
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int REPS = 2048;
#define R4(X) X X X X
#define B8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define I_FMA(i) "v_fma_f64 %" #i ", %16, %17, %" #i "\n"
#define I_MAD(i) "v_mad_u64_u32 %" #i ", s[20:21], %18, -1, %" #i "\n"
#define I_LSHLADD(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %16\n"
#define I_LSHL64(i) "v_lshlrev_b64 %" #i ", 16, %" #i "\n"
#define I_CMP64(i) "v_cmp_lt_u64 vcc, %" #i ", %16\n"
#define I_XOR(i) "v_xor_b32 %[q" #i "], %18, %[q" #i "]\n"
#define I_CND64(i) "v_cndmask_b32_e64 %[q" #i "], 0, 1, s[22:23]\n"

// mode_a / mode_b: instruction kind of the lower / upper four waves of the block (0 f64 fma, 1 mad_u64_u32, 2 lshl_add_u64,
// 3 lshlrev_b64, 4 cmp_lt_u64, 5 xor (VOP2), 6 cndmask_e64, 7 nothing)
__global__ void __launch_bounds__(512) k(uint64_t *out, int mode_a, int mode_b, double fb, double fc, uint32_t ib) {
  const int mode = (threadIdx.x >> 8) ? mode_b : mode_a;
  double a[8];
  uint32_t q[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i, q[i] = threadIdx.x * 3 + i;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
#define LOOP(BODY)                                                                                                       \
  for (int r = 0; r < REPS; ++r)                                                                                         \
    asm volatile(R4(B8(BODY)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                   , [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [q4] "+v"(q[4]), [q5] "+v"(q[5]), [q6] "+v"(q[6]), [q7] "+v"(q[7]) \
                 : "v"(fb), "v"(fc), "v"(ib) : "vcc", "s20", "s21", "s22", "s23");
  switch (mode) {
    case 0: LOOP(I_FMA) break;
    case 1: LOOP(I_MAD) break;
    case 2: LOOP(I_LSHLADD) break;
    case 3: LOOP(I_LSHL64) break;
    case 4: LOOP(I_CMP64) break;
    case 5: LOOP(I_XOR) break;
    case 6: LOOP(I_CND64) break;
    default: break;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + q[i];
  if (s == 12345.678) out[0] = (uint64_t)s;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  uint64_t *d;
  (void)hipMalloc(&d, (1 + 256 * 8) * 8);
  static uint64_t h[1 + 256 * 8];
  const char *names[] = {"v_fma_f64", "v_mad_u64_u32", "v_lshl_add_u64", "v_lshlrev_b64", "v_cmp_lt_u64", "v_xor_b32 (VOP2)",
                         "v_cndmask_b32_e64", "(idle)"};
  printf("lower four waves        upper four waves        cycles per instruction: lower  upper   (2 waves per SIMD; 32 x %d instructions each)\n", REPS);
  const int pairs[][2] = {{0, 7}, {0, 0}, {1, 7}, {1, 1}, {2, 2}, {3, 3}, {4, 4}, {5, 5}, {6, 6}, {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 5}, {2, 5}};
  for (auto &p : pairs) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, p[0], p[1], 1.0000001, 1e-9, 3u);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double lo = 0, up = 0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? lo : up) += (double)h[1 + b * 8 + w];
    const double n = (double)REPS * 32;
    printf("%-23s %-23s %6.2f %6.2f\n", names[p[0]], names[p[1]], p[0] == 7 ? 0.0 : lo / (256 * 4) / n, p[1] == 7 ? 0.0 : up / (256 * 4) / n);
  }
  return 0;
}
