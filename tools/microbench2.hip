// microbench2.hip — more issue-rate probes (64-bit integer adds, bit ops, integer multiplies, scalar work
// interleaved with vector work).  Same conventions as microbench.hip: WPS waves per SIMD on every CU,
// cycles per wave-instruction at 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench2.hip -o gpurun_out/microbench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int REPS = 4096;

#define R4(X) X X X X
#define BODY8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

#define PROBE64(NAME, INSTR, ...)                                                                       \
  __global__ void NAME(uint64_t *out, uint64_t b, uint64_t c) {                                          \
    uint64_t a[8];                                                                                       \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;                                                  \
    for (int r = 0; r < REPS; ++r)                                                                       \
      asm volatile(R4(BODY8(INSTR))                                                                      \
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                   : "v"(b), "v"(c) : __VA_ARGS__);                                                             \
    uint64_t s = 0;                                                                                      \
    for (int i = 0; i < 8; ++i) s += a[i];                                                               \
    if (s == 12345678u) out[0] = s;                                                                      \
  }
#define PROBE32(NAME, INSTR, ...)                                                                       \
  __global__ void NAME(uint64_t *out, uint32_t b, uint32_t c) {                                          \
    uint32_t a[8];                                                                                       \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;                                                  \
    for (int r = 0; r < REPS; ++r)                                                                       \
      asm volatile(R4(BODY8(INSTR))                                                                      \
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                   : "v"(b), "v"(c) : __VA_ARGS__);                                                             \
    uint32_t s = 0;                                                                                      \
    for (int i = 0; i < 8; ++i) s += a[i];                                                               \
    if (s == 12345678u) out[0] = s;                                                                      \
  }

#define I_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %8, 0, %" #i "\n"
#define I_LSHL64(i) "v_lshlrev_b64 %" #i ", 3, %" #i "\n"
PROBE64(k_lshladd64, I_LSHLADD64, "memory")
PROBE64(k_lshl64, I_LSHL64, "memory")
__global__ void k_madu64(uint64_t *out, uint32_t b, uint32_t c) {
  uint64_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
#define I_MADU64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"
  for (int r = 0; r < REPS; ++r)
    asm volatile(R4(BODY8(I_MADU64))
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                 : "v"(b), "v"(c) : "vcc");
  uint64_t s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345678u) out[0] = s;
}

#define I_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define I_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define I_MIN(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define I_MIN3(i) "v_min3_i32 %" #i ", %" #i ", %8, %9\n"
#define I_ASHR(i) "v_ashrrev_i32 %" #i ", 9, %" #i "\n"
#define I_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define I_XOR(i) "v_xor_b32 %" #i ", %8, %" #i "\n"
#define I_XOR_S(i) "v_xor_b32 %" #i ", s20, %" #i "\n"
#define I_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define BODY4(I) I(0, 4) I(1, 5) I(2, 6) I(3, 7)
#define PROBEP(NAME, INSTR, ...)                                                                         \
  __global__ void NAME(uint64_t *out, uint32_t b, uint32_t c) {                                          \
    uint32_t a[8];                                                                                       \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;                                                  \
    for (int r = 0; r < REPS; ++r)                                                                       \
      asm volatile(R4(BODY4(INSTR) BODY4(INSTR))                                                         \
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                   : "v"(b), "v"(c) : __VA_ARGS__);                                                      \
    uint32_t s = 0;                                                                                      \
    for (int i = 0; i < 8; ++i) s += a[i];                                                               \
    if (s == 12345678u) out[0] = s;                                                                      \
  }
#define I_SUBPAIR(l, h) "v_sub_co_u32 %" #l ", vcc, %" #l ", %8\n s_nop 0\n v_subb_co_u32 %" #h ", vcc, %" #h ", %9, vcc\n"
#define I_SUBPAIR_NN(l, h) "v_sub_co_u32 %" #l ", vcc, %" #l ", %8\n v_subb_co_u32 %" #h ", vcc, %" #h ", %9, vcc\n"
#define I_ADDPAIR_S(l, h) "v_add_co_u32_e64 %" #l ", s[20:21], %" #l ", %8\n s_nop 1\n v_addc_co_u32_e64 %" #h ", s[22:23], %" #h ", %9, s[20:21]\n"
PROBEP(k_subpair, I_SUBPAIR, "vcc")
PROBEP(k_addpair_s, I_ADDPAIR_S, "s20", "s21", "s22", "s23")
PROBE32(k_bitop3, I_BITOP3, "memory")
PROBE32(k_mullo, I_MULLO, "memory")
PROBE32(k_mulhi, I_MULHI, "memory")
PROBE32(k_min, I_MIN, "memory")
PROBE32(k_min3, I_MIN3, "memory")
PROBE32(k_ashr, I_ASHR, "memory")
PROBE32(k_cnds, I_CNDS, "s20", "s21")
PROBE32(k_add3, I_ADD3, "memory")
PROBE32(k_xor, I_XOR, "memory")
PROBE32(k_xor_s, I_XOR_S, "s20")
PROBE32(k_mul24, I_MUL24, "memory")
PROBE32(k_mad24, I_MAD24, "memory")

// one f64 FMA followed by K scalar ALU instructions: are the scalar ones hidden behind the vector pipe?
#define FMA_S0(i) "v_fma_f64 %" #i ", %8, %9, %" #i "\n"
#define FMA_S1(i) "v_fma_f64 %" #i ", %8, %9, %" #i "\n s_add_u32 s20, s20, 1\n"
#define FMA_S2(i) "v_fma_f64 %" #i ", %8, %9, %" #i "\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s20, 0x7c0\n"
#define FMA_S4(i) "v_fma_f64 %" #i ", %8, %9, %" #i "\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s20, 0x7c0\n s_sub_u32 s22, s21, s20\n s_xor_b32 s23, s22, s21\n"
#define FMA_N1(i) "v_fma_f64 %" #i ", %8, %9, %" #i "\n s_nop 0\n"

#define PROBEF(NAME, INSTR)                                                                              \
  __global__ void NAME(uint64_t *out, double b, double c) {                                              \
    double a[8];                                                                                         \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;                                                  \
    for (int r = 0; r < REPS; ++r)                                                                       \
      asm volatile(R4(BODY8(INSTR))                                                                      \
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(c) \
                   : "v"(b) : "s20", "s21", "s22", "s23", "scc", "v200");                                        \
    double s = c;                                                                                        \
    for (int i = 0; i < 8; ++i) s += a[i];                                                               \
    if (s == 12345.678) out[0] = (uint64_t)s;                                                            \
  }
#undef FMA_S0
#define FMA_S0(i) "v_fma_f64 %" #i ", %9, %8, %" #i "\n"
#undef FMA_S1
#define FMA_S1(i) "v_fma_f64 %" #i ", %9, %8, %" #i "\n s_add_u32 s20, s20, 1\n"
#undef FMA_S2
#define FMA_S2(i) "v_fma_f64 %" #i ", %9, %8, %" #i "\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s20, 0x7c0\n"
#undef FMA_S4
#define FMA_S4(i) "v_fma_f64 %" #i ", %9, %8, %" #i "\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s20, 0x7c0\n s_sub_u32 s22, s21, s20\n s_xor_b32 s23, s22, s21\n"
#undef FMA_N1
#define FMA_N1(i) "v_fma_f64 %" #i ", %9, %8, %" #i "\n s_nop 0\n"
#undef FMA_V1
#define FMA_V1(i) "v_fma_f64 %" #i ", %9, %8, %" #i "\n v_add_u32 v200, v200, v200\n"
PROBEF(k_fma_s0, FMA_S0)
PROBEF(k_fma_s1, FMA_S1)
PROBEF(k_fma_s2, FMA_S2)
PROBEF(k_fma_s4, FMA_S4)
PROBEF(k_fma_n1, FMA_N1)
PROBEF(k_fma_v1, FMA_V1)

template <class F>
static float time_ms(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const double GHZ = 2.4;
  uint64_t *dout;
  CK(hipMalloc(&dout, 1024));
  printf("probe                       waves/SIMD    ms     cycles per probe group (at %.1f GHz)\n", GHZ);
  for (int wps = 1; wps <= 2; ++wps) {
    const dim3 grid(256 * wps), block(256);
    const double n = (double)REPS * 32;
    auto rep = [&](const char *name, float ms) { printf("%-28s %d          %7.3f  %.2f\n", name, wps, ms, ms * 1e-3 * GHZ * 1e9 / (n * wps)); };
#define RUN64(K) rep(#K, time_ms([&] { hipLaunchKernelGGL(K, grid, block, 0, 0, dout, (uint64_t)3, (uint64_t)5); }))
#define RUN32(K) rep(#K, time_ms([&] { hipLaunchKernelGGL(K, grid, block, 0, 0, dout, 3u, 5u); }))
#define RUNF(K) rep(#K, time_ms([&] { hipLaunchKernelGGL(K, grid, block, 0, 0, dout, 1.0000001, 1e-9); }))
    RUN64(k_lshladd64); RUN64(k_lshl64); RUN32(k_madu64);
    printf("(next two: 8 pairs per group of 32 slots -> divide by the pair count yourself: value*32/8... reported per 1/32 of a trip with 8 pairs + nops)\n");
    RUN32(k_subpair); RUN32(k_addpair_s);
    RUN32(k_bitop3); RUN32(k_mullo); RUN32(k_mulhi); RUN32(k_min); RUN32(k_min3); RUN32(k_ashr); RUN32(k_cnds);
    RUN32(k_add3); RUN32(k_xor); RUN32(k_xor_s); RUN32(k_mul24); RUN32(k_mad24);
    RUNF(k_fma_s0); RUNF(k_fma_s1); RUNF(k_fma_s2); RUNF(k_fma_s4); RUNF(k_fma_n1); RUNF(k_fma_v1);
  }
  return 0;
}
