// microbench3.hip — does the f64 matrix pipe run beside the f64 vector pipe on gfx950?
// v_mfma_f64_16x16x4_f64 (2048 flop) and v_fma_f64 (128 flop per wave) have the same peak rate on this part; the
// question for the transform kernels is whether a SIMD can keep both busy at once (a wave's matrix instruction under
// its own or its neighbour's vector instructions).  Probes, each on every CU with WPS waves per SIMD:
//   fma      32 independent v_fma_f64 per trip
//   mfma     8 v_mfma_f64_16x16x4_f64 on 4 independent accumulators per trip
//   mix      the two bodies interleaved in ONE wave (1 matrix instruction per 4 vector ones)
//   split    waves 0..3 of a workgroup (one per SIMD) run the mfma body, waves 4..7 the fma body (WPS = 2 only)
// Reported: ms and cycles per trip at 2.4 GHz; if the pipes overlap, mix ~ max(fma, mfma) and split ~ max, else the sum.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench3.hip -o gpurun_out/microbench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int REPS = 8192;
typedef double d4 __attribute__((ext_vector_type(4)));

#define FMA8 "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n" \
             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
#define MFMA4 "v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n" \
              "v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n"

__device__ __forceinline__ void fma_trip(double (&a)[8], double b, double c) {
  asm volatile(FMA8 FMA8 FMA8 FMA8
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
               : "v"(b), "v"(c));
}
__device__ __forceinline__ void mfma_trip(d4 (&m)[4], double b, double c) {
  asm volatile(MFMA4 MFMA4 : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]) : "v"(b), "v"(c));
}
// one matrix instruction, then four vector ones, eight times: 8 mfma + 32 fma per trip
#define MIX1(M) "v_mfma_f64_16x16x4_f64 %" #M ", %12, %13, %" #M "\n"
#define MIXF(A, B, C, D) "v_fma_f64 %" #A ", %" #A ", %12, %13\n v_fma_f64 %" #B ", %" #B ", %12, %13\n" \
                         "v_fma_f64 %" #C ", %" #C ", %12, %13\n v_fma_f64 %" #D ", %" #D ", %12, %13\n"
__device__ __forceinline__ void mix_trip(d4 (&m)[4], double (&a)[8], double b, double c) {
  asm volatile(MIX1(0) MIXF(4, 5, 6, 7) MIX1(1) MIXF(8, 9, 10, 11) MIX1(2) MIXF(4, 5, 6, 7) MIX1(3) MIXF(8, 9, 10, 11)
               MIX1(0) MIXF(4, 5, 6, 7) MIX1(1) MIXF(8, 9, 10, 11) MIX1(2) MIXF(4, 5, 6, 7) MIX1(3) MIXF(8, 9, 10, 11)
               : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]),
                 "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
               : "v"(b), "v"(c));
}

template <int MODE>
__global__ void probe(double *out, double b, double c) {
  double a[8];
  d4 m[4];
  for (int i = 0; i < 8; ++i) a[i] = (double)threadIdx.x + i;
  for (int i = 0; i < 4; ++i) m[i] = d4{(double)threadIdx.x, 1.0, 2.0, 3.0 + i};
  const bool second = (threadIdx.x >> 6) >= 4;
  for (int r = 0; r < REPS; ++r) {
    if (MODE == 0) fma_trip(a, b, c);
    else if (MODE == 1) mfma_trip(m, b, c);
    else if (MODE == 2) mix_trip(m, a, b, c);
    else if (second) fma_trip(a, b, c);
    else mfma_trip(m, b, c);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += m[i].x + m[i].y + m[i].z + m[i].w;
  if (s == 12345.678) out[0] = s;
}

// rounding of the matrix instruction: D = C + sum_k A[i][k] B[k][j] — a chain of fused multiply-adds in k order, or
// something else?  A row of (1, 2^-53, 2^-53, 2^-53) against B = 1 and C = 1 - 2^-53 ... tells a k-ordered fma chain
// (each tiny addend rounds away or not) from an exact sum rounded once.
__global__ void rounding(double *out) {
  const int lane = threadIdx.x;
  const int k = lane >> 4, i = lane & 15;
  const double tiny = 0x1p-53;
  // A[i][k]: row 0 = (1, tiny, tiny, tiny); row 1 = (tiny, tiny, tiny, 1); row 2 = (tiny, 1, tiny, tiny); others 0
  double av = 0.0;
  if (i == 0) av = k == 0 ? 1.0 : tiny;
  if (i == 1) av = k == 3 ? 1.0 : tiny;
  if (i == 2) av = k == 1 ? 1.0 : tiny;
  if (i == 3) av = tiny;                 // (tiny, tiny, tiny, tiny) onto C = 1
  const double bv = 1.0;                 // B[k][j] = 1
  d4 cacc = d4{0.0, 0.0, 0.0, 0.0};
  if (i == 0 && false) cacc.x = 0.0;
  d4 c1 = cacc;
  // C rows: lane holds rows (lane >> 4) + 4 reg; row 3 (lane group 3, reg 0) starts at 1.0
  if ((lane >> 4) == 3) c1.x = 1.0;
  d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
  // rows 0..3 of column 0: lanes 0, 16, 32, 48 reg 0
  if ((lane & 15) == 0) out[lane >> 4] = d.x;
}

// D = A B + C on random operands, for the host to compare with candidate summation orders
__global__ void mfma_once(const double *A, const double *B, const double *Cm, double *D) {
  const int lane = threadIdx.x;
  const double av = A[(lane & 15) * 4 + (lane >> 4)];   // A[i][k], i = lane & 15, k = lane >> 4
  const double bv = B[(lane >> 4) * 16 + (lane & 15)];  // B[k][j], k = lane >> 4, j = lane & 15
  d4 c;
  c.x = Cm[((lane >> 4) + 0) * 16 + (lane & 15)];
  c.y = Cm[((lane >> 4) + 4) * 16 + (lane & 15)];
  c.z = Cm[((lane >> 4) + 8) * 16 + (lane & 15)];
  c.w = Cm[((lane >> 4) + 12) * 16 + (lane & 15)];
  const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
  D[((lane >> 4) + 0) * 16 + (lane & 15)] = d.x;
  D[((lane >> 4) + 4) * 16 + (lane & 15)] = d.y;
  D[((lane >> 4) + 8) * 16 + (lane & 15)] = d.z;
  D[((lane >> 4) + 12) * 16 + (lane & 15)] = d.w;
}

static int summation_order_probe() {
  double hA[64], hB[64], hC[256], hD[256];
  double *dA, *dB, *dC, *dD;
  CK(hipMalloc(&dA, sizeof(hA)));
  CK(hipMalloc(&dB, sizeof(hB)));
  CK(hipMalloc(&dC, sizeof(hC)));
  CK(hipMalloc(&dD, sizeof(hD)));
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {  // xorshift; values of mixed magnitude so that the order of the roundings shows
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    const double m = (double)(int64_t)(st >> 11) / 9007199254740992.0 - 0.5;
    return m * (double)(1ull << ((st >> 3) & 31));
  };
  int bad[5] = {0, 0, 0, 0, 0}, total = 0;
  for (int trial = 0; trial < 64; ++trial) {
    for (double &x : hA) x = rnd();
    for (double &x : hB) x = rnd();
    for (double &x : hC) x = rnd();
    CK(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    CK(hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        const double *a = &hA[i * 4];
        const double b[4] = {hB[j], hB[16 + j], hB[32 + j], hB[48 + j]};
        const double c = hC[i * 16 + j], d = hD[i * 16 + j];
        double m0 = c;  // C first, fused multiply-adds in k order
        for (int k = 0; k < 4; ++k) m0 = __builtin_fma(a[k], b[k], m0);
        double m1 = c;  // ... in reverse k order
        for (int k = 3; k >= 0; --k) m1 = __builtin_fma(a[k], b[k], m1);
        double m2 = a[0] * b[0];  // products chained first, C last
        for (int k = 1; k < 4; ++k) m2 = __builtin_fma(a[k], b[k], m2);
        m2 += c;
        __float128 e = (__float128)c;  // exact sum, rounded once
        for (int k = 0; k < 4; ++k) e += (__float128)a[k] * (__float128)b[k];
        const double m3 = (double)e;
        const double m4 = __builtin_fma(a[3], b[3], __builtin_fma(a[2], b[2], 0.0)) + __builtin_fma(a[1], b[1], __builtin_fma(a[0], b[0], c));  // two halves
        bad[0] += d != m0; bad[1] += d != m1; bad[2] += d != m2; bad[3] += d != m3; bad[4] += d != m4;
        ++total;
      }
  }
  printf("summation order of v_mfma_f64_16x16x4_f64 on %d random entries, mismatches per model:\n", total);
  printf("  C then fma k=0..3: %d   C then fma k=3..0: %d   products chained, C last: %d   exact sum rounded once: %d   two halves: %d\n",
         bad[0], bad[1], bad[2], bad[3], bad[4]);
  return 0;
}

template <class F>
static float time_ms(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double *dout;
  CK(hipMalloc(&dout, 64 * sizeof(double)));
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  printf("probe        waves/SIMD    ms     cycles per trip (2.4 GHz)   [fma trip = 32 v_fma_f64, mfma trip = 8 v_mfma_f64_16x16x4]\n");
  for (int wps = 1; wps <= 2; ++wps) {
    dim3 grid(cus), block(256 * wps);
    auto rep = [&](const char *n, float ms) { printf("%-12s %d            %7.3f  %8.1f\n", n, wps, ms, ms * 1e-3 * 2.4e9 / REPS); };
    rep("fma", time_ms([&] { hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
    rep("mfma", time_ms([&] { hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
    rep("mix", time_ms([&] { hipLaunchKernelGGL(probe<2>, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
    if (wps == 2) rep("split", time_ms([&] { hipLaunchKernelGGL(probe<3>, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
  }
  hipLaunchKernelGGL(rounding, dim3(1), dim3(64), 0, 0, dout);
  double h[4];
  CK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
  printf("rounding rows (1,t,t,t) (t,t,t,1) (t,1,t,t) [C=0], (t,t,t,t) [C=1], t = 2^-53:\n");
  for (int i = 0; i < 4; ++i) printf("  row %d: %a\n", i, h[i]);
  if (summation_order_probe()) return 1;
  printf("  (k-ordered fma chain: 1+t rounds to 1 (ties-to-even) at every step -> rows 0 and 3 stay 0x1p+0; an exact sum rounded once gives 0x1.0000000000001p+0)\n");
  return 0;
}
