// microbench.hip — issue-rate / latency probes for the instructions the PBS wave kernel is made of.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench ; run on the GPU box.
// Every probe runs WPS waves per SIMD on every CU (grid = 256 CUs * 4 SIMDs * WPS / waves-per-block)
// and reports cycles per wave-instruction assuming the 2.4 GHz shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

constexpr int REPS = 4096;

// 8 independent chains, 32 instructions per loop trip
#define CHAIN8(INSTR)                                                                     \
  asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)    \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)    \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)    \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)    \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]),  \
                 "+v"(a[6]), "+v"(a[7])                                                   \
               : "v"(b), "v"(c))

#define FMA64(i) "v_fma_f64 %" #i ", %8, %9, %" #i "\n"
#define ADD64(i) "v_add_f64 %" #i ", %8, %" #i "\n"
#define MUL64(i) "v_mul_f64 %" #i ", %8, %" #i "\n"
#define RND64(i) "v_rndne_f64 %" #i ", %" #i "\n"
#define LDEXP64(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define FLOOR64(i) "v_floor_f64 %" #i ", %" #i "\n"

__global__ void k_fma64(double *out, double b, double c) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int r = 0; r < REPS; ++r) CHAIN8(FMA64);
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void k_add64(double *out, double b, double c) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int r = 0; r < REPS; ++r) CHAIN8(ADD64);
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void k_mul64(double *out, double b, double c) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int r = 0; r < REPS; ++r) CHAIN8(MUL64);
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void k_rnd64(double *out, double b, double c) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i + 0.3;
  for (int r = 0; r < REPS; ++r) CHAIN8(RND64);
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void k_ldexp64(double *out, double b, double c) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i + 0.3;
  for (int r = 0; r < REPS; ++r) CHAIN8(LDEXP64);
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void k_floor64(double *out, double b, double c) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i + 0.3;
  for (int r = 0; r < REPS; ++r) CHAIN8(FLOOR64);
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678) out[0] = s;
}

// single dependent chain (latency)
__global__ void k_fma64_dep(double *out, double b, double c) {
  double a = threadIdx.x;
  for (int r = 0; r < REPS; ++r)
    asm volatile(
        "v_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\n"
        "v_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\n"
        "v_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\n"
        "v_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\n"
        "v_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\n"
        "v_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\n"
        "v_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\n"
        "v_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\nv_fma_f64 %0, %1, %2, %0\n"
        : "+v"(a)
        : "v"(b), "v"(c));
  if (a == 12345.678) out[0] = a;
}

// 32-bit integer ops, 8 chains
#define CHAIN8I(INSTR)                                                                    \
  asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)    \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)    \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)    \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)    \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]),  \
                 "+v"(a[6]), "+v"(a[7])                                                   \
               : "v"(b), "v"(c))
#define ADD32(i) "v_add_u32 %" #i ", %8, %" #i "\n"
#define XOR32(i) "v_xor_b32 %" #i ", %8, %" #i "\n"
#define LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define CNDMASK(i) "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"
#define SUBCO(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define CVTF64I32(i) "v_cvt_i32_f64 %" #i ", %8\n"

__global__ void k_add32(uint32_t *out, uint32_t b, uint32_t c) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int r = 0; r < REPS; ++r) CHAIN8I(ADD32);
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345678u) out[0] = s;
}
__global__ void k_lshladd(uint32_t *out, uint32_t b, uint32_t c) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int r = 0; r < REPS; ++r) CHAIN8I(LSHLADD);
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345678u) out[0] = s;
}
__global__ void k_cndmask(uint32_t *out, uint32_t b, uint32_t c) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int r = 0; r < REPS; ++r) asm volatile(
      CNDMASK(0) CNDMASK(1) CNDMASK(2) CNDMASK(3) CNDMASK(4) CNDMASK(5) CNDMASK(6) CNDMASK(7)
      CNDMASK(0) CNDMASK(1) CNDMASK(2) CNDMASK(3) CNDMASK(4) CNDMASK(5) CNDMASK(6) CNDMASK(7)
      CNDMASK(0) CNDMASK(1) CNDMASK(2) CNDMASK(3) CNDMASK(4) CNDMASK(5) CNDMASK(6) CNDMASK(7)
      CNDMASK(0) CNDMASK(1) CNDMASK(2) CNDMASK(3) CNDMASK(4) CNDMASK(5) CNDMASK(6) CNDMASK(7)
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
      : "v"(b), "v"(c) : "vcc");
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345678u) out[0] = s;
}
__global__ void k_subco(uint32_t *out, uint32_t b, uint32_t c) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int r = 0; r < REPS; ++r) asm volatile(
      SUBCO(0) SUBCO(1) SUBCO(2) SUBCO(3) SUBCO(4) SUBCO(5) SUBCO(6) SUBCO(7)
      SUBCO(0) SUBCO(1) SUBCO(2) SUBCO(3) SUBCO(4) SUBCO(5) SUBCO(6) SUBCO(7)
      SUBCO(0) SUBCO(1) SUBCO(2) SUBCO(3) SUBCO(4) SUBCO(5) SUBCO(6) SUBCO(7)
      SUBCO(0) SUBCO(1) SUBCO(2) SUBCO(3) SUBCO(4) SUBCO(5) SUBCO(6) SUBCO(7)
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
      : "v"(b), "v"(c) : "vcc");
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345678u) out[0] = s;
}
__global__ void k_cvt_i32_f64(uint32_t *out, double b, uint32_t c) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int r = 0; r < REPS; ++r) asm volatile(
      CVTF64I32(0) CVTF64I32(1) CVTF64I32(2) CVTF64I32(3) CVTF64I32(4) CVTF64I32(5) CVTF64I32(6) CVTF64I32(7)
      CVTF64I32(0) CVTF64I32(1) CVTF64I32(2) CVTF64I32(3) CVTF64I32(4) CVTF64I32(5) CVTF64I32(6) CVTF64I32(7)
      CVTF64I32(0) CVTF64I32(1) CVTF64I32(2) CVTF64I32(3) CVTF64I32(4) CVTF64I32(5) CVTF64I32(6) CVTF64I32(7)
      CVTF64I32(0) CVTF64I32(1) CVTF64I32(2) CVTF64I32(3) CVTF64I32(4) CVTF64I32(5) CVTF64I32(6) CVTF64I32(7)
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
      : "v"(b), "v"(c));
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345678u) out[0] = s;
}

// LDS: 16 x b128 writes then 16 x b128 reads per trip (the transposes), conflict-free addresses
__global__ void k_lds_transpose(double *out, int reps) {
  extern __shared__ double2 sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double2 *buf = sm + wave * 1088;
  double2 d[16];
  for (int r = 0; r < 16; ++r) d[r] = double2{(double)(lane + r), (double)r};
  for (int it = 0; it < reps; ++it) {
    double2 *p1 = buf + lane;
    for (int r = 0; r < 16; ++r) p1[68 * r] = d[r];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const double2 *p2 = buf + (lane >> 2) * 68 + (lane & 3);
    for (int r = 0; r < 16; ++r) d[r] = p2[4 * r];
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(d[r].x), "+v"(d[r].y));
  }
  double s = 0;
  for (int r = 0; r < 16; ++r) s += d[r].x + d[r].y;
  if (s == 12345.678) out[0] = s;
}

// dependent LDS read chain (latency)
__global__ void k_lds_latency(uint32_t *out, int reps) {
  __shared__ uint32_t sm[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = (i * 17 + 5) & 1023;
  __syncthreads();
  uint32_t p = threadIdx.x;
  for (int it = 0; it < reps; ++it) p = sm[p];
  if (p == 0xFFFFFFFFu) out[0] = p;
}

// dependent global (L2-resident) load chain, one lane per wave active
__global__ void k_l2_latency(const uint32_t *chain, uint32_t *out, int reps) {
  uint32_t p = (blockIdx.x * 977u) & 0xFFFFu;
  if ((threadIdx.x & 63) == 0) {
    for (int it = 0; it < reps; ++it) p = __builtin_nontemporal_load(chain + p) * 0 + chain[p];
  }
  if (p == 0xFFFFFFFFu) out[0] = p;
}

// streaming 16-byte loads of a 32 KB slice shared by all waves (the key-row pattern), 32 in flight
__global__ void k_key_stream(const double2 *key, double *out, int reps, int span) {
  const int lane = threadIdx.x & 63;
  double2 acc{0, 0};
  for (int it = 0; it < reps; ++it) {
    const double2 *b = key + (size_t)(it % span) * 2048 + lane;
    double2 v[32];
    for (int j = 0; j < 32; ++j) v[j] = b[j * 64];
    for (int j = 0; j < 32; ++j) {
      acc.x += v[j].x;
      acc.y += v[j].y;
    }
  }
  if (acc.x == 12345.678) out[0] = acc.y;
}

template <class F>
static float time_ms(F launch, int iters = 3) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int i = 0; i < iters; ++i) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const double GHZ = 2.4;
  double *dout;
  CK(hipMalloc(&dout, 1024));
  uint32_t *uout = (uint32_t *)dout;
  printf("probe                          waves/SIMD  ms      cycles/wave-instr (at %.1f GHz)\n", GHZ);
  for (int wps = 1; wps <= 4; wps *= 2) {
    const dim3 grid(256 * wps), block(256);  // 4 waves per block = 1 per SIMD per block
    const double n_instr = (double)REPS * 32;
    auto rep = [&](const char *name, float ms) {
      // a SIMD executes wps waves' instructions back to back
      printf("%-30s %d           %7.3f  %.2f\n", name, wps, ms, ms * 1e-3 * GHZ * 1e9 / (n_instr * wps));
    };
    rep("v_fma_f64 (8 chains)", time_ms([&] { hipLaunchKernelGGL(k_fma64, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
    rep("v_add_f64", time_ms([&] { hipLaunchKernelGGL(k_add64, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
    rep("v_mul_f64", time_ms([&] { hipLaunchKernelGGL(k_mul64, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
    rep("v_rndne_f64", time_ms([&] { hipLaunchKernelGGL(k_rnd64, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
    rep("v_ldexp_f64", time_ms([&] { hipLaunchKernelGGL(k_ldexp64, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
    rep("v_floor_f64", time_ms([&] { hipLaunchKernelGGL(k_floor64, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
    rep("v_fma_f64 (1 dependent chain)", time_ms([&] { hipLaunchKernelGGL(k_fma64_dep, grid, block, 0, 0, dout, 1.0000001, 1e-9); }));
    rep("v_add_u32", time_ms([&] { hipLaunchKernelGGL(k_add32, grid, block, 0, 0, uout, 3u, 5u); }));
    rep("v_lshl_add_u32", time_ms([&] { hipLaunchKernelGGL(k_lshladd, grid, block, 0, 0, uout, 3u, 5u); }));
    rep("v_cndmask_b32", time_ms([&] { hipLaunchKernelGGL(k_cndmask, grid, block, 0, 0, uout, 3u, 5u); }));
    rep("v_sub_co_u32", time_ms([&] { hipLaunchKernelGGL(k_subco, grid, block, 0, 0, uout, 3u, 5u); }));
    rep("v_cvt_i32_f64", time_ms([&] { hipLaunchKernelGGL(k_cvt_i32_f64, grid, block, 0, 0, uout, 3.5, 5u); }));
  }
  // LDS transposes: cycles per (16 writes + 16 reads) round
  for (int wpc = 4; wpc <= 8; wpc *= 2) {
    const int reps = 20000;
    CK(hipFuncSetAttribute((const void *)k_lds_transpose, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 1088 * 16));
    float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_transpose, dim3(256), dim3(64 * wpc), wpc * 1088 * 16, 0, dout, reps); });
    printf("lds transpose 16w+16r b128     %d waves/CU  %7.3f  %.1f cycles/round/wave  (%.1f per CU-round)\n", wpc, ms,
           ms * 1e-3 * GHZ * 1e9 / reps, ms * 1e-3 * GHZ * 1e9 / reps / wpc);
  }
  {
    const int reps = 100000;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_latency, dim3(256), dim3(64), 0, 0, uout, reps); });
    printf("lds dependent read latency                 %7.3f  %.1f cycles\n", ms, ms * 1e-3 * GHZ * 1e9 / reps);
  }
  {
    // 64K-entry chain (256 KB), stride-permuted so consecutive hops land on different lines
    std::vector<uint32_t> h(65536);
    for (uint32_t i = 0; i < 65536; ++i) h[i] = (i * 4099u + 64u) & 0xFFFFu;
    uint32_t *dchain;
    CK(hipMalloc(&dchain, h.size() * 4));
    CK(hipMemcpy(dchain, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int reps = 20000;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_l2_latency, dim3(256), dim3(64), 0, 0, dchain, uout, reps); });
    printf("global dependent load (256 KB set, 2 loads/hop) %7.3f  %.1f cycles per load\n", ms,
           ms * 1e-3 * GHZ * 1e9 / reps / 2);
  }
  {
    double2 *key;
    const int span = 918;
    CK(hipMalloc(&key, (size_t)span * 2048 * 16));
    CK(hipMemset(key, 0, (size_t)span * 2048 * 16));
    for (int wpc = 4; wpc <= 8; wpc *= 2) {
      const int reps = 918 * 2;
      float ms = time_ms([&] { hipLaunchKernelGGL(k_key_stream, dim3(256), dim3(64 * wpc), 0, 0, key, dout, reps, span); });
      printf("key stream 32x1KiB per wave-trip %d waves/CU %7.3f  %.0f cycles per trip, %.2f TB/s L2->CU aggregate\n", wpc, ms,
             ms * 1e-3 * GHZ * 1e9 / reps, (double)reps * 32768 * wpc * 256 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
