// cumask_sweep.hip — where does the headline kernel's time per workgroup go when more of the chip is busy?
// (VERDICT r04 "Next round" #1.)  Runs pbs_fft_wave_kernel<1,23> of a MEASUREMENT build of the library
// (variants/lib_ts.so = -DWAVE_PROBE_TS=1: every workgroup records HW_ID / XCC_ID and the 100 MHz + shader-clock
// timestamps around its CMUX loop) on streams made with hipExtStreamCreateWithCUMask, and a register-only f64 FMA
// kernel with a known cycle count as clock probe / power load.  One JSON object per configuration on stdout.
//
// Build (cross-compiles here):  hipcc --offload-arch=gfx950 -O2 tools/cumask_sweep.hip -o tools/bin/cumask_sweep -ldl
// Run on the GPU box:           tools/bin/cumask_sweep variants/lib_ts.so all > gpurun_out/cumask_sweep.jsonl
//                               tools/bin/cumask_sweep variants/lib_ts.so pmc <batch> <lwes_per_block>   (one launch; under rocprofv3 --pmc)
//
// CU-mask bit order (amdkfd mqd_symmetrically_map_cu_mask, multi-XCC): bit i -> XCC i % 8, then shader engine, then CU.
// Every mask used here keeps at least one CU of EVERY XCC enabled (workgroups are dealt to the XCCs round-robin whatever
// the mask says; an XCC without CUs could not run its share).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include <algorithm>
#include <map>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// ---------------------------------------------------------------- probe kernels
__device__ inline uint64_t where_am_i() {
  return (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((uint64_t)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) << 32);
}

// placement: many small workgroups that each linger a little, so that every enabled CU takes some
__global__ void where_kernel(uint64_t *rec) {
  if (threadIdx.x == 0) rec[blockIdx.x] = where_am_i();
  for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(127);
}

// f64 FMA load with a known cycle count: 8 waves per workgroup (two per SIMD), `iters` x 32 v_fma_f64 per wave and
// segment = 256 * iters SIMD cycles per segment (4 cycles per wave64 f64 FMA, two waves alternating); one workgroup per CU
// (the dynamic LDS request forbids a second one).  Records HW_ID|XCC then (realtime, memtime) at every segment edge.
__global__ void __launch_bounds__(512) fma_kernel(uint64_t *rec, int iters, int segs, int recw, uint32_t xcc_mask) {
  extern __shared__ char lds_dummy[];
  const uint64_t w = where_am_i();
  if (!((xcc_mask >> (uint32_t)(w >> 32)) & 1u)) return;
  uint64_t *r = rec + (size_t)blockIdx.x * recw;
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double b = 0.999999, c = 1e-9;
  if (threadIdx.x == 0) r[0] = w;
  for (int s = 0; s < segs; ++s) {
    if (threadIdx.x == 0) {
      r[2 + 2 * s] = __builtin_amdgcn_s_memrealtime();
      r[3 + 2 * s] = __builtin_amdgcn_s_memtime();
    }
    for (int it = 0; it < iters; ++it) {
#define F8 "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n" \
           "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
      asm volatile(F8 F8 F8 F8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
  }
  if (threadIdx.x == 0) {
    r[2 + 2 * segs] = __builtin_amdgcn_s_memrealtime();
    r[3 + 2 * segs] = __builtin_amdgcn_s_memtime();
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) r[1] = 1;
}

// ---------------------------------------------------------------- the library's C ABI (include/tfhe_hip_backend.h)
struct Lib {
  void *h;
  void *(*cuda_malloc)(uint64_t, uint32_t);
  void (*cuda_drop)(void *, uint32_t);
  void (*cuda_memcpy_async_to_gpu)(void *, const void *, uint64_t, void *, uint32_t);
  void (*cuda_synchronize_device)(uint32_t);
  void (*convert)(void *, uint32_t, void *, const void *, uint32_t, uint32_t, uint32_t, uint32_t);
  uint64_t (*scratch)(void *, uint32_t, int8_t **, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, bool, int);
  void (*pbs)(void *, uint32_t, void *, const void *, const void *, const void *, const void *, const void *, const void *,
              int8_t *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);
  void (*cleanup)(void *, uint32_t, int8_t **);
  void (*set_fft_kernel)(uint32_t);
  uint32_t (*last_kernel)();
  void (*probe)(uint64_t *, uint32_t);  // measurement builds only
};
template <class F> static void bind(void *h, F &f, const char *name, bool required = true) {
  f = (F)dlsym(h, name);
  if (!f && required) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
}
static Lib load(const char *path) {
  Lib L{};
  L.h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!L.h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(2); }
  bind(L.h, L.cuda_malloc, "cuda_malloc");
  bind(L.h, L.cuda_drop, "cuda_drop");
  bind(L.h, L.cuda_memcpy_async_to_gpu, "cuda_memcpy_async_to_gpu");
  bind(L.h, L.cuda_synchronize_device, "cuda_synchronize_device");
  bind(L.h, L.convert, "cuda_convert_lwe_programmable_bootstrap_key_64_async");
  bind(L.h, L.scratch, "scratch_cuda_programmable_bootstrap_64_async");
  bind(L.h, L.pbs, "cuda_programmable_bootstrap_64_async");
  bind(L.h, L.cleanup, "cleanup_cuda_programmable_bootstrap_64");
  bind(L.h, L.set_fft_kernel, "hip_backend_set_fft_kernel");
  bind(L.h, L.last_kernel, "hip_backend_last_pbs_kernel");
  bind(L.h, L.probe, "hip_probe_wave_timestamps", false);
  return L;
}

// PARAM_MESSAGE_2_CARRY_2 (tests/common.py C1)
constexpr uint32_t LWE_N = 918, GLWE_K = 1, POLY_N = 2048, BASE_LOG = 23, LEVEL = 1;
constexpr int MAX_B = 4096, RECW_PBS = 8;

struct Work {
  Lib L;
  void *bsk, *lwe_in, *lwe_out, *lut, *idx, *lidx;
  uint64_t *rec;  // device records of the PBS workgroups
};

static Work setup(const char *libpath) {
  Work W{};
  W.L = load(libpath);
  std::mt19937_64 rng(7);
  const size_t bsk_words = (size_t)LWE_N * LEVEL * 4 * POLY_N;
  std::vector<uint64_t> h(bsk_words);
  for (auto &x : h) x = rng();
  W.bsk = W.L.cuda_malloc(bsk_words * 8, 0);
  W.L.convert(nullptr, 0, W.bsk, h.data(), LWE_N, GLWE_K, LEVEL, POLY_N);
  W.L.cuda_synchronize_device(0);
  std::vector<uint64_t> in((size_t)MAX_B * (LWE_N + 1)), lut(2 * POLY_N), idx(MAX_B), z(MAX_B, 0);
  for (auto &x : in) x = rng();
  for (auto &x : lut) x = rng();
  for (int i = 0; i < MAX_B; ++i) idx[i] = i;
  W.lwe_in = W.L.cuda_malloc(in.size() * 8, 0);
  W.lwe_out = W.L.cuda_malloc((size_t)MAX_B * (POLY_N + 1) * 8, 0);
  W.lut = W.L.cuda_malloc(lut.size() * 8, 0);
  W.idx = W.L.cuda_malloc(MAX_B * 8, 0);
  W.lidx = W.L.cuda_malloc(MAX_B * 8, 0);
  CK(hipMemcpy(W.lwe_in, in.data(), in.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(W.lut, lut.data(), lut.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(W.idx, idx.data(), MAX_B * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(W.lidx, z.data(), MAX_B * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&W.rec, (size_t)MAX_B * RECW_PBS * 8));
  W.L.set_fft_kernel(2);  // the throughput kernel at every batch size
  return W;
}

static void pbs_launch(Work &W, hipStream_t st, int8_t *buf, uint32_t batch) {
  W.L.pbs(st, 0, W.lwe_out, W.idx, W.lut, W.lidx, W.lwe_in, W.idx, W.bsk, buf, LWE_N, GLWE_K, POLY_N, BASE_LOG, LEVEL, batch, 1, 0);
}

// ---------------------------------------------------------------- masks
struct Cu { int xcc, se, cu; };  // logical position of a mask bit
static int bit_of(int xcc, int se, int cu) { return ((cu * 4 + se) * 8) + xcc; }
using Mask = std::vector<uint32_t>;
static Mask mask_of(const std::vector<int> &bits) {
  Mask m(8, 0);
  for (int b : bits) m[b >> 5] |= 1u << (b & 31);
  return m;
}
static hipStream_t masked_stream(const Mask &m) {
  hipStream_t s;
  CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
  return s;
}

static std::string json_u64(const std::vector<uint64_t> &v) {
  std::string s = "[";
  for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + std::to_string(v[i]);
  return s + "]";
}

// one PBS configuration: `launches` back-to-back launches, records of the last one
static void run_pbs(Work &W, const char *name, const Mask *mask, uint32_t batch, uint32_t per_block, int launches = 2,
                    const char *note = "", const std::function<void()> &after_warmup = nullptr) {
  hipStream_t st;
  if (mask) st = masked_stream(*mask); else CK(hipStreamCreate(&st));
  int8_t *buf = nullptr;
  W.L.scratch(st, 0, &buf, LWE_N, GLWE_K, POLY_N, LEVEL, batch, true, 0);
  const uint32_t pb = per_block ? per_block : std::min<uint32_t>(4, std::max<uint32_t>(1, (batch + 255) / 256));
  const uint32_t blocks = (batch + pb - 1) / pb;
  if (W.L.probe) W.L.probe(W.rec, per_block);
  CK(hipMemsetAsync(W.rec, 0, (size_t)blocks * RECW_PBS * 8, st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  pbs_launch(W, st, buf, batch);  // warm-up
  CK(hipStreamSynchronize(st));
  if (after_warmup) after_warmup();
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < launches; ++i) pbs_launch(W, st, buf, batch);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint64_t> r((size_t)blocks * RECW_PBS);
  CK(hipMemcpy(r.data(), W.rec, r.size() * 8, hipMemcpyDeviceToHost));
  std::vector<uint64_t> t0(blocks), t1(blocks), c(blocks), hw(blocks);
  for (uint32_t b = 0; b < blocks; ++b) {
    t0[b] = r[b * RECW_PBS + 0]; t1[b] = r[b * RECW_PBS + 3];
    c[b] = r[b * RECW_PBS + 4] - r[b * RECW_PBS + 1];
    hw[b] = r[b * RECW_PBS + 2];
  }
  const uint64_t base = *std::min_element(t0.begin(), t0.end());
  for (uint32_t b = 0; b < blocks; ++b) { t0[b] -= base; t1[b] -= base; }
  printf("{\"what\": \"pbs\", \"name\": \"%s\", \"note\": \"%s\", \"batch\": %u, \"lwes_per_block\": %u, \"blocks\": %u, \"kernel_id\": %u, "
         "\"ms_per_launch\": %.4f, \"start_ticks\": %s, \"end_ticks\": %s, \"memtime_delta\": %s, \"hwid_xcc\": %s}\n",
         name, note, batch, pb, blocks, W.L.last_kernel(), ms / launches, json_u64(t0).c_str(), json_u64(t1).c_str(),
         json_u64(c).c_str(), json_u64(hw).c_str());
  fflush(stdout);
  W.L.cleanup(st, 0, &buf);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  CK(hipStreamDestroy(st));
}

// the FMA kernel on `blocks` workgroups of a stream; returns the device record (caller frees) — asynchronous
struct FmaRun { uint64_t *rec; int blocks, segs, recw, iters; };
static FmaRun fma_launch(hipStream_t st, int blocks, int iters, int segs, uint32_t xcc_mask) {
  FmaRun f{nullptr, blocks, segs, 2 + 2 * (segs + 1), iters};
  CK(hipMalloc(&f.rec, (size_t)blocks * f.recw * 8));
  CK(hipMemsetAsync(f.rec, 0, (size_t)blocks * f.recw * 8, st));
  CK(hipFuncSetAttribute((const void *)fma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(512), 120 * 1024, st, f.rec, iters, segs, f.recw, xcc_mask);
  CK(hipGetLastError());
  return f;
}
static void fma_report(const char *name, FmaRun f, const char *note = "") {
  std::vector<uint64_t> r((size_t)f.blocks * f.recw);
  CK(hipMemcpy(r.data(), f.rec, r.size() * 8, hipMemcpyDeviceToHost));
  CK(hipFree(f.rec));
  // per workgroup: hwid, start tick, then per segment realtime ticks and memtime ticks
  uint64_t base = ~0ull;
  for (int b = 0; b < f.blocks; ++b) if (r[(size_t)b * f.recw]) base = std::min(base, r[(size_t)b * f.recw + 2]);
  printf("{\"what\": \"fma\", \"name\": \"%s\", \"note\": \"%s\", \"blocks\": %d, \"iters\": %d, \"segs\": %d, \"cycles_per_seg\": %lld, \"wgs\": [",
         name, note, f.blocks, f.iters, f.segs, 256ll * f.iters);
  bool first = true;
  for (int b = 0; b < f.blocks; ++b) {
    const uint64_t *p = &r[(size_t)b * f.recw];
    if (!p[0]) continue;  // exited (XCC filter)
    std::vector<uint64_t> rt, mt;
    for (int s = 0; s < f.segs; ++s) { rt.push_back(p[2 + 2 * (s + 1)] - p[2 + 2 * s]); mt.push_back(p[3 + 2 * (s + 1)] - p[3 + 2 * s]); }
    printf("%s{\"hwid_xcc\": %llu, \"start\": %llu, \"rt\": %s, \"mt\": %s}", first ? "" : ",", (unsigned long long)p[0],
           (unsigned long long)(p[2] - base), json_u64(rt).c_str(), json_u64(mt).c_str());
    first = false;
  }
  printf("]}\n");
  fflush(stdout);
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: cumask_sweep <lib.so> all | pmc <batch> <lwes_per_block> | quick\n"); return 2; }
  const std::string mode = argv[2];
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  Work W = setup(argv[1]);
  if (mode == "pmc") {
    const uint32_t batch = (uint32_t)atoi(argv[3]), per_block = argc > 4 ? (uint32_t)atoi(argv[4]) : 0;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int8_t *buf = nullptr;
    W.L.scratch(st, 0, &buf, LWE_N, GLWE_K, POLY_N, LEVEL, batch, true, 0);
    if (W.L.probe) W.L.probe(nullptr, per_block);
    pbs_launch(W, st, buf, batch);
    CK(hipStreamSynchronize(st));
    W.L.cleanup(st, 0, &buf);
    return 0;
  }
  printf("{\"what\": \"device\", \"name\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"probe_build\": %s}\n", prop.name,
         prop.multiProcessorCount, prop.clockRate, W.L.probe ? "true" : "false");
  const int NCU = prop.multiProcessorCount;  // 256
  const int PER_XCC = NCU / 8, CUS_PER_SE = PER_XCC / 4;

  // ---- 0. logical mask bit -> physical (xcc, se, cu_id): leave one bit out, see which CU stays empty
  std::vector<int> all_bits;
  for (int b = 0; b < NCU; ++b) all_bits.push_back(b);
  auto placement = [&](const Mask &m) {
    hipStream_t s = masked_stream(m);
    const int nb = 8192;
    uint64_t *d;
    CK(hipMalloc(&d, nb * 8));
    hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(64), 0, s, d);
    CK(hipStreamSynchronize(s));
    std::vector<uint64_t> h(nb);
    CK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
    CK(hipFree(d));
    CK(hipStreamDestroy(s));
    std::map<uint32_t, int> seen;  // key: xcc<<16 | se<<8 | cu
    for (uint64_t w : h) {
      const uint32_t hwid = (uint32_t)w, xcc = (uint32_t)(w >> 32);
      seen[(xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 7) | ((hwid >> 8) & 15)]++;
    }
    return seen;
  };
  std::vector<uint32_t> phys_of_bit(NCU, 0xffffffffu);
  {
    auto full = placement(mask_of(all_bits));
    std::vector<uint64_t> keys;
    for (auto &kv : full) keys.push_back(kv.first);
    printf("{\"what\": \"placement_full\", \"distinct_cus\": %zu, \"keys_xcc16_se8_sh7_cu\": %s}\n", keys.size(), json_u64(keys).c_str());
    fflush(stdout);
    if (mode == "all") {
      for (int b = 0; b < NCU; ++b) {
        std::vector<int> bits;
        for (int x = 0; x < NCU; ++x) if (x != b) bits.push_back(x);
        auto got = placement(mask_of(bits));
        for (auto &kv : full) if (!got.count(kv.first)) phys_of_bit[b] = kv.first;
      }
      std::vector<uint64_t> v(phys_of_bit.begin(), phys_of_bit.end());
      printf("{\"what\": \"bit_to_physical\", \"key_of_bit\": %s}\n", json_u64(v).c_str());
      fflush(stdout);
    }
  }

  // ---- 1. clock under an f64 load: idle chip (one CU per XCC), full chip short and long
  {
    std::vector<int> one_per_xcc;
    for (int x = 0; x < 8; ++x) one_per_xcc.push_back(bit_of(x, 0, 0));
    hipStream_t s1 = masked_stream(mask_of(one_per_xcc));
    FmaRun f = fma_launch(s1, 8, 2048, 16, 0xff);
    CK(hipStreamSynchronize(s1));
    fma_report("fma_8cus_idle_chip", f, "one CU per XCC");
    CK(hipStreamDestroy(s1));
    hipStream_t s2;
    CK(hipStreamCreate(&s2));
    f = fma_launch(s2, NCU, 2048, 160, 0xff);  // 160 segments of ~0.22 ms: ~35 ms of full-chip f64
    CK(hipStreamSynchronize(s2));
    fma_report("fma_256cus_35ms", f, "full chip, unmasked stream");
    CK(hipStreamDestroy(s2));
  }

  // ---- 2. the headline kernel, natural stream: LWEs per workgroup x workgroups
  run_pbs(W, "nat_b4_pb4", nullptr, 4, 4);
  run_pbs(W, "nat_b32_pb4", nullptr, 32, 4);
  run_pbs(W, "nat_b256_pb4", nullptr, 256, 4, 2, "64 workgroups of 4 LWEs");
  run_pbs(W, "nat_b512_pb4", nullptr, 512, 4, 2, "128 workgroups of 4 LWEs");
  run_pbs(W, "nat_b768_pb4", nullptr, 768, 4, 2, "192 workgroups of 4 LWEs");
  run_pbs(W, "nat_b1024_pb4", nullptr, 1024, 4, 2, "256 workgroups of 4 LWEs");
  run_pbs(W, "nat_b256_pb1", nullptr, 256, 1, 2, "256 workgroups of 1 LWE (the library's choice at batch 256)");
  run_pbs(W, "nat_b512_pb2", nullptr, 512, 2, 2, "256 workgroups of 2 LWEs");
  run_pbs(W, "nat_b768_pb3", nullptr, 768, 3, 2, "256 workgroups of 3 LWEs");
  run_pbs(W, "nat_b4096_pb4", nullptr, 4096, 4, 2, "1024 workgroups of 4 LWEs: four rounds");
  if (mode == "quick") return 0;

  // ---- 3. CU masks, 256 workgroups of 4 LWEs each (32 per XCC): enabled CUs take them in rounds
  auto bits_where = [&](auto pred) {
    std::vector<int> bits;
    for (int x = 0; x < 8; ++x) for (int se = 0; se < 4; ++se) for (int cu = 0; cu < CUS_PER_SE; ++cu)
      if (pred(x, se, cu) || (se == 3 && cu == CUS_PER_SE - 1)) bits.push_back(bit_of(x, se, cu));  // the last CU of every XCC always on
    return bits;
  };
  struct Cfg { const char *name; std::vector<int> bits; const char *note; };
  std::vector<Cfg> cfgs;
  cfgs.push_back({"mask_all", all_bits, "all 256 CUs through a masked stream"});
  cfgs.push_back({"mask_xcc0to3_full", bits_where([](int x, int, int) { return x < 4; }), "XCC 0-3 full, one CU on each of XCC 4-7"});
  cfgs.push_back({"mask_xcc0_full", bits_where([](int x, int, int) { return x == 0; }), "XCC 0 full, one CU on each other XCC"});
  cfgs.push_back({"mask_even_cus", bits_where([](int, int, int cu) { return (cu & 1) == 0; }), "logical CUs 0,2,4,6 of every SE: 128 CUs, one of every assumed pair"});
  cfgs.push_back({"mask_cu_pairs", bits_where([](int, int, int cu) { return (cu & 2) == 0; }), "logical CUs 0,1,4,5 of every SE: 128 CUs, both CUs of half the assumed pairs"});
  cfgs.push_back({"mask_low_cus", bits_where([](int, int, int cu) { return cu < 4; }), "logical CUs 0-3 of every SE: 128 CUs"});
  cfgs.push_back({"mask_se01", bits_where([](int, int se, int) { return se < 2; }), "shader engines 0,1 of every XCC full: 128 CUs"});
  cfgs.push_back({"mask_2_per_se_pair", bits_where([](int, int, int cu) { return cu < 2; }), "logical CUs 0,1 of every SE: 64 CUs"});
  cfgs.push_back({"mask_2_per_se_apart", bits_where([](int, int, int cu) { return cu == 0 || cu == 2; }), "logical CUs 0,2 of every SE: 64 CUs"});
  cfgs.push_back({"mask_192", bits_where([](int, int, int cu) { return cu < 6; }), "logical CUs 0-5 of every SE: 192 CUs"});
  for (auto &c : cfgs) {
    Mask m = mask_of(c.bits);
    auto seen = placement(m);
    std::vector<uint64_t> keys;
    for (auto &kv : seen) keys.push_back(kv.first);
    printf("{\"what\": \"placement\", \"name\": \"%s\", \"mask_bits\": %zu, \"distinct_cus\": %zu, \"keys_xcc16_se8_sh7_cu\": %s}\n", c.name,
           c.bits.size(), keys.size(), json_u64(keys).c_str());
    run_pbs(W, c.name, &m, 1024, 4, 1, c.note);
  }

  // ---- 4. the headline kernel on XCC 0-3 while XCC 4-7 run the f64 load (power / clock coupling across XCCs)
  {
    // PBS: XCC 0-3 full + the last CU of XCC 4-7; load: XCC 4-7 without that CU (+ the last CU of XCC 0-3, whose share
    // of the load's workgroups leaves at once through the kernel's XCC filter)
    Mask mp = mask_of(bits_where([](int x, int, int) { return x < 4; }));
    std::vector<int> vb;
    for (int x = 0; x < 8; ++x) for (int se = 0; se < 4; ++se) for (int cu = 0; cu < CUS_PER_SE; ++cu) {
      const bool last = se == 3 && cu == CUS_PER_SE - 1;
      if ((x >= 4 && !last) || (x < 4 && last)) vb.push_back(bit_of(x, se, cu));
    }
    Mask mv = mask_of(vb);
    hipStream_t sv = masked_stream(mv);
    FmaRun f{};
    run_pbs(W, "pbs_xcc0to3_while_fma_xcc4to7", &mp, 1024, 4, 1, "f64 load on 31 CUs of each of XCC 4-7 during the launch",
            [&]() { f = fma_launch(sv, 8 * (PER_XCC - 1), 2048, 500, 0xf0); });  // ~110 ms
    CK(hipStreamSynchronize(sv));
    fma_report("fma_xcc4to7_during_pbs", f, "the load itself: its clock while XCC 0-3 run the PBS");
    CK(hipStreamDestroy(sv));
  }
  return 0;
}
