// abi.hip — extern "C" entry points of libtfhe_hip_backend.so (declared, with the reference
// interface each one replaces, in include/tfhe_hip_backend.h).
#include "../../include/tfhe_hip_backend.h"
#include "kernels.h"
#include "arena.h"
#include "profile.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

using namespace tfhe_hip;

namespace {
inline hipStream_t S(void *stream) { return static_cast<hipStream_t>(stream); }
inline void set_device(uint32_t gpu_index) { HX_CHECK(hipSetDevice((int)gpu_index)); }

std::atomic<uint32_t> g_fft_kernel_choice{0};
std::atomic<uint32_t> g_last_pbs_kernel{0};

constexpr uint32_t kPbsMagic = 0x50425331;   // "PBS1"
constexpr uint32_t kMbMagic = 0x4d425031;    // "MBP1"

// Host-side descriptor behind the opaque `int8_t *buffer` of the scratch/cleanup triple
// (the reference's pbs_buffer<Torus, CLASSICAL>, cuda/include/pbs/pbs_utilities.h:100-260).
// The whole CMUX loop runs on-chip, so the classic PBS needs no global scratch — except for the rings of
// 8192 and 16384 coefficients, whose accumulator is a per-sample device buffer (acc_scratch).
struct PbsBuffer {
  uint32_t magic;
  uint32_t lwe_dimension, glwe_dimension, polynomial_size, level_count, max_samples;
  uint32_t ms_type;
  bool gpu_memory_allocated;
  FftTables fft;
  NttTables ntt;
  uint64_t *acc_scratch = nullptr;
  uint64_t *split_acc = nullptr;  // exact engine, split-key form: (k+1) N accumulator words per sample in device memory
  uint32_t *split_flag = nullptr; // ... and its round-off flag (PbsArgs::roundoff_flag), allocated with split_acc
  uint32_t *split_bad = nullptr;  // ... one word per sample: the ciphertexts of the last launch the integer kernel had to redo
  bool split_unrecovered = false; // a launch ran without the NTT-domain twin of its key: a raised flag is then fatal
  // hip_keyswitch_programmable_bootstrap_chain_64_async: keyswitch operands written by the bootstrap of the previous
  // call for ITS outputs (what they are valid for is recorded; anything else falls back to the digit pass)
  int8_t *emit_a = nullptr;
  int32_t *emit_suma = nullptr;
  size_t emit_bytes = 0;
  std::vector<void *> emit_retired;  // smaller emit buffers a captured graph may still write: freed with the scratch
  struct {
    bool valid = false;
    const void *array = nullptr, *indexes = nullptr;
    uint32_t count = 0, steps = 0, base_log = 0, level = 0;
  } emitted;
  uint64_t *ks_out = nullptr;  // hip_keyswitch_programmable_bootstrap_64_async: the keyswitched LWEs (small key)
  uint64_t *trivial = nullptr; // 0, 1, ..., max_samples - 1 (indexes of that list)
};
struct MultiBitBuffer {
  uint32_t magic;
  uint32_t glwe_dimension, polynomial_size, level_count, max_samples;
  bool gpu_memory_allocated;
  FftTables fft;
  uint32_t chunk;
  uint64_t *acc;   // latency path: the accumulators crossing passes
  uint32_t *pace = nullptr;  // throughput kernel: per-XCD progress counters (PbsArgs::pace)
  // latency path (small batches): the keybundles of a pass, lat_bytes / (batch * kb_per_sample) groups at a time
  cplx *kb_lat = nullptr;
  uint32_t lat_samples = 0;
  uint64_t lat_bytes = 0, kb_per_sample = 0;
};

// max_n: 16384 for the classic and the multi-bit f64 PBS with k = 1 (the reference's kernels support rings up to
// 2^14), 4096 for the NTT / exact engines
void check_pow2_poly(uint32_t N, uint32_t max_n = 4096) {
  HX_PANIC_IF_FALSE(N >= 256 && N <= max_n && (N & (N - 1)) == 0,
                    "polynomial_size %u not supported by the MI355X PBS (256..%u, power of two)", N, max_n);
}

constexpr uint64_t kMultiBitLatencyBytes = 256ull << 20;  // keybundle scratch of the multi-bit latency path
constexpr uint32_t kMultiBitLatencyMaxBatch = 128;  // multi-bit PBS: two-launch latency path up to this many LWEs
// N = 2048, k = 1 (slot-walking keybundle kernel + latency kernel): measured against the throughput kernel
// (tools/measure_all.py mbcross), g = 4: 256 LWEs 4.35 vs 6.24 ms, 384: 6.76 vs 5.92; g = 3: 7.43 vs 8.52, 11.98 vs 9.89
constexpr uint32_t kMultiBitLatencyMaxBatch2048 = 256;
std::atomic<uint32_t> g_multibit_latency_groups{0};  // test hook: cap of the groups per pass (0 = what the scratch holds)
constexpr uint32_t kLatencyKernelMaxBatch = 256;  // measured (tools/measure_all.py latency): 3.7-3.9 ms vs 5.9 ms up to 256 LWEs, slower beyond

PbsArgs make_args(void *lwe_array_out, void const *lwe_output_indexes, void const *lut_vector,
                  void const *lut_vector_indexes, void const *lwe_array_in, void const *lwe_input_indexes,
                  void const *bootstrapping_key, uint32_t lwe_dimension, uint32_t base_log, uint32_t level_count,
                  uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride, uint32_t ms_type) {
  PbsArgs a;
  a.lwe_out = (uint64_t *)lwe_array_out;
  a.out_idx = (const uint64_t *)lwe_output_indexes;
  a.lut = (const uint64_t *)lut_vector;
  a.lut_idx = (const uint64_t *)lut_vector_indexes;
  a.lwe_in = (const uint64_t *)lwe_array_in;
  a.in_idx = (const uint64_t *)lwe_input_indexes;
  a.bsk = bootstrapping_key;
  a.n = lwe_dimension;
  a.base_log = base_log;
  a.level = level_count;
  a.num_samples = num_samples;
  a.num_many_lut = num_many_lut;
  a.lut_stride = lut_stride;
  a.ms_type = ms_type;
  return a;
}
}  // namespace

extern "C" {

// =========================================================================== device runtime
void *cuda_create_stream_ffi(uint32_t gpu_index) {
  set_device(gpu_index);
  hipStream_t s;
  HX_CHECK(hipStreamCreate(&s));
  arena_register_stream((int)gpu_index, s);
  return s;
}
void cuda_destroy_stream(void *stream, uint32_t gpu_index) {
  set_device(gpu_index);
  HX_CHECK(hipStreamSynchronize(S(stream)));
  ksd_release_stream((int)gpu_index, S(stream));
  arena_release_stream((int)gpu_index, S(stream));
  HX_CHECK(hipStreamDestroy(S(stream)));
}
void cuda_synchronize_stream(void *stream, uint32_t gpu_index) {
  set_device(gpu_index);
  HX_CHECK(hipStreamSynchronize(S(stream)));
}
// ---- the split-key exact engine's way out of a raised round-off flag: the NTT-domain form of the same key (the integer
// Goldilocks kernel's operand), made next to the split form by hip_convert_lwe_programmable_bootstrap_key_ntt64_split_async
// and kept by the library for as long as the split key's device memory is neither dropped nor overwritten
struct SplitTwin {
  int device;
  const void *split_key;
  size_t split_bytes;
  void *ntt_key;
};
static std::mutex g_twin_mutex;
static std::vector<SplitTwin> g_split_twins;
static void split_twin_forget_range(int device, const void *p, size_t bytes) {
  if (p == nullptr) return;
  std::vector<void *> dead;
  {
    std::lock_guard<std::mutex> lock(g_twin_mutex);
    const char *lo = (const char *)p, *hi = lo + (bytes ? bytes : 1);
    for (size_t i = 0; i < g_split_twins.size();) {
      const char *klo = (const char *)g_split_twins[i].split_key, *khi = klo + g_split_twins[i].split_bytes;
      if (g_split_twins[i].device == device && klo < hi && lo < khi) {
        dead.push_back(g_split_twins[i].ntt_key);
        g_split_twins.erase(g_split_twins.begin() + i);
      } else {
        ++i;
      }
    }
  }
  for (void *d : dead) device_free_sync(d);  // synchronises the device: no launch still reads it
}
static const void *split_twin_of(int device, const void *split_key) {
  std::lock_guard<std::mutex> lock(g_twin_mutex);
  for (const SplitTwin &t : g_split_twins)
    if (t.device == device && t.split_key == split_key) return t.ntt_key;
  return nullptr;
}
// device memory [p, p + bytes) is about to be freed or written: what the library derived from it goes
static void device_range_changes(int device, const void *p, size_t bytes) {
  ksm_invalidate_range(device, p, bytes);
  split_twin_forget_range(device, p, bytes);
}
uint32_t cuda_is_available(void) { return hipSetDevice(0) == hipSuccess; }
void *cuda_malloc(uint64_t size, uint32_t gpu_index) {
  set_device(gpu_index);
  return device_alloc_sync(size);  // hipMalloc (an arena block in red-zone mode: cuda_drop knows both)
}
// Stream-ordered allocations (the reference: cudaMallocAsync on the device pool, tfhe-cuda-common/cuda/src/device.cu:176-226;
// CudaVec::new_async / Drop allocate and drop per operation, tfhe/src/core_crypto/gpu/vec.rs:94-150,487-495).  Default: the
// library's own arena (arena.hip) — an allocation is an enqueue, never a device synchronisation, and works under stream
// capture.  hipMallocAsync's pool is not usable on this runtime (ROCm 7.2.0, MI355X: a LIVE pool allocation changed after
// other pool allocations were freed and re-made, in a program that uses nothing but the runtime — tools/probes/
// pool_probe.hip, profiles/r04h_ks_cpp_diag*.txt); it and the plain hipMalloc / hipFree pair of round 4 stay selectable for
// comparison: TFHE_HIP_MALLOC_ASYNC=arena (default) | sync | pool | pool_hipfree.
static std::mutex g_pool_mutex;
static std::unordered_map<const void *, size_t> g_pool_allocations;
static int malloc_async_mode() {  // 3 = arena (default), 1 = hipMalloc, 0 = pool + tracked free, 2 = pool + hipFree
  static const int mode = [] {
    const char *e = std::getenv("TFHE_HIP_MALLOC_ASYNC");
    if (e == nullptr || !std::strcmp(e, "arena")) return 3;
    if (!std::strcmp(e, "sync")) return 1;
    if (!std::strcmp(e, "pool")) return 0;
    if (!std::strcmp(e, "pool_hipfree")) return 2;
    HX_PANIC("TFHE_HIP_MALLOC_ASYNC=%s: expected arena, sync, pool or pool_hipfree", e);
    return 0;
  }();
  return mode;
}
void *cuda_malloc_async(uint64_t size, void *stream, uint32_t gpu_index) {
  set_device(gpu_index);
  void *p = nullptr;
  if (malloc_async_mode() == 3) return arena_alloc((int)gpu_index, size, S(stream));
  if (malloc_async_mode() == 1) {
    HX_CHECK(hipMalloc(&p, size));
    return p;
  }
  HX_CHECK(hipMallocAsync(&p, size, S(stream)));
  if (malloc_async_mode() == 0 && p != nullptr) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pool_allocations[p] = size;
  }
  return p;
}
bool cuda_check_valid_malloc(uint64_t size, uint32_t gpu_index) {
  set_device(gpu_index);
  size_t free_mem = 0, total_mem = 0;
  HX_CHECK(hipMemGetInfo(&free_mem, &total_mem));
  return size <= free_mem;
}
uint64_t cuda_device_total_memory(uint32_t gpu_index) {
  set_device(gpu_index);
  size_t free_mem = 0, total_mem = 0;
  HX_CHECK(hipMemGetInfo(&free_mem, &total_mem));
  return total_mem;
}
void cuda_memcpy_async_to_gpu(void *dest, const void *src, uint64_t size, void *stream, uint32_t gpu_index) {
  if (size == 0) return;
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(dest != nullptr && src != nullptr, "memcpy to gpu: null pointer");
  device_range_changes((int)gpu_index, dest, size);  // a keyswitch key may be rewritten in place
  HX_CHECK(hipMemcpyAsync(dest, src, size, hipMemcpyHostToDevice, S(stream)));
}
void cuda_memcpy_async_gpu_to_gpu(void *dest, void const *src, uint64_t size, void *stream, uint32_t gpu_index) {
  if (size == 0) return;
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(dest != nullptr && src != nullptr, "memcpy gpu to gpu: null pointer");
  device_range_changes((int)gpu_index, dest, size);
  HX_CHECK(hipMemcpyAsync(dest, src, size, hipMemcpyDeviceToDevice, S(stream)));
}
void cuda_memcpy_gpu_to_gpu(void *dest, void const *src, uint64_t size, uint32_t gpu_index) {
  if (size == 0) return;
  set_device(gpu_index);
  device_range_changes((int)gpu_index, dest, size);
  HX_CHECK(hipMemcpy(dest, src, size, hipMemcpyDeviceToDevice));
}
void cuda_memcpy_async_to_cpu(void *dest, const void *src, uint64_t size, void *stream, uint32_t gpu_index) {
  if (size == 0) return;
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(dest != nullptr && src != nullptr, "memcpy to cpu: null pointer");
  HX_CHECK(hipMemcpyAsync(dest, src, size, hipMemcpyDeviceToHost, S(stream)));
}
void cuda_memset_async(void *dest, uint64_t val, uint64_t size, void *stream, uint32_t gpu_index) {
  if (size == 0) return;
  set_device(gpu_index);
  device_range_changes((int)gpu_index, dest, size);
  HX_CHECK(hipMemsetAsync(dest, (int)val, size, S(stream)));
}
int cuda_get_number_of_gpus(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int cuda_get_number_of_sms(void) {
  hipDeviceProp_t prop;
  HX_CHECK(hipGetDeviceProperties(&prop, 0));
  return prop.multiProcessorCount;  // compute units
}
void cuda_synchronize_device(uint32_t gpu_index) {
  set_device(gpu_index);
  HX_CHECK(hipDeviceSynchronize());
}
void cuda_drop(void *ptr, uint32_t gpu_index) {
  set_device(gpu_index);
  size_t pool_bytes = 0;
  bool from_pool = false;
  {
    // an arena block goes back to the arena in stream order: no runtime call, no synchronisation (arena.hip)
    size_t user_bytes = 0;
    if (ptr != nullptr && arena_free((int)gpu_index, ptr, &user_bytes)) {
      device_range_changes((int)gpu_index, ptr, user_bytes);  // a keyswitch key in it takes its cached layout along
      return;
    }
  }
  if (ptr != nullptr) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    auto it = g_pool_allocations.find(ptr);
    if (it != g_pool_allocations.end()) {
      from_pool = true;
      pool_bytes = it->second;
      g_pool_allocations.erase(it);
    }
  }
  if (from_pool) {
    device_range_changes((int)gpu_index, ptr, pool_bytes);  // the allocation's own bytes, not the pool's block
    HX_CHECK(hipFreeAsync(ptr, nullptr));
    HX_CHECK(hipDeviceSynchronize());  // cudaFree synchronises; the memory is back in the pool for every stream
    return;
  }
  if (ptr != nullptr && ksm_cache_entries() != 0) {
    // a keyswitch key inside this allocation takes its cached matrix-core layout with it
    void *base = ptr;
    size_t bytes = 1;
    if (hipMemGetAddressRange(&base, &bytes, ptr) != hipSuccess) base = ptr, bytes = 1;
    device_range_changes((int)gpu_index, base, bytes);
  }
  HX_CHECK(hipFree(ptr));
}

// =========================================================================== classic PBS
static void convert_bsk_common(bool ntt, void *stream, uint32_t gpu_index, void *dest, void const *src,
                               uint32_t input_lwe_dim, uint32_t glwe_dim, uint32_t level_count,
                               uint32_t polynomial_size) {
  set_device(gpu_index);
  check_pow2_poly(polynomial_size, ntt ? 4096 : 16384);
  HX_PANIC_IF_FALSE(dest != nullptr && src != nullptr, "bootstrap key conversion: null pointer");
  const size_t polys = (size_t)input_lwe_dim * level_count * (glwe_dim + 1) * (glwe_dim + 1);
  const size_t bytes = polys * polynomial_size * sizeof(uint64_t);
  // stage the standard-domain key on the device, transform polynomial by polynomial
  void *tmp = device_alloc_sync(bytes);
  HX_CHECK(hipMemcpyAsync(tmp, src, bytes, hipMemcpyHostToDevice, S(stream)));
  if (ntt) {
    const NttTables tb = get_ntt_tables(gpu_index, S(stream), polynomial_size);
    launch_bsk_to_ntt(S(stream), polynomial_size, (const uint64_t *)tmp, dest, polys, tb);
  } else {
    const FftTables tb = get_fft_tables(gpu_index, S(stream), polynomial_size);
    launch_bsk_to_fourier(S(stream), polynomial_size, glwe_dim, (const uint64_t *)tmp, dest, polys, tb);
  }
  // the staging buffer must outlive the kernel: release it once the stream reaches here
  HX_CHECK(hipStreamSynchronize(S(stream)));
  device_free_sync(tmp);
}

void cuda_convert_lwe_programmable_bootstrap_key_64_async(void *stream, uint32_t gpu_index, void *dest,
                                                          void const *src, uint32_t input_lwe_dim,
                                                          uint32_t glwe_dim, uint32_t level_count,
                                                          uint32_t polynomial_size) {
  convert_bsk_common(false, stream, gpu_index, dest, src, input_lwe_dim, glwe_dim, level_count, polynomial_size);
}
void hip_convert_lwe_programmable_bootstrap_key_ntt64_async(void *stream, uint32_t gpu_index, void *dest,
                                                            void const *src, uint32_t input_lwe_dim,
                                                            uint32_t glwe_dim, uint32_t level_count,
                                                            uint32_t polynomial_size) {
  convert_bsk_common(true, stream, gpu_index, dest, src, input_lwe_dim, glwe_dim, level_count, polynomial_size);
}

uint64_t scratch_cuda_programmable_bootstrap_64_async(void *stream, uint32_t gpu_index, int8_t **buffer,
                                                      uint32_t lwe_dimension, uint32_t glwe_dimension,
                                                      uint32_t polynomial_size, uint32_t level_count,
                                                      uint32_t input_lwe_ciphertext_count, bool allocate_gpu_memory,
                                                      enum PBS_MS_REDUCTION_T noise_reduction_type) {
  set_device(gpu_index);
  check_pow2_poly(polynomial_size, 16384);
  HX_PANIC_IF_FALSE(glwe_dimension >= 1 && glwe_dimension <= 3, "glwe_dimension %u not supported", glwe_dimension);
  HX_PANIC_IF_FALSE(polynomial_size <= 4096 || glwe_dimension == 1,
                    "polynomial_size %u is supported with glwe_dimension 1 only", polynomial_size);
  auto *b = new PbsBuffer();
  b->magic = kPbsMagic;
  b->lwe_dimension = lwe_dimension;
  b->glwe_dimension = glwe_dimension;
  b->polynomial_size = polynomial_size;
  b->level_count = level_count;
  b->max_samples = input_lwe_ciphertext_count;
  b->ms_type = (uint32_t)noise_reduction_type;
  b->gpu_memory_allocated = allocate_gpu_memory;
  if (allocate_gpu_memory) {
    // constant tables are built here (not in the launch) so the launch stays capture-safe
    b->fft = get_fft_tables(gpu_index, S(stream), polynomial_size);
    if (polynomial_size <= 4096) {
      b->ntt = get_ntt_tables(gpu_index, S(stream), polynomial_size);
    }
  }
  // bytes of device scratch: none up to N = 4096 (the accumulator never leaves the chip), one accumulator per
  // sample beyond
  const uint64_t bytes = polynomial_size <= 4096 ? 0
                                                 : (uint64_t)input_lwe_ciphertext_count * (glwe_dimension + 1) *
                                                       polynomial_size * sizeof(uint64_t);
  if (allocate_gpu_memory && bytes) b->acc_scratch = (uint64_t *)scratch_alloc(bytes);
  *buffer = reinterpret_cast<int8_t *>(b);
  return bytes;
}

// Scratch of the one-call KS -> PBS entry point (hip_keyswitch_programmable_bootstrap_64_async): the classic PBS
// scratch plus room for the keyswitched list and its trivial indexes (filled by a kernel on `stream`: nothing
// blocks).  A caller that only bootstraps takes the plain scratch above and pays for neither.
uint64_t hip_scratch_keyswitch_programmable_bootstrap_64_async(void *stream, uint32_t gpu_index, int8_t **buffer,
                                                               uint32_t lwe_dimension, uint32_t glwe_dimension,
                                                               uint32_t polynomial_size, uint32_t level_count,
                                                               uint32_t input_lwe_ciphertext_count,
                                                               bool allocate_gpu_memory,
                                                               enum PBS_MS_REDUCTION_T noise_reduction_type) {
  const uint64_t bytes = scratch_cuda_programmable_bootstrap_64_async(
      stream, gpu_index, buffer, lwe_dimension, glwe_dimension, polynomial_size, level_count,
      input_lwe_ciphertext_count, allocate_gpu_memory, noise_reduction_type);
  auto *b = reinterpret_cast<PbsBuffer *>(*buffer);
  const uint64_t ks_bytes = (uint64_t)input_lwe_ciphertext_count * (lwe_dimension + 1) * sizeof(uint64_t);
  const uint64_t idx_bytes = (uint64_t)input_lwe_ciphertext_count * sizeof(uint64_t);
  if (allocate_gpu_memory && ks_bytes) {
    b->ks_out = (uint64_t *)scratch_alloc(ks_bytes);
    b->trivial = (uint64_t *)scratch_alloc(idx_bytes);
    launch_iota_u64(S(stream), b->trivial, input_lwe_ciphertext_count);
  }
  return bytes + ks_bytes + idx_bytes;
}

static PbsBuffer *checked_buffer(int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
                                 uint32_t polynomial_size, uint32_t level_count, uint32_t num_samples) {
  auto *b = reinterpret_cast<PbsBuffer *>(buffer);
  HX_PANIC_IF_FALSE(b != nullptr && b->magic == kPbsMagic, "PBS buffer was not created by scratch_cuda_programmable_bootstrap_64_async");
  HX_PANIC_IF_FALSE(b->gpu_memory_allocated, "PBS buffer was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(b->lwe_dimension == lwe_dimension && b->glwe_dimension == glwe_dimension &&
                        b->polynomial_size == polynomial_size && b->level_count == level_count,
                    "PBS buffer parameters do not match the call");
  HX_PANIC_IF_FALSE(num_samples <= b->max_samples, "num_samples %u exceeds the scratch capacity %u", num_samples,
                    b->max_samples);
  return b;
}

// kernel choice of the classic f64 PBS; returns the id hip_backend_last_pbs_kernel reports
static uint32_t launch_classic_pbs(hipStream_t st, PbsArgs &a, PbsBuffer *b, uint32_t glwe_dimension,
                                   uint32_t polynomial_size) {
  const uint32_t level_count = a.level, base_log = a.base_log, num_samples = a.num_samples;
  a.acc_scratch = b->acc_scratch;
  const uint32_t choice = g_fft_kernel_choice.load();
  const bool wave_ok = pbs_fft_wave_supported(polynomial_size, glwe_dimension, level_count) && base_log <= 31;
  const bool wave3_ok = pbs_fft_wave3_supported(polynomial_size, glwe_dimension, level_count);
  const bool block_ok = pbs_fft_block_supported(polynomial_size, glwe_dimension, level_count);
  if (choice == 2) HX_PANIC_IF_FALSE(wave_ok || wave3_ok, "throughput kernel requested for an unsupported parameter set");
  if (choice == 3 || choice == 4) HX_PANIC_IF_FALSE(block_ok, "latency kernel requested for an unsupported parameter set");
  uint32_t id;
  // automatic choice: up to one LWE per CU the latency kernel finishes first; beyond, the throughput kernel
  if (choice == 3 || choice == 4 || (choice == 0 && block_ok && num_samples <= kLatencyKernelMaxBatch)) {
    // 3 / automatic: one polynomial per half workgroup (512 threads); 4: dual-stream variant (256 threads),
    // measured slower (4.7 vs 4.2 ms), kept for comparison
    launch_pbs_fft_block(st, a, b->fft, choice == 4 ? 0 : 1);
    id = choice == 4 ? 8 : 7;
  } else if (wave_ok && (choice == 0 || choice == 2)) {
    launch_pbs_fft_wave(st, a, b->fft);
    id = 2;
  } else if (wave3_ok && (choice == 0 || choice == 2)) {  // N = 1024: one wave per polynomial, 512-point transforms
    launch_pbs_fft_wave3(st, glwe_dimension, a, b->fft);
    id = 9;
  } else {
    launch_pbs_fft_generic(st, polynomial_size, glwe_dimension, a, b->fft);
    id = 1;
  }
  g_last_pbs_kernel.store(id);
  return id;
}

void cuda_programmable_bootstrap_64_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                          void const *lwe_output_indexes, void const *lut_vector,
                                          void const *lut_vector_indexes, void const *lwe_array_in,
                                          void const *lwe_input_indexes, void const *bootstrapping_key,
                                          int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
                                          uint32_t polynomial_size, uint32_t base_log, uint32_t level_count,
                                          uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride) {
  set_device(gpu_index);
  PbsBuffer *b = checked_buffer(buffer, lwe_dimension, glwe_dimension, polynomial_size, level_count, num_samples);
  HX_PANIC_IF_FALSE(base_log >= 1 && base_log * level_count < 64, "invalid decomposition (base_log=%u, level=%u)",
                    base_log, level_count);
  HX_PANIC_IF_FALSE(num_many_lut >= 1, "num_many_lut must be >= 1");
  if (num_samples == 0) return;
  PbsArgs a = make_args(lwe_array_out, lwe_output_indexes, lut_vector, lut_vector_indexes, lwe_array_in,
                        lwe_input_indexes, bootstrapping_key, lwe_dimension, base_log, level_count, num_samples,
                        num_many_lut, lut_stride, b->ms_type);
  b->emitted.valid = false;  // whatever a chained call left describes an array this call may overwrite
  launch_classic_pbs(S(stream), a, b, glwe_dimension, polynomial_size);
}

// The shortint atomic pattern (tfhe/src/shortint/atomic_pattern/standard.rs:162-199; GPU: integer.cuh:869-990) in
// one call: keyswitch big -> small key into the scratch, then the PBS on the result — two launches on `stream`,
// nothing allocated, no host synchronisation, the key layout of the matrix-core keyswitch served from its cache.
// lwe_array_in holds ciphertexts under the BIG key (dimension k*N); the PBS reads the keyswitched list trivially.
void hip_keyswitch_programmable_bootstrap_64_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                                   void const *lwe_output_indexes, void const *lut_vector,
                                                   void const *lut_vector_indexes, void const *lwe_array_in,
                                                   void const *lwe_input_indexes, void const *ksk,
                                                   void const *bootstrapping_key, int8_t *buffer,
                                                   uint32_t lwe_dimension, uint32_t glwe_dimension,
                                                   uint32_t polynomial_size, uint32_t ks_base_log, uint32_t ks_level,
                                                   uint32_t base_log, uint32_t level_count, uint32_t num_samples,
                                                   uint32_t num_many_lut, uint32_t lut_stride) {
  set_device(gpu_index);
  PbsBuffer *b = checked_buffer(buffer, lwe_dimension, glwe_dimension, polynomial_size, level_count, num_samples);
  HX_PANIC_IF_FALSE(b->ks_out != nullptr || num_samples == 0,
                    "PBS buffer has no keyswitch scratch: create it with hip_scratch_keyswitch_programmable_bootstrap_64_async");
  if (num_samples == 0) return;
  // the keyswitch writes block s at position s of the scratch list; the PBS reads that list trivially
  launch_keyswitch(S(stream), b->ks_out, b->trivial, (const uint64_t *)lwe_array_in,
                   (const uint64_t *)lwe_input_indexes, (const uint64_t *)ksk, glwe_dimension * polynomial_size,
                   lwe_dimension, ks_base_log, ks_level, num_samples);
  cuda_programmable_bootstrap_64_async(stream, gpu_index, lwe_array_out, lwe_output_indexes, lut_vector,
                                       lut_vector_indexes, b->ks_out, b->trivial, bootstrapping_key,
                                       buffer, lwe_dimension, glwe_dimension, polynomial_size, base_log, level_count,
                                       num_samples, num_many_lut, lut_stride);
}

// The same KS -> PBS call for CHAINS of rounds in which a round's keyswitch reads exactly what the previous round's
// bootstrap wrote (shortint apply-lookup-table chains): the sample extraction of the bootstrap is fused with the
// digit pass of the NEXT keyswitch.
//   HIP_KSPBS_EMIT_DIGITS        the bootstrap also writes, for every output ciphertext, the shifted keyswitch digits of
//                                its mask (decomposition ks_base_log / ks_level of THIS call) as int8 A operands of the
//                                keyswitch GEMM plus their per-sample sums, into the scratch;
//   HIP_KSPBS_INPUT_FROM_PREVIOUS the caller states that lwe_array_in / lwe_input_indexes are the lwe_array_out /
//                                lwe_output_indexes of the previous call on this scratch and that nothing wrote to them
//                                since: the keyswitch then starts from the emitted operands (no digit pass).  The
//                                library checks pointers, count and decomposition against what it recorded and falls
//                                back to the digit pass when they differ or nothing was emitted.
// Emission is done by the N = 2048, k = 1 throughput kernel (more than 256 LWEs), for keyswitch level counts padded to
// 4 or 8 and base_log * level <= 30; otherwise the flags change nothing.  Same bits with and without the flags.
void hip_keyswitch_programmable_bootstrap_chain_64_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                                         void const *lwe_output_indexes, void const *lut_vector,
                                                         void const *lut_vector_indexes, void const *lwe_array_in,
                                                         void const *lwe_input_indexes, void const *ksk,
                                                         void const *bootstrapping_key, int8_t *buffer,
                                                         uint32_t lwe_dimension, uint32_t glwe_dimension,
                                                         uint32_t polynomial_size, uint32_t ks_base_log,
                                                         uint32_t ks_level, uint32_t base_log, uint32_t level_count,
                                                         uint32_t num_samples, uint32_t num_many_lut,
                                                         uint32_t lut_stride, uint32_t flags) {
  set_device(gpu_index);
  PbsBuffer *b = checked_buffer(buffer, lwe_dimension, glwe_dimension, polynomial_size, level_count, num_samples);
  HX_PANIC_IF_FALSE(b->ks_out != nullptr || num_samples == 0,
                    "PBS buffer has no keyswitch scratch: create it with hip_scratch_keyswitch_programmable_bootstrap_64_async");
  if (num_samples == 0) return;
  const uint32_t n_big = glwe_dimension * polynomial_size;
  uint32_t level_pad = 0, steps = 0;
  const bool emittable = keyswitch_digits_emittable(n_big, ks_base_log, ks_level, &level_pad, &steps);
  // ---- keyswitch: from the operands the previous bootstrap left, when they are what this call reads
  KsDigits ready{b->emit_a, b->emit_suma, b->emitted.steps, b->emitted.base_log, b->emitted.level};
  const bool use_ready = (flags & HIP_KSPBS_INPUT_FROM_PREVIOUS) && b->emitted.valid && b->emitted.array == lwe_array_in &&
                         b->emitted.indexes == lwe_input_indexes && b->emitted.count == num_samples &&
                         b->emitted.base_log == ks_base_log && b->emitted.level == ks_level && emittable;
  launch_keyswitch(S(stream), b->ks_out, b->trivial, (const uint64_t *)lwe_array_in,
                   (const uint64_t *)lwe_input_indexes, (const uint64_t *)ksk, n_big, lwe_dimension, ks_base_log,
                   ks_level, num_samples, use_ready ? &ready : nullptr);
  b->emitted.valid = false;  // the operands described the INPUT of this call; its output replaces them below or not
  // ---- bootstrap, with the emission when the throughput kernel takes the launch
  HX_PANIC_IF_FALSE(base_log >= 1 && base_log * level_count < 64, "invalid decomposition (base_log=%u, level=%u)",
                    base_log, level_count);
  HX_PANIC_IF_FALSE(num_many_lut >= 1, "num_many_lut must be >= 1");
  PbsArgs a = make_args(lwe_array_out, lwe_output_indexes, lut_vector, lut_vector_indexes, b->ks_out, b->trivial,
                        bootstrapping_key, lwe_dimension, base_log, level_count, num_samples, num_many_lut, lut_stride,
                        b->ms_type);
  // The operands only pay off where the NEXT keyswitch takes the GEMM path (keyswitch_mfma consumes them from
  // g_keyswitch_gemm_min LWEs on): below that, the flag changes nothing.
  if ((flags & HIP_KSPBS_EMIT_DIGITS) && emittable && num_many_lut == 1 && polynomial_size == 2048 && glwe_dimension == 1 &&
      num_samples >= g_keyswitch_gemm_min.load()) {
    const size_t a_bytes = (size_t)((b->max_samples + 31) / 32) * steps * 1024;
    const size_t need = a_bytes + (size_t)((b->max_samples + 31) / 32) * 32 * sizeof(int32_t);
    if (b->emit_bytes < need && !stream_is_capturing(S(stream))) {
      // first use (or another decomposition): an allocation, hence never under stream capture; a buffer that is
      // replaced is kept until the scratch is cleaned up — a graph captured earlier may hold its address
      if (b->emit_a) b->emit_retired.push_back(b->emit_a);
      b->emit_a = (int8_t *)scratch_alloc(need);
      b->emit_bytes = need;
    }
    if (b->emit_bytes >= need) {  // (a capture that found no buffer: no emission, the next keyswitch runs its digit pass)
      b->emit_suma = (int32_t *)(b->emit_a + a_bytes);
      a.emit_a = b->emit_a;
      a.emit_suma = b->emit_suma;
      a.emit_base_log = ks_base_log;
      a.emit_level = ks_level;
      a.emit_level_pad = level_pad;
      a.emit_steps = steps;
      b->emitted.valid = true;
      b->emitted.array = lwe_array_out;
      b->emitted.indexes = lwe_output_indexes;
      b->emitted.count = num_samples;
      b->emitted.steps = steps;
      b->emitted.base_log = ks_base_log;
      b->emitted.level = ks_level;
    }
  }
  if (launch_classic_pbs(S(stream), a, b, glwe_dimension, polynomial_size) != 2)
    b->emitted.valid = false;  // only the throughput kernel emits: a launch another kernel took left nothing usable
}

void hip_programmable_bootstrap_ntt64_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                            void const *lwe_output_indexes, void const *lut_vector,
                                            void const *lut_vector_indexes, void const *lwe_array_in,
                                            void const *lwe_input_indexes, void const *bootstrapping_key,
                                            int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
                                            uint32_t polynomial_size, uint32_t base_log, uint32_t level_count,
                                            uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride) {
  set_device(gpu_index);
  PbsBuffer *b = checked_buffer(buffer, lwe_dimension, glwe_dimension, polynomial_size, level_count, num_samples);
  HX_PANIC_IF_FALSE(base_log >= 1 && base_log * level_count < 64, "invalid decomposition (base_log=%u, level=%u)",
                    base_log, level_count);
  if (num_samples == 0) return;
  const PbsArgs a = make_args(lwe_array_out, lwe_output_indexes, lut_vector, lut_vector_indexes, lwe_array_in,
                              lwe_input_indexes, bootstrapping_key, lwe_dimension, base_log, level_count,
                              num_samples, num_many_lut, lut_stride, b->ms_type);
  launch_pbs_ntt_generic(S(stream), polynomial_size, glwe_dimension, a, b->ntt);
  g_last_pbs_kernel.store(3);
}

// ---- the exact engine in its split-key f64 form (pbs_fft_wave.hip, LIMBS mode): the same function as
// hip_programmable_bootstrap_ntt64_async — bit for bit — for N = 2048, k = 1, one level, base_log 22 / 23, on the
// throughput kernel's machinery.  The key takes NTT_SPLIT_LIMBS times the bytes of the classic Fourier key.
bool hip_programmable_bootstrap_ntt64_split_supported(uint32_t glwe_dimension, uint32_t polynomial_size,
                                                      uint32_t level_count, uint32_t base_log) {
  return pbs_ntt_split_supported(polynomial_size, glwe_dimension, level_count, base_log);
}
void hip_convert_lwe_programmable_bootstrap_key_ntt64_split_async(void *stream, uint32_t gpu_index, void *dest,
                                                                  void const *src, uint32_t input_lwe_dim,
                                                                  uint32_t glwe_dim, uint32_t level_count,
                                                                  uint32_t polynomial_size) {
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(dest != nullptr && src != nullptr, "bootstrap key conversion: null pointer");
  HX_PANIC_IF_FALSE(polynomial_size == 2048 && glwe_dim == 1 && level_count == 1,
                    "split-key exact engine: parameter set not supported (N = 2048, k = 1, one level)");
  const size_t polys = (size_t)input_lwe_dim * level_count * (glwe_dim + 1) * (glwe_dim + 1);
  const size_t bytes = polys * polynomial_size * sizeof(uint64_t);
  void *tmp = device_alloc_sync(bytes);
  HX_CHECK(hipMemcpyAsync(tmp, src, bytes, hipMemcpyHostToDevice, S(stream)));
  device_range_changes((int)gpu_index, dest, bytes * NTT_SPLIT_LIMBS);
  launch_bsk_to_split(S(stream), polynomial_size, (const uint64_t *)tmp, dest, polys,
                      get_fft_tables(gpu_index, S(stream), polynomial_size));
  // the key's NTT-domain twin (as many bytes as the standard key): what the integer kernel recomputes a flagged
  // ciphertext with (hip_programmable_bootstrap_ntt64_split_async)
  void *twin = device_alloc_sync(bytes);
  launch_bsk_to_ntt(S(stream), polynomial_size, (const uint64_t *)tmp, twin, polys,
                    get_ntt_tables(gpu_index, S(stream), polynomial_size));
  HX_CHECK(hipStreamSynchronize(S(stream)));  // the staging buffer must outlive the kernels
  device_free_sync(tmp);
  std::lock_guard<std::mutex> lock(g_twin_mutex);
  g_split_twins.push_back(SplitTwin{(int)gpu_index, dest, bytes * NTT_SPLIT_LIMBS, twin});
}
void hip_programmable_bootstrap_ntt64_split_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                                  void const *lwe_output_indexes, void const *lut_vector,
                                                  void const *lut_vector_indexes, void const *lwe_array_in,
                                                  void const *lwe_input_indexes, void const *bootstrapping_key,
                                                  int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
                                                  uint32_t polynomial_size, uint32_t base_log, uint32_t level_count,
                                                  uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride) {
  set_device(gpu_index);
  PbsBuffer *b = checked_buffer(buffer, lwe_dimension, glwe_dimension, polynomial_size, level_count, num_samples);
  HX_PANIC_IF_FALSE(pbs_ntt_split_supported(polynomial_size, glwe_dimension, level_count, base_log),
                    "split-key exact engine: parameter set not supported (N=%u, k=%u, level=%u, base_log=%u)",
                    polynomial_size, glwe_dimension, level_count, base_log);
  if (num_samples == 0) return;
  if (b->split_acc == nullptr) {
    // first use of this engine with this scratch: the accumulators' home (an allocation — not under stream capture;
    // a capture must be preceded by one plain launch, like the keyswitch's first use of a key)
    HX_PANIC_IF_FALSE(!stream_is_capturing(S(stream)),
                      "split-key exact engine: the first launch on a scratch allocates and cannot be captured; run it once before the capture");
    b->split_acc = (uint64_t *)scratch_alloc((size_t)b->max_samples * (glwe_dimension + 1) * polynomial_size * sizeof(uint64_t));
    // word 0: the round-off flag; words 64 .. 319: the per-XCD progress counters of the paced loop (PbsArgs::pace)
    b->split_flag = (uint32_t *)scratch_alloc((64 + 8 * 32) * sizeof(uint32_t));
    HX_CHECK(hipMemsetAsync(b->split_flag, 0, (64 + 8 * 32) * sizeof(uint32_t), S(stream)));
    b->split_bad = (uint32_t *)scratch_alloc((size_t)b->max_samples * sizeof(uint32_t));
  }
  PbsArgs a = make_args(lwe_array_out, lwe_output_indexes, lut_vector, lut_vector_indexes, lwe_array_in,
                        lwe_input_indexes, bootstrapping_key, lwe_dimension, base_log, level_count, num_samples,
                        num_many_lut, lut_stride, b->ms_type);
  a.acc_scratch = b->split_acc;
  a.roundoff_flag = b->split_flag;
  a.pace = b->split_flag + 64;
  // A ciphertext whose f64 limb products were not within 1/4 of integers (the engine's round-off check: statistical bound,
  // adversarial data can reach it — tests/test_split_engine_worst_case.py) is RECOMPUTED by the integer Goldilocks kernel
  // on the same stream, with the key's NTT-domain twin: per sample a flag word, the second launch's workgroups return at
  // once where it is 0 (a few microseconds for the launch when nothing was flagged).  Both compute ntt64_bnf_pbs.rs:208-280,
  // so the outputs are the exact ones whatever the data; the status call counts the recomputed ciphertexts.
  const void *twin = split_twin_of((int)gpu_index, bootstrapping_key);
  if (twin != nullptr) {
    HX_CHECK(hipMemsetAsync(b->split_bad, 0, (size_t)num_samples * sizeof(uint32_t), S(stream)));
    a.bad_samples = b->split_bad;
  } else {
    b->split_unrecovered = true;  // a key this library did not convert (copied in by the caller): the flag stays fatal
  }
  launch_pbs_ntt_split_wave(S(stream), a, b->fft);
  if (twin != nullptr) {
    PbsArgs r = a;
    r.bsk = twin;
    r.acc_scratch = nullptr;
    r.roundoff_flag = nullptr;
    r.bad_samples = nullptr;
    r.pace = nullptr;
    r.only_flagged = b->split_bad;
    r.recomputed = b->split_flag + 1;
    launch_pbs_ntt_generic(S(stream), polynomial_size, glwe_dimension, r, b->ntt);
  }
  g_last_pbs_kernel.store(13);
}
// The number of ciphertexts the integer kernel had to recompute behind the split-key launches on this scratch since the last
// call (their f64 limb products were further than 1/4 from integers: 0 on every real parameter set's data); the outputs are
// the exact ones either way.  Synchronises the stream and clears the count.  Panics if the flag went up in a launch whose
// key had no NTT-domain twin (a split key the caller copied instead of converting): those outputs cannot be trusted.
uint32_t hip_programmable_bootstrap_ntt64_split_roundoff_status(void *stream, uint32_t gpu_index, int8_t *buffer) {
  set_device(gpu_index);
  PbsBuffer *b = reinterpret_cast<PbsBuffer *>(buffer);
  HX_PANIC_IF_FALSE(b != nullptr && b->magic == kPbsMagic, "roundoff_status: foreign scratch pointer");
  if (b->split_flag == nullptr) return 0;
  uint32_t v[2] = {0, 0};
  HX_CHECK(hipMemcpyAsync(v, b->split_flag, sizeof(v), hipMemcpyDeviceToHost, S(stream)));
  HX_CHECK(hipStreamSynchronize(S(stream)));
  HX_PANIC_IF_FALSE(v[0] == 0 || !b->split_unrecovered,
                    "split-key exact engine: an f64 limb product was further than 1/4 from an integer in a launch whose key has no "
                    "NTT-domain twin (convert the key with hip_convert_lwe_programmable_bootstrap_key_ntt64_split_async) — its "
                    "outputs are not the exact ones");
  if (v[0] != 0 || v[1] != 0) HX_CHECK(hipMemsetAsync(b->split_flag, 0, sizeof(v), S(stream)));
  return v[1];
}

// ---- reference-order f64 engine (pbs_ref64.hip): the key in tfhe-fft's dif4 transform order
void hip_convert_lwe_programmable_bootstrap_key_ref64_async(void *stream, uint32_t gpu_index, void *dest,
                                                            void const *src, uint32_t input_lwe_dim,
                                                            uint32_t glwe_dim, uint32_t level_count,
                                                            uint32_t polynomial_size) {
  set_device(gpu_index);
  check_pow2_poly(polynomial_size, 2048);
  HX_PANIC_IF_FALSE(dest != nullptr && src != nullptr, "bootstrap key conversion: null pointer");
  const size_t polys = (size_t)input_lwe_dim * level_count * (glwe_dim + 1) * (glwe_dim + 1);
  const size_t bytes = polys * polynomial_size * sizeof(uint64_t);
  void *tmp = device_alloc_sync(bytes);
  HX_CHECK(hipMemcpyAsync(tmp, src, bytes, hipMemcpyHostToDevice, S(stream)));
  launch_bsk_to_ref64(S(stream), polynomial_size, (const uint64_t *)tmp, dest, polys,
                      get_ref_tables(gpu_index, S(stream), polynomial_size));
  HX_CHECK(hipStreamSynchronize(S(stream)));
  device_free_sync(tmp);
}
void hip_programmable_bootstrap_ref64_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                            void const *lwe_output_indexes, void const *lut_vector,
                                            void const *lut_vector_indexes, void const *lwe_array_in,
                                            void const *lwe_input_indexes, void const *bootstrapping_key,
                                            int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
                                            uint32_t polynomial_size, uint32_t base_log, uint32_t level_count,
                                            uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride) {
  set_device(gpu_index);
  PbsBuffer *b = checked_buffer(buffer, lwe_dimension, glwe_dimension, polynomial_size, level_count, num_samples);
  HX_PANIC_IF_FALSE(base_log >= 1 && base_log * level_count < 64, "invalid decomposition (base_log=%u, level=%u)",
                    base_log, level_count);
  if (num_samples == 0) return;
  const PbsArgs a = make_args(lwe_array_out, lwe_output_indexes, lut_vector, lut_vector_indexes, lwe_array_in,
                              lwe_input_indexes, bootstrapping_key, lwe_dimension, base_log, level_count,
                              num_samples, num_many_lut, lut_stride, b->ms_type);
  launch_pbs_ref64(S(stream), polynomial_size, glwe_dimension, a, get_ref_tables(gpu_index, S(stream), polynomial_size));
  g_last_pbs_kernel.store(11);
}

void hip_convert_lwe_programmable_bootstrap_key_exact64_async(void *stream, uint32_t gpu_index, void *dest,
                                                              void const *src, uint32_t input_lwe_dim,
                                                              uint32_t glwe_dim, uint32_t level_count,
                                                              uint32_t polynomial_size) {
  // the exact engine consumes the key in the standard domain: plain upload
  set_device(gpu_index);
  const size_t bytes = (size_t)input_lwe_dim * level_count * (glwe_dim + 1) * (glwe_dim + 1) * polynomial_size * 8;
  HX_CHECK(hipMemcpyAsync(dest, src, bytes, hipMemcpyHostToDevice, S(stream)));
}

void hip_programmable_bootstrap_exact64_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                              void const *lwe_output_indexes, void const *lut_vector,
                                              void const *lut_vector_indexes, void const *lwe_array_in,
                                              void const *lwe_input_indexes, void const *bootstrapping_key,
                                              int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
                                              uint32_t polynomial_size, uint32_t base_log, uint32_t level_count,
                                              uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride) {
  set_device(gpu_index);
  PbsBuffer *b = checked_buffer(buffer, lwe_dimension, glwe_dimension, polynomial_size, level_count, num_samples);
  HX_PANIC_IF_FALSE(base_log >= 1 && base_log * level_count < 64, "invalid decomposition (base_log=%u, level=%u)",
                    base_log, level_count);
  if (num_samples == 0) return;
  const PbsArgs a = make_args(lwe_array_out, lwe_output_indexes, lut_vector, lut_vector_indexes, lwe_array_in,
                              lwe_input_indexes, bootstrapping_key, lwe_dimension, base_log, level_count,
                              num_samples, num_many_lut, lut_stride, b->ms_type);
  launch_pbs_exact_generic(S(stream), polynomial_size, glwe_dimension, a);
  g_last_pbs_kernel.store(5);
}

void cleanup_cuda_programmable_bootstrap_64(void *stream, uint32_t gpu_index, int8_t **pbs_buffer) {
  set_device(gpu_index);
  auto *b = reinterpret_cast<PbsBuffer *>(*pbs_buffer);
  HX_PANIC_IF_FALSE(b != nullptr && b->magic == kPbsMagic, "cleanup of a foreign PBS buffer");
  HX_CHECK(hipStreamSynchronize(S(stream)));  // cleanup_* synchronises (pbs_utilities.h:261-271)
  if (b->split_flag && b->split_unrecovered) {
    // a launch of the split-key exact engine ran without the NTT-domain twin of its key: its round-off flag is checked here
    // even if nobody polled it — a host that never asks must not keep untrustworthy "exact" outputs
    uint32_t v = 0;
    HX_CHECK(hipMemcpy(&v, b->split_flag, sizeof(uint32_t), hipMemcpyDeviceToHost));
    HX_PANIC_IF_FALSE(v == 0, "split-key exact engine: an f64 limb product was further than 1/4 from an integer in a launch on "
                              "this scratch whose key has no NTT-domain twin — its outputs are not the exact ones");
  }
  if (b->acc_scratch) scratch_free(b->acc_scratch);
  if (b->split_acc) scratch_free(b->split_acc);
  if (b->split_flag) scratch_free(b->split_flag);
  if (b->split_bad) scratch_free(b->split_bad);
  for (void *r : b->emit_retired) scratch_free(r);
  if (b->emit_a) scratch_free(b->emit_a);
  if (b->ks_out) scratch_free(b->ks_out);
  if (b->trivial) scratch_free(b->trivial);
  b->magic = 0;
  delete b;
  *pbs_buffer = nullptr;
}

// =========================================================================== multi-bit PBS
bool has_support_to_cuda_programmable_bootstrap_cg_multi_bit(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) {
  return false;  // no cooperative-groups variant: one persistent launch per group step is used instead
}

void cuda_convert_lwe_multi_bit_programmable_bootstrap_key_64_async(void *stream, uint32_t gpu_index, void *dest,
                                                                    void const *src, uint32_t input_lwe_dim,
                                                                    uint32_t glwe_dim, uint32_t level_count,
                                                                    uint32_t polynomial_size,
                                                                    uint32_t grouping_factor) {
  // The CPU reference keeps the multi-bit key in the Fourier domain (FourierLweMultiBitBootstrapKey,
  // cc/algorithms/lwe_multi_bit_bootstrap_key_conversion.rs); its GPU backend uploads the standard-domain key
  // unchanged (cuda/src/pbs/bootstrapping_key.cu:78-93).  Here the key is transformed once at conversion time —
  // same byte size (N u64 -> N/2 complex per polynomial), same nesting [group][subset][level][row][col] — so that
  // the keybundle of every group is a pointwise combine (multibit.hip).
  set_device(gpu_index);
  check_pow2_poly(polynomial_size, glwe_dim == 1 ? 16384 : glwe_dim == 2 ? 2048 : 1024);
  HX_PANIC_IF_FALSE(dest != nullptr && src != nullptr, "multi-bit bootstrap key conversion: null pointer");
  HX_PANIC_IF_FALSE(grouping_factor >= 1 && input_lwe_dim % grouping_factor == 0,
                    "input_lwe_dim %u not a multiple of grouping_factor %u", input_lwe_dim, grouping_factor);
  const size_t polys = (size_t)(input_lwe_dim / grouping_factor) * ((size_t)1 << grouping_factor) * level_count *
                       (glwe_dim + 1) * (glwe_dim + 1);
  const size_t bytes = polys * polynomial_size * sizeof(uint64_t);
  void *tmp = device_alloc_sync(bytes);
  HX_CHECK(hipMemcpyAsync(tmp, src, bytes, hipMemcpyHostToDevice, S(stream)));
  const FftTables tb = get_fft_tables(gpu_index, S(stream), polynomial_size);
  launch_bsk_to_fourier(S(stream), polynomial_size, glwe_dim, (const uint64_t *)tmp, dest, polys, tb);
  HX_CHECK(hipStreamSynchronize(S(stream)));  // the staging buffer must outlive the kernel
  device_free_sync(tmp);
}

uint64_t scratch_cuda_multi_bit_programmable_bootstrap_64_async(void *stream, uint32_t gpu_index,
                                                                int8_t **pbs_buffer, uint32_t glwe_dimension,
                                                                uint32_t polynomial_size, uint32_t level_count,
                                                                uint32_t input_lwe_ciphertext_count,
                                                                bool allocate_gpu_memory) {
  set_device(gpu_index);
  check_pow2_poly(polynomial_size, glwe_dimension == 1 ? 16384 : glwe_dimension == 2 ? 2048 : 1024);
  auto *b = new MultiBitBuffer();
  b->magic = kMbMagic;
  b->glwe_dimension = glwe_dimension;
  b->polynomial_size = polynomial_size;
  b->level_count = level_count;
  b->max_samples = input_lwe_ciphertext_count;
  b->gpu_memory_allocated = allocate_gpu_memory;
  b->acc = nullptr;
  const size_t k1 = glwe_dimension + 1;
  const size_t kb_per_sample = (size_t)level_count * k1 * k1 * (polynomial_size / 2) * sizeof(cplx);
  const size_t acc_per_sample = 2 * k1 * polynomial_size * sizeof(uint64_t);
  b->chunk = input_lwe_ciphertext_count ? input_lwe_ciphertext_count : 1;
  // the throughput kernels build every keybundle element in registers: no per-sample keybundle scratch
  const uint64_t bytes = (uint64_t)b->chunk * acc_per_sample;
  // latency path: up to kMultiBitLatencyMaxBatch ciphertexts; the keybundles of as many groups per pass as
  // kMultiBitLatencyBytes hold (the scratch is sized without knowing n or the grouping factor, like the
  // reference's lwe_chunk_size); allocated here, never inside the launch
  const uint32_t lat_cap = (polynomial_size == 2048 && glwe_dimension == 1) ? kMultiBitLatencyMaxBatch2048 : kMultiBitLatencyMaxBatch;
  b->lat_samples = b->chunk < lat_cap ? b->chunk : lat_cap;
  if (polynomial_size > 4096) b->lat_samples = 0;  // rings of 2^13, 2^14: the one-launch kernel only
  b->kb_per_sample = kb_per_sample;
  uint64_t slots = kMultiBitLatencyBytes / kb_per_sample;  // (ciphertext, group) keybundles held at once
  const uint64_t most = (uint64_t)b->lat_samples * 1024;  // never more than 1024 groups per pass
  slots = slots < b->lat_samples ? b->lat_samples : slots > most ? most : slots;  // at least one group each
  const uint64_t lat_bytes = b->lat_samples ? slots * kb_per_sample : 0;
  if (allocate_gpu_memory) {
    b->fft = get_fft_tables(gpu_index, S(stream), polynomial_size, true);
    b->acc = (uint64_t *)scratch_alloc((size_t)b->chunk * acc_per_sample);
    b->pace = (uint32_t *)scratch_alloc(8 * 32 * sizeof(uint32_t));
    if (lat_bytes) b->kb_lat = (cplx *)scratch_alloc(lat_bytes);
  }
  b->lat_bytes = lat_bytes;
  *pbs_buffer = reinterpret_cast<int8_t *>(b);
  return bytes + lat_bytes;
}

void cuda_multi_bit_programmable_bootstrap_64_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                                    void const *lwe_output_indexes, void const *lut_vector,
                                                    void const *lut_vector_indexes, void const *lwe_array_in,
                                                    void const *lwe_input_indexes, void const *bootstrapping_key,
                                                    int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension,
                                                    uint32_t polynomial_size, uint32_t grouping_factor,
                                                    uint32_t base_log, uint32_t level_count, uint32_t num_samples,
                                                    uint32_t num_many_lut, uint32_t lut_stride) {
  set_device(gpu_index);
  auto *b = reinterpret_cast<MultiBitBuffer *>(buffer);
  HX_PANIC_IF_FALSE(b != nullptr && b->magic == kMbMagic, "multi-bit PBS buffer was not created by its scratch function");
  HX_PANIC_IF_FALSE(b->gpu_memory_allocated, "multi-bit PBS buffer was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(b->glwe_dimension == glwe_dimension && b->polynomial_size == polynomial_size &&
                        b->level_count == level_count && num_samples <= b->max_samples,
                    "multi-bit PBS buffer parameters do not match the call");
  HX_PANIC_IF_FALSE(grouping_factor >= 1 && grouping_factor <= 4 && lwe_dimension % grouping_factor == 0,
                    "unsupported grouping_factor %u for lwe_dimension %u", grouping_factor, lwe_dimension);
  if (num_samples == 0) return;
  MultiBitArgs m;
  m.pbs = make_args(lwe_array_out, lwe_output_indexes, lut_vector, lut_vector_indexes, lwe_array_in,
                    lwe_input_indexes, bootstrapping_key, lwe_dimension, base_log, level_count, num_samples,
                    num_many_lut, lut_stride, 0);
  m.grouping_factor = grouping_factor;
  const uint32_t choice = g_fft_kernel_choice.load();
  const bool wave_ok = pbs_multi_bit_wave_supported(polynomial_size, glwe_dimension, level_count, base_log,
                                                    grouping_factor);
  if (choice == 2) HX_PANIC_IF_FALSE(wave_ok, "throughput kernel requested for an unsupported parameter set");
  if (choice == 5 || choice == 6)
    HX_PANIC_IF_FALSE(num_samples <= b->lat_samples, "multi-bit latency path: %u samples exceed %u", num_samples,
                      b->lat_samples);
  if (choice == 5 || choice == 6 || (choice == 0 && num_samples <= b->lat_samples)) {
    m.pbs.mb_generic_products = choice == 6;  // 6: the products on the generic kernels (comparison)
    // few ciphertexts: every (group, keybundle polynomial) gets its own workgroup, then the products run alone
    const uint64_t fit = b->lat_bytes / ((uint64_t)num_samples * b->kb_per_sample);  // groups per pass for this batch
    const uint32_t fit_groups = (uint32_t)(fit > 1024 ? 1024 : fit);
    uint32_t gc = g_multibit_latency_groups.load();
    gc = (gc == 0 || gc > fit_groups) ? fit_groups : gc;
    launch_pbs_multi_bit_latency(S(stream), polynomial_size, glwe_dimension, m, b->fft, b->kb_lat, gc, b->acc);
    g_last_pbs_kernel.store(10);
  } else if ((choice == 0 && wave_ok) || choice == 2 || ((choice == 7 || choice == 8) && wave_ok)) {
    m.pbs.mb_no_share = choice == 7;  // 7: every wave pair loads its own key (comparison)
    m.pbs.mb_no_octet = choice == 8;  // 8: at most quads of waves share the key loads (comparison)
    m.pbs.grouping = grouping_factor;
    m.pbs.pace = b->pace;
    launch_pbs_multi_bit_wave(S(stream), m.pbs, b->fft);
    g_last_pbs_kernel.store(6);
  } else {
    launch_pbs_multi_bit(S(stream), polynomial_size, glwe_dimension, m, b->fft, b->acc);
    g_last_pbs_kernel.store(4);
  }
}

void cleanup_cuda_multi_bit_programmable_bootstrap_64(void *stream, uint32_t gpu_index, int8_t **pbs_buffer) {
  set_device(gpu_index);
  auto *b = reinterpret_cast<MultiBitBuffer *>(*pbs_buffer);
  HX_PANIC_IF_FALSE(b != nullptr && b->magic == kMbMagic, "cleanup of a foreign multi-bit PBS buffer");
  HX_CHECK(hipStreamSynchronize(S(stream)));
  if (b->acc) scratch_free(b->acc);
  if (b->kb_lat) scratch_free(b->kb_lat);
  if (b->pace) scratch_free(b->pace);
  b->magic = 0;
  delete b;
  *pbs_buffer = nullptr;
}

// The reference's noise tests run the blind rotation on an input that the multi-bit switch has ALREADY been applied to
// (cuda/include/pbs/programmable_bootstrap_multibit.h:44-60, cuda/src/pbs/programmable_bootstrap_multibit.cu:650-760;
// bound by tfhe/src/core_crypto/gpu/ffi.rs:322-397): lwe_array_in = [ the input ciphertext, n + 1 words | the output of
// cuda_modulus_switch_multi_bit_64_async, (n / g) 2^g words ]; the keybundle takes its monomial degrees from the second
// part (programmable_bootstrap_multibit.cuh:85-107), the body and everything else are read where they always are.
// One ciphertext per call and N = 2048 only, as there.  scratch / cleanup are the standard ones under their own names.
uint64_t scratch_cuda_multi_bit_programmable_bootstrap_noise_tests_64_async(void *stream, uint32_t gpu_index,
                                                                            int8_t **pbs_buffer, uint32_t glwe_dimension,
                                                                            uint32_t polynomial_size, uint32_t level_count,
                                                                            uint32_t input_lwe_ciphertext_count,
                                                                            bool allocate_gpu_memory) {
  return scratch_cuda_multi_bit_programmable_bootstrap_64_async(stream, gpu_index, pbs_buffer, glwe_dimension,
                                                                polynomial_size, level_count, input_lwe_ciphertext_count,
                                                                allocate_gpu_memory);
}
void cleanup_cuda_multi_bit_programmable_bootstrap_noise_tests_64(void *stream, uint32_t gpu_index, int8_t **pbs_buffer) {
  cleanup_cuda_multi_bit_programmable_bootstrap_64(stream, gpu_index, pbs_buffer);
}
void cuda_multi_bit_programmable_bootstrap_noise_tests_64_async(
    void *stream, uint32_t gpu_index, void *lwe_array_out, void const *lwe_output_indexes, void const *lut_vector,
    void const *lut_vector_indexes, void const *lwe_array_in, void const *lwe_input_indexes, void const *bootstrapping_key,
    int8_t *buffer, uint32_t lwe_dimension, uint32_t glwe_dimension, uint32_t polynomial_size, uint32_t grouping_factor,
    uint32_t base_log, uint32_t level_count, uint32_t num_samples, uint32_t num_many_lut, uint32_t lut_stride) {
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(num_samples == 1, "(multi-bit PBS): num_samples (%u) should be 1", num_samples);
  HX_PANIC_IF_FALSE(base_log <= 64, "(multi-bit PBS): base log (%u) should be <= 64", base_log);
  HX_PANIC_IF_FALSE(polynomial_size == 2048, "(multi-bit PBS noise tests): only polynomial size 2048 is supported, got %u.",
                    polynomial_size);
  auto *b = reinterpret_cast<MultiBitBuffer *>(buffer);
  HX_PANIC_IF_FALSE(b != nullptr && b->magic == kMbMagic, "multi-bit PBS buffer was not created by its scratch function");
  HX_PANIC_IF_FALSE(b->gpu_memory_allocated, "multi-bit PBS buffer was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(b->glwe_dimension == glwe_dimension && b->polynomial_size == polynomial_size &&
                        b->level_count == level_count && num_samples <= b->max_samples,
                    "multi-bit PBS buffer parameters do not match the call");
  HX_PANIC_IF_FALSE(grouping_factor >= 1 && grouping_factor <= 4 && lwe_dimension % grouping_factor == 0,
                    "unsupported grouping_factor %u for lwe_dimension %u", grouping_factor, lwe_dimension);
  MultiBitArgs m;
  m.pbs = make_args(lwe_array_out, lwe_output_indexes, lut_vector, lut_vector_indexes, lwe_array_in, lwe_input_indexes,
                    bootstrapping_key, lwe_dimension, base_log, level_count, num_samples, num_many_lut, lut_stride, 0);
  m.grouping_factor = grouping_factor;
  m.pbs.mb_degrees = (const uint64_t *)lwe_array_in + (lwe_dimension + 1);
  // one ciphertext, read once: the one-launch kernel (the only one that takes its degrees from memory)
  launch_pbs_multi_bit(S(stream), polynomial_size, glwe_dimension, m, b->fft, b->acc);
  g_last_pbs_kernel.store(4);
}

// =========================================================================== keyswitch
void cuda_keyswitch_lwe_ciphertext_vector_64_64_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                                      void const *lwe_output_indexes, void const *lwe_array_in,
                                                      void const *lwe_input_indexes, void const *ksk,
                                                      uint32_t lwe_dimension_in, uint32_t lwe_dimension_out,
                                                      uint32_t base_log, uint32_t level_count,
                                                      uint32_t num_samples) {
  set_device(gpu_index);
  launch_keyswitch(S(stream), (uint64_t *)lwe_array_out, (const uint64_t *)lwe_output_indexes,
                   (const uint64_t *)lwe_array_in, (const uint64_t *)lwe_input_indexes, (const uint64_t *)ksk,
                   lwe_dimension_in, lwe_dimension_out, base_log, level_count, num_samples);
}
void cuda_keyswitch_gemm_64_64_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                     void const *lwe_output_indexes, void const *lwe_array_in,
                                     void const *lwe_input_indexes, void const *ksk, uint32_t lwe_dimension_in,
                                     uint32_t lwe_dimension_out, uint32_t base_log, uint32_t level_count,
                                     uint32_t num_samples, bool uses_trivial_indexes) {
  // one tiled kernel serves both entry points (it already reuses each key row across a tile
  // of samples); indexes are honoured whether trivial or not
  (void)uses_trivial_indexes;
  cuda_keyswitch_lwe_ciphertext_vector_64_64_async(stream, gpu_index, lwe_array_out, lwe_output_indexes,
                                                   lwe_array_in, lwe_input_indexes, ksk, lwe_dimension_in,
                                                   lwe_dimension_out, base_log, level_count, num_samples);
}
void cuda_keyswitch_lwe_ciphertext_vector_64_32_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                                      void const *lwe_output_indexes, void const *lwe_array_in,
                                                      void const *lwe_input_indexes, void const *ksk,
                                                      uint32_t lwe_dimension_in, uint32_t lwe_dimension_out,
                                                      uint32_t base_log, uint32_t level_count,
                                                      uint32_t num_samples) {
  set_device(gpu_index);
  launch_keyswitch_64_32(S(stream), (uint32_t *)lwe_array_out, (const uint64_t *)lwe_output_indexes,
                         (const uint64_t *)lwe_array_in, (const uint64_t *)lwe_input_indexes, (const uint32_t *)ksk,
                         lwe_dimension_in, lwe_dimension_out, base_log, level_count, num_samples);
}
void cuda_keyswitch_gemm_64_32_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                     void const *lwe_output_indexes, void const *lwe_array_in,
                                     void const *lwe_input_indexes, void const *ksk, uint32_t lwe_dimension_in,
                                     uint32_t lwe_dimension_out, uint32_t base_log, uint32_t level_count,
                                     uint32_t num_samples, bool uses_trivial_indexes) {
  (void)uses_trivial_indexes;  // as for 64_64: one tiled kernel, indexes honoured either way
  cuda_keyswitch_lwe_ciphertext_vector_64_32_async(stream, gpu_index, lwe_array_out, lwe_output_indexes,
                                                   lwe_array_in, lwe_input_indexes, ksk, lwe_dimension_in,
                                                   lwe_dimension_out, base_log, level_count, num_samples);
}
void cuda_closest_representable_64_async(void *stream, uint32_t gpu_index, void const *input, void *output,
                                         uint32_t base_log, uint32_t level_count) {
  set_device(gpu_index);
  launch_closest_representable(S(stream), (const uint64_t *)input, (uint64_t *)output, base_log, level_count);
}

// =========================================================================== ciphertext helpers
void cuda_convert_lwe_ciphertext_vector_to_gpu_64_async(void *stream, uint32_t gpu_index, void *dest,
                                                        void const *src, uint32_t number_of_cts,
                                                        uint32_t lwe_dimension) {
  cuda_memcpy_async_to_gpu(dest, src, (uint64_t)number_of_cts * (lwe_dimension + 1) * sizeof(uint64_t), stream,
                           gpu_index);
}
void cuda_convert_lwe_ciphertext_vector_to_cpu_64_async(void *stream, uint32_t gpu_index, void *dest,
                                                        void const *src, uint32_t number_of_cts,
                                                        uint32_t lwe_dimension) {
  cuda_memcpy_async_to_cpu(dest, src, (uint64_t)number_of_cts * (lwe_dimension + 1) * sizeof(uint64_t), stream,
                           gpu_index);
}
void cuda_glwe_sample_extract_64_async(void *stream, uint32_t gpu_index, void *lwe_array_out,
                                       void const *glwe_array_in, uint32_t const *nth_array, uint32_t num_nths,
                                       uint32_t num_lwes_to_extract_per_glwe, uint32_t num_lwes_stored_per_glwe,
                                       uint32_t glwe_dimension, uint32_t polynomial_size) {
  set_device(gpu_index);
  launch_sample_extract(S(stream), (uint64_t *)lwe_array_out, (const uint64_t *)glwe_array_in, nth_array, num_nths,
                        num_lwes_to_extract_per_glwe, num_lwes_stored_per_glwe, glwe_dimension, polynomial_size);
}
void cuda_modulus_switch_inplace_64_async(void *stream, uint32_t gpu_index, void *lwe_array_out, uint32_t size,
                                          uint32_t log_modulus) {
  set_device(gpu_index);
  launch_modulus_switch(S(stream), (uint64_t *)lwe_array_out, (const uint64_t *)lwe_array_out, size, log_modulus);
}
void cuda_modulus_switch_64_async(void *stream, uint32_t gpu_index, void *lwe_out, const void *lwe_in, uint32_t size,
                                  uint32_t log_modulus) {
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(lwe_out != lwe_in, "Output and input pointers must be different for out-of-place operations");
  launch_modulus_switch(S(stream), (uint64_t *)lwe_out, (const uint64_t *)lwe_in, size, log_modulus);
}
void cuda_centered_modulus_switch_64_async(void *stream, uint32_t gpu_index, void *lwe_out, const void *lwe_in,
                                           uint32_t lwe_dimension, uint32_t log_modulus) {
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(lwe_out != lwe_in, "Output and input pointers must be different for out-of-place operations");
  launch_centered_modulus_switch(S(stream), (uint64_t *)lwe_out, (const uint64_t *)lwe_in, lwe_dimension,
                                 log_modulus);
}
// cuda/include/ciphertext.h:34-37, cuda/src/crypto/{ciphertext.cu:106-116, torus.cuh:435-465}: one LWE, block of shape
// (block_dim_x, block_dim_y); 128 or 512 threads, anything else panics as there
void cuda_centered_modulus_switch_cooperative_64_async(void *stream, uint32_t gpu_index, void *lwe_out, const void *lwe_in,
                                                       uint32_t lwe_dimension, uint32_t log_modulus, uint32_t block_dim_x,
                                                       uint32_t block_dim_y) {
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(lwe_out != lwe_in, "Output and input pointers must be different for out-of-place operations");
  HX_PANIC_IF_FALSE(launch_centered_modulus_switch_cooperative(S(stream), (uint64_t *)lwe_out, (const uint64_t *)lwe_in,
                                                               lwe_dimension, log_modulus, block_dim_x, block_dim_y),
                    "Unsupported block size for the cooperative centered modulus switch, supported sizes are 128 and 512 "
                    "threads per block");
}
// cuda/include/ciphertext.h:45-50, cuda/src/crypto/{ciphertext.cu:166-178, torus.cuh:612-653}: `size` words of
// lwe_array_in are read as size / grouping_factor groups (the reference's caller passes the whole ciphertext, body
// included; the integer division drops it), 2^g words per group are written.  As in the reference the switch goes to
// 2 * degree whatever log_modulus says (torus.cuh:158 uses params::log2_degree + 1) and only degree 2048 is accepted.
void cuda_modulus_switch_multi_bit_64_async(void *stream, uint32_t gpu_index, void *lwe_array_out, void *lwe_array_in,
                                            uint32_t size, uint32_t log_modulus, uint32_t degree,
                                            uint32_t grouping_factor) {
  set_device(gpu_index);
  (void)log_modulus;
  HX_PANIC_IF_FALSE(degree == 2048, "unsupported polynomial size. Supported N's are powers of two in the interval [2048].");
  HX_PANIC_IF_FALSE(grouping_factor >= 1 && grouping_factor <= 4, "unsupported grouping_factor %u", grouping_factor);
  launch_modulus_switch_multi_bit(S(stream), (uint64_t *)lwe_array_out, (const uint64_t *)lwe_array_in,
                                  size / grouping_factor, 12, grouping_factor);
}

// =========================================================================== extensions
void hip_backend_set_fft_kernel(uint32_t which) { g_fft_kernel_choice.store(which); }
uint64_t hip_backend_trim_allocator(uint32_t gpu_index) {
  set_device(gpu_index);
  return arena_trim((int)gpu_index);
}
void hip_backend_allocator_stats(uint32_t gpu_index, uint64_t *out7) {
  const ArenaStats s = arena_stats((int)gpu_index);
  out7[0] = s.allocations; out7[1] = s.reuses; out7[2] = s.runtime_allocations; out7[3] = s.frees;
  out7[4] = s.cross_stream_waits; out7[5] = s.live_bytes; out7[6] = s.cached_bytes;
}
uint64_t hip_backend_redzone_checks(uint32_t gpu_index) { return arena_redzone_checks((int)gpu_index); }
uint64_t hip_backend_profile_ranges(void) { return profile_range_count(); }
void hip_backend_set_keyswitch_kernel(uint32_t which) {
  g_keyswitch_use_mfma.store(which != 1);
  g_keyswitch_split_digits.store(which != 2);
  g_keyswitch_gemm_min.store(which == 3 ? 129u : 769u);
}
void hip_backend_set_ntt_kernel(uint32_t which) { g_ntt_kernel_serial = (which == 1); }
void hip_backend_set_multibit_latency_groups(uint32_t groups) { g_multibit_latency_groups.store(groups); }
uint32_t hip_backend_last_pbs_kernel(void) { return g_last_pbs_kernel.load(); }
uint32_t hip_backend_last_keyswitch_path(void) { return g_last_keyswitch_path.load(); }
void hip_backend_set_keyswitch_kparts(uint32_t parts) { g_keyswitch_kparts.store(parts ? parts : 1); }
const char *hip_backend_version(void) {
#if defined(TFHE_HIPEMU)
  return "tfhe-hip-backend 0.1 (HOST EMULATION - test build, not a product)";
#else
  return "tfhe-hip-backend 0.1 (gfx950)";
#endif
}
// HIP events on the caller's stream: bench.py times the kernels with these (torch.cuda.Event
// only sees torch's current stream).
void *hip_event_create(void) {
  hipEvent_t e;
  HX_CHECK(hipEventCreate(&e));
  return e;
}
void hip_event_record(void *event, void *stream) { HX_CHECK(hipEventRecord((hipEvent_t)event, S(stream))); }
float hip_event_elapsed_ms(void *start, void *stop) {
  float ms = 0.f;
  HX_CHECK(hipEventSynchronize((hipEvent_t)stop));
  HX_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
  return ms;
}
void hip_event_destroy(void *event) { HX_CHECK(hipEventDestroy((hipEvent_t)event)); }

// ---- the transform's own entry points (cuda/include/pbs/programmable_bootstrap.h:8-45; csrc/fourier.hip)
// cuda/src/pbs/bootstrapping_key.cu:175-355: sizes 256 .. 16384, any other falls through without a launch, as there
void cuda_fourier_polynomial_mul_async(void *stream, uint32_t gpu_index, void const *input1, void const *input2, void *output,
                                       uint32_t polynomial_size, uint32_t total_polynomials) {
  set_device(gpu_index);
  launch_fourier(S(stream), gpu_index, 3, const_cast<void *>(input1), input2, output, polynomial_size, total_polynomials);
}
// The reference restricts the next four to polynomial_size == 2048 (its throughput kernel's transform) and to sm_90; here they
// run the same generic transform at 2048 on any gfx950 device — same restriction on the size, same messages.
void cuda_fourier_polynomial_mul_fft16x4x16_async(void *stream, uint32_t gpu_index, void const *input1, void const *input2,
                                                  void *output, uint32_t polynomial_size, uint32_t total_polynomials) {
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(polynomial_size == 2048, "cuda_fourier_polynomial_mul_fft16x4x16_async only supports polynomial_size == 2048");
  launch_fourier(S(stream), gpu_index, 3, const_cast<void *>(input1), input2, output, polynomial_size, total_polynomials);
}
void cuda_forward_fft_classic_async(void *stream, uint32_t gpu_index, void const *input, void *output, uint32_t polynomial_size,
                                    uint32_t total_polynomials) {
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(polynomial_size == 2048, "cuda_forward_fft_classic_async only supports polynomial_size == 2048");
  launch_fourier(S(stream), gpu_index, 0, const_cast<void *>(input), nullptr, output, polynomial_size, total_polynomials);
}
void cuda_forward_fft16x4x16_async(void *stream, uint32_t gpu_index, void const *input, void *output, uint32_t polynomial_size,
                                   uint32_t total_polynomials) {
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(polynomial_size == 2048, "cuda_forward_fft16x4x16_async only supports polynomial_size == 2048");
  launch_fourier(S(stream), gpu_index, 1, const_cast<void *>(input), nullptr, output, polynomial_size, total_polynomials);
}
void cuda_backward_fft16x4x16_async(void *stream, uint32_t gpu_index, void const *input, void *output, uint32_t polynomial_size,
                                    uint32_t total_polynomials) {
  set_device(gpu_index);
  HX_PANIC_IF_FALSE(polynomial_size == 2048, "cuda_backward_fft16x4x16_async only supports polynomial_size == 2048");
  launch_fourier(S(stream), gpu_index, 2, const_cast<void *>(input), nullptr, output, polynomial_size, total_polynomials);
}
// the reference answers "compute capability 9.x"; these entry points exist on every device this library runs on
bool cuda_fft16x4x16_is_supported_async(uint32_t gpu_index) {
  set_device(gpu_index);
  return true;
}

void hip_test_arith_async(void *stream, uint32_t gpu_index, uint32_t op, void const *in, void *out, uint32_t count,
                          uint32_t p0, uint32_t p1) {
  set_device(gpu_index);
  launch_test_arith(S(stream), op, (const uint64_t *)in, (uint64_t *)out, count, p0, p1);
}
void hip_test_transform_async(void *stream, uint32_t gpu_index, uint32_t op, uint32_t polynomial_size,
                              void const *in, void *out) {
  set_device(gpu_index);
  check_pow2_poly(polynomial_size);
  launch_test_transform(S(stream), op, polynomial_size, in, out, gpu_index);
}
void hip_test_fft_tables_host(uint32_t polynomial_size, double *fwd, double *inv, double *untwist) {
  fill_fft_tables_host(polynomial_size, fwd, inv, untwist);
}
void hip_test_monomial_table_host(uint32_t polynomial_size, double *mono) {
  fill_monomial_table_host(polynomial_size, mono);
}

}  // extern "C"
