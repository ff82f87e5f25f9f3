// integer.hip — radix-integer layer on top of the KS -> PBS path ("next" row N1 of SURVEY §8):
// a batched apply-LUT round driver and, built from it, carry propagation, addition and
// schoolbook multiplication of radix ciphertexts.
//
// Replaces (host orchestration + three small kernels; all heavy work is the keyswitch and PBS
// kernels of this library):
//   backends/tfhe-cuda-backend/cuda/src/integer/integer.cuh:869-990   apply_univariate_lookup_table
//   .../integer/integer.cuh:1266-1305                                 LUT generation
//   .../integer/integer.cuh (propagate_single_carry), addition.cuh, multiplication.cuh
//   semantics: tfhe/src/integer/server_key/radix_parallel/{add.rs,mul.rs}, shortint bivariate_pbs.rs
//
// MI355X-first differences from the reference's shape:
//   * a CudaRadixCiphertextFFI may hold a BATCH of independent integers ([ciphertext][block], the
//     scratch's num_blocks = blocks per integer); every round is then ONE keyswitch launch and ONE
//     PBS launch over all blocks of all integers (thousands of PBS per launch is where the PBS
//     kernel is efficient), instead of one stream per integer.
//   * rounds are described by device index arrays (gather / scatter / LUT index per block), so no
//     ciphertext is ever moved to be "aligned" for a round.
#include "kernels.h"
#include "arena.h"
#include "profile.h"
#include "../../include/tfhe_hip_backend.h"

#include <algorithm>
#include <functional>
#include <map>
#include <mutex>
#include <vector>

namespace tfhe_hip {
namespace radix {

// ------------------------------------------------------------------ device kernels
// out[oi] = a[ai] * scalar + b[bi]   (b optional), `words` u64 per LWE; null index = identity
__global__ void __launch_bounds__(256) lwe_axpy_kernel(uint64_t *out, const uint64_t *out_idx, const uint64_t *a,
                                                       const uint64_t *a_idx, uint64_t scalar, const uint64_t *b,
                                                       const uint64_t *b_idx, uint32_t words, uint32_t count) {
  const uint32_t s = blockIdx.x;
  if (s >= count) return;
  const size_t oi = out_idx ? out_idx[s] : s, ai = a_idx ? a_idx[s] : s;
  const uint64_t *pa = a + ai * words;
  uint64_t *po = out + oi * words;
  if (b) {
    const uint64_t *pb = b + (b_idx ? b_idx[s] : s) * words;
    for (uint32_t j = threadIdx.x; j < words; j += blockDim.x) po[j] = pa[j] * scalar + pb[j];
  } else {
    for (uint32_t j = threadIdx.x; j < words; j += blockDim.x) po[j] = pa[j] * scalar;
  }
}
// out[g] = sum of pool[members[m]] for m in [offsets[g], offsets[g+1])  (CSR groups)
__global__ void __launch_bounds__(256) lwe_group_sum_kernel(uint64_t *out, const uint64_t *pool,
                                                            const uint64_t *offsets, const uint64_t *members,
                                                            uint32_t words, uint32_t groups) {
  const uint32_t g = blockIdx.x;
  if (g >= groups) return;
  const uint64_t lo = offsets[g], hi = offsets[g + 1];
  for (uint32_t j = threadIdx.x; j < words; j += blockDim.x) {
    uint64_t acc = 0;
    for (uint64_t m = lo; m < hi; ++m) acc += pool[members[m] * words + j];
    out[(size_t)g * words + j] = acc;
  }
}

// Levelled block operations (no bootstrap).  out[s] = -in[s] on every word, the body takes c0 (block 0 of an integer of
// `per` blocks) or c1 (the other blocks) on top: the negation with its correcting term (negation.cuh:20-49), the
// bitwise NOT (bitwise_ops.cuh:164-191) and the plain negation are instances.  In place allowed.
__global__ void __launch_bounds__(256) lwe_negate_const_kernel(uint64_t *out, const uint64_t *in, uint32_t words,
                                                               uint32_t count, uint32_t per, uint64_t c0, uint64_t c1) {
  const uint32_t s = blockIdx.x;
  if (s >= count) return;
  const uint64_t *pi = in + (size_t)s * words;
  uint64_t *po = out + (size_t)s * words;
  const uint64_t c = (s % per) == 0 ? c0 : c1;
  for (uint32_t j = threadIdx.x; j < words; j += blockDim.x) po[j] = ((uint64_t)0 - pi[j]) + (j + 1 == words ? c : 0);
}
// body[s] += scalars[s] * delta (scalar_addition.cuh:14-25): one thread per block
__global__ void __launch_bounds__(256) lwe_body_add_scalars_kernel(uint64_t *v, const uint64_t *scalars, uint32_t words,
                                                                   uint32_t count, uint64_t delta) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < count) v[(size_t)s * words + words - 1] += scalars[s] * delta;
}

static void axpy(hipStream_t st, uint64_t *out, const uint64_t *out_idx, const uint64_t *a, const uint64_t *a_idx,
                 uint64_t scalar, const uint64_t *b, const uint64_t *b_idx, uint32_t words, uint32_t count) {
  if (count == 0) return;
  HX_LAUNCH(lwe_axpy_kernel, dim3(count), dim3(256), 0, st, out, out_idx, a, a_idx, scalar, b, b_idx, words, count);
}

// ------------------------------------------------------------------ host helpers
struct Params {
  uint32_t big_n, small_n, k, N, pbs_base_log, pbs_level, ks_base_log, ks_level, msg, carry, ms_type;
  uint32_t grouping;  // 0: classic PBS; g >= 1: multi-bit PBS with that grouping factor (pbs_type = MULTI_BIT)
};

static Params make_params(CudaLweBootstrapKeyParamsFFI b, CudaLweKeyswitchKeyParamsFFI k, uint32_t msg, uint32_t carry,
                          uint32_t ms_type) {
  HX_PANIC_IF_FALSE(b.pbs_type == CLASSICAL || b.pbs_type == MULTI_BIT, "radix layer: unknown pbs_type %u", b.pbs_type);
  HX_PANIC_IF_FALSE(b.pbs_type == CLASSICAL || (b.grouping_factor >= 1 && b.grouping_factor <= 4 &&
                                                b.input_lwe_dimension % b.grouping_factor == 0),
                    "radix layer: multi-bit PBS needs a grouping factor in 1..4 dividing the LWE dimension (got %u)",
                    b.grouping_factor);
  HX_PANIC_IF_FALSE(b.glwe_dimension * b.polynomial_size == k.input_lwe_dimension &&
                        k.output_lwe_dimension == b.input_lwe_dimension,
                    "radix layer: keyswitch and bootstrap key dimensions do not chain");
  HX_PANIC_IF_FALSE(msg >= 2 && carry >= msg && (b.polynomial_size % (msg * carry)) == 0,
                    "radix layer: unsupported message/carry moduli (%u, %u)", msg, carry);
  return Params{k.input_lwe_dimension, b.input_lwe_dimension, b.glwe_dimension, b.polynomial_size, b.base_log,
                b.level_count,         k.base_log,            k.level_count,    msg,               carry,
                ms_type,               b.pbs_type == MULTI_BIT ? b.grouping_factor : 0u};
}

// cuda/src/integer/integer.cuh:1266-1305 (generate_lookup_table_with_encoding, same in/out encoding)
static void generate_lut(const Params &p, uint64_t *acc, const std::function<uint64_t(uint64_t)> &f) {
  const uint32_t sup = p.msg * p.carry, box = p.N / sup;
  const uint64_t delta = ((uint64_t)1 << 63) / sup;
  std::fill(acc, acc + (size_t)p.k * p.N, 0);
  uint64_t *body = acc + (size_t)p.k * p.N;
  for (uint32_t i = 0; i < sup; ++i)
    for (uint32_t j = i * box; j < (i + 1) * box; ++j) body[j] = f(i) * delta;
  const uint32_t half = box / 2;
  for (uint32_t i = 0; i < half; ++i) body[i] = (uint64_t)0 - body[i];
  std::rotate(body, body + half, body + p.N);
}

// shortint's many-LUT accumulator (tfhe/src/shortint/engine/mod.rs:169-254 fill_many_lut_accumulator): the plaintext
// space is shared by fs.size() functions of inputs below sup / fs.size(); function t sits in sub-table t, which the
// sample extraction reaches at coefficient t * many_lut_stride(p, fs.size())
static uint32_t many_lut_stride(const Params &p, uint32_t fn) { return p.N / fn; }
static void generate_many_lut(const Params &p, uint64_t *acc, const std::vector<std::function<uint64_t(uint64_t)>> &fs) {
  const uint32_t sup = p.msg * p.carry, box = p.N / sup, fn = (uint32_t)fs.size(), inputs = sup / fn;
  const uint64_t delta = ((uint64_t)1 << 63) / sup;
  std::fill(acc, acc + (size_t)(p.k + 1) * p.N, 0);
  uint64_t *body = acc + (size_t)p.k * p.N;
  for (uint32_t t = 0; t < fn; ++t)
    for (uint32_t i = 0; i < inputs; ++i)
      for (uint32_t j = 0; j < box; ++j) body[(size_t)t * inputs * box + (size_t)i * box + j] = fs[t](i) * delta;
  const uint32_t half = box / 2;
  for (uint32_t i = 0; i < half; ++i) body[i] = (uint64_t)0 - body[i];
  std::rotate(body, body + half, body + p.N);
}

// scratch_* with allocate_gpu_memory = false only reports the device bytes the scratch would take
// (gpu/ffi.rs:95-132 "size on gpu" queries): allocations are counted, not made, and nothing is uploaded
static thread_local bool t_dry = false;
static thread_local uint64_t t_bytes = 0;
static void radix_alloc(void **p, size_t bytes) {
  t_bytes += bytes;
  *p = nullptr;
  if (!t_dry) *p = scratch_alloc(bytes);
}

template <class T>
static T *dev_upload(hipStream_t st, const std::vector<T> &h) {
  T *d = nullptr;
  if (h.empty()) return d;
  if (t_dry) {
    t_bytes += h.size() * sizeof(T);
    return d;
  }
  d = (T *)scratch_alloc(h.size() * sizeof(T));
  HX_CHECK(hipMemcpyAsync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
  HX_CHECK(hipStreamSynchronize(st));  // h may be a temporary
  return d;
}

// The KS -> PBS round driver (integer.cuh:869-990): `count` blocks gathered from `in` through in_idx are
// keyswitched, bootstrapped with LUT lut_idx[s] and scattered to out[out_idx[s]].
//
// Multi-GPU (helper_multi_gpu.cuh:170-294, helper_multi_gpu.cu:39-101): the ciphertexts live on the first GPU of
// the stream set; a round over enough blocks is split into contiguous shards by the reference's rule
// (get_num_inputs_on_gpu), GPU 0 works on its shard in place, every other active GPU i receives its shard
// (gathered on GPU 0, one peer copy on stream i), runs the same two launches with ITS key replicas
// (ksks[i], bsks[i]) and scratch, and sends the results back for a scatter on GPU 0.  Events order the streams;
// no host synchronisation, no collective.  The same code runs with several streams on ONE device (the
// reference's debug-fake-multi-gpu idea), which is how the 1-GPU test box exercises it.
// Blocks per GPU from which a round spreads over one more GPU of the stream set.  0 (default): the reference's rule
// (get_active_gpu_count, helper_multi_gpu.cu:16-48) — 12 for multi-bit keys, (compute units of the first GPU) + 1 for
// classic ones; hip_integer_set_multi_gpu_threshold overrides it (tests, tuning).  The crossover on 8 MI355X is not
// measured (one GPU per box): the rule is the reference's, the setter is there to move it.
static std::atomic<uint32_t> g_multi_gpu_min_blocks{0};
static uint32_t multi_gpu_threshold(const Params &p, uint32_t first_gpu) {
  const uint32_t forced = g_multi_gpu_min_blocks.load();
  if (forced != 0) return forced;
  if (p.grouping) return 12;  // THRESHOLD_MULTI_GPU_WITH_MULTI_BIT_PARAMS
  static std::atomic<uint32_t> cus{0};  // get_threshold_multi_gpu_classical: computed once, first GPU's count
  uint32_t c = cus.load();
  if (c == 0) {
    hipDeviceProp_t prop;
    HX_CHECK(hipGetDeviceProperties(&prop, (int)first_gpu));
    c = (uint32_t)prop.multiProcessorCount;
    cus.store(c);
  }
  return c + 1;
}

// Copies between the GPUs of a stream set: direct (peer access enabled once per ordered pair of devices) where the
// devices can reach each other, through a pinned host buffer where they cannot (helper_multi_gpu.cuh:170-294 relies on
// cudaMemcpyPeerAsync's own fallback; here the two paths are explicit so that both are testable).
static std::mutex g_peer_mutex;
static std::map<std::pair<int, int>, bool> g_peer_direct;  // (device that issues the copy, other device) -> direct?
static bool peer_direct(int dev, int other) {
  if (dev == other) return true;
  std::lock_guard<std::mutex> lock(g_peer_mutex);
  auto it = g_peer_direct.find({dev, other});
  if (it != g_peer_direct.end()) return it->second;
  int can = 0, cur = 0;
  HX_CHECK(hipGetDevice(&cur));
  HX_CHECK(hipDeviceCanAccessPeer(&can, dev, other));
  if (can) {
    HX_CHECK(hipSetDevice(dev));
    const hipError_t e = hipDeviceEnablePeerAccess(other, 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
    (void)hipGetLastError();
    HX_CHECK(hipSetDevice(cur));
  }
  g_peer_direct[{dev, other}] = can != 0;
  return can != 0;
}

static uint32_t num_inputs_on_gpu(uint32_t total, uint32_t gpu, uint32_t gpus) {  // helper_multi_gpu.cu:71-101
  if (gpus > total) return gpu < total ? 1u : 0u;
  const uint32_t large = (total + gpus - 1) / gpus, small = total / gpus, cutoff = total % gpus;
  return (cutoff != 0 && gpu < cutoff) ? large : small;
}

struct LutDriver {
  static constexpr uint32_t kMagic = 0x52445231;  // "RDR1"
  uint32_t magic = kMagic;
  Params p{};
  uint32_t cap = 0, num_luts = 0, many_max = 1;
  struct PerGpu {
    uint32_t gpu = 0;
    uint64_t *d_ks = nullptr, *d_luts = nullptr, *d_trivial = nullptr;  // d_trivial = 0, 1, ..., cap - 1
    int8_t *pbs_buf = nullptr;
    // GPUs other than the first: shard buffers on this GPU and their twins on the first GPU
    uint64_t *d_in = nullptr, *d_out = nullptr, *d_lut_idx = nullptr;   // on this GPU
    uint64_t *d0_in = nullptr, *d0_out = nullptr;                       // on the first GPU
    uint64_t *d_many = nullptr;  // first GPU only, drivers created for many-LUT rounds: dense outputs before the scatter
    bool direct = true;          // this GPU and the first one reach each other's memory (peer access enabled)
    uint64_t *h_stage = nullptr; // pinned host buffer for the copies when they do not
    hipEvent_t staged = nullptr, done = nullptr, copied = nullptr;
  };
  std::vector<PerGpu> gpus;

  void scratch_pbs(hipStream_t st, PerGpu &g) {
    if (p.grouping)
      scratch_cuda_multi_bit_programmable_bootstrap_64_async(st, g.gpu, &g.pbs_buf, p.k, p.N, p.pbs_level, cap, !t_dry);
    else
      scratch_cuda_programmable_bootstrap_64_async(st, g.gpu, &g.pbs_buf, p.small_n, p.k, p.N, p.pbs_level, cap,
                                                   !t_dry, (enum PBS_MS_REDUCTION_T)p.ms_type);
  }

  // max_many: the most functions one round extracts per bootstrap (sizes the dense output buffers)
  void init(const CudaStreamsFFI &s, const Params &params, uint32_t capacity,
            const std::vector<std::vector<uint64_t>> &luts, uint32_t max_many = 1) {
    HX_PANIC_IF_FALSE(s.gpu_count >= 1 && s.streams != nullptr, "radix layer: empty stream set");
    p = params;
    cap = capacity;
    many_max = std::max(1u, max_many);
    num_luts = (uint32_t)luts.size();
    const size_t lw = (size_t)(p.k + 1) * p.N, w = (size_t)p.big_n + 1;
    std::vector<uint64_t> triv(cap);
    for (uint32_t i = 0; i < cap; ++i) triv[i] = i;
    gpus.resize(s.gpu_count);
    for (uint32_t i = 0; i < s.gpu_count; ++i) {
      PerGpu &g = gpus[i];
      g.gpu = s.gpu_indexes ? s.gpu_indexes[i] : 0;
      const hipStream_t st = (hipStream_t)s.streams[i];
      HX_CHECK(hipSetDevice((int)g.gpu));
      radix_alloc((void **)&g.d_ks, (size_t)cap * (p.small_n + 1) * sizeof(uint64_t));
      radix_alloc((void **)&g.d_luts, std::max<size_t>(1, num_luts) * lw * sizeof(uint64_t));
      for (uint32_t t = 0; t < num_luts && !t_dry; ++t)
        HX_CHECK(hipMemcpyAsync(g.d_luts + t * lw, luts[t].data(), lw * sizeof(uint64_t), hipMemcpyHostToDevice, st));
      g.d_trivial = dev_upload(st, triv);
      if (i == 0 && many_max > 1) radix_alloc((void **)&g.d_many, (size_t)many_max * cap * w * sizeof(uint64_t));
      if (i > 0) {
        radix_alloc((void **)&g.d_in, (size_t)cap * w * sizeof(uint64_t));
        radix_alloc((void **)&g.d_out, (size_t)many_max * cap * w * sizeof(uint64_t));
        radix_alloc((void **)&g.d_lut_idx, (size_t)cap * sizeof(uint64_t));
        // An event is recorded on a stream of ITS device only (hipEventRecord rejects a foreign stream): `staged` and
        // `copied` are recorded on the first GPU's stream, `done` on this GPU's; waiting across devices is allowed.
        HX_CHECK(hipSetDevice((int)gpus[0].gpu));
        radix_alloc((void **)&g.d0_in, (size_t)cap * w * sizeof(uint64_t));
        radix_alloc((void **)&g.d0_out, (size_t)many_max * cap * w * sizeof(uint64_t));
        if (!t_dry) {
          HX_CHECK(hipEventCreateWithFlags(&g.staged, hipEventDisableTiming));
          HX_CHECK(hipEventCreateWithFlags(&g.copied, hipEventDisableTiming));
        }
        HX_CHECK(hipSetDevice((int)g.gpu));
        if (!t_dry) {
          HX_CHECK(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
          // the copies of a round are issued on this GPU's stream in both directions
          g.direct = peer_direct((int)g.gpu, (int)gpus[0].gpu) && peer_direct((int)gpus[0].gpu, (int)g.gpu);
          // host-staged copies: three pinned regions — shard in, its LUT indexes, results out — so that each leg of a copy
          // runs on the stream of the device whose memory it touches and no region is rewritten before its reader is done
          if (!g.direct) HX_CHECK(hipHostMalloc((void **)&g.h_stage, ((size_t)cap * w + cap + (size_t)many_max * cap * w) * sizeof(uint64_t), 0));
        }
      }
      HX_CHECK(hipStreamSynchronize(st));  // the LUT sources may be temporaries
      scratch_pbs(st, g);
    }
    HX_CHECK(hipSetDevice((int)gpus[0].gpu));
  }

  void ks_pbs(hipStream_t st, const PerGpu &g, uint64_t *out, const uint64_t *out_idx, const uint64_t *in,
              const uint64_t *in_idx, const uint64_t *lut_idx, uint32_t c, const void *ksk, const void *bsk,
              uint32_t many = 1, uint32_t stride = 0) const {
    {
      HX_RANGE("keyswitch (%u blocks, gpu %u)", c, g.gpu);
      cuda_keyswitch_lwe_ciphertext_vector_64_64_async(st, g.gpu, g.d_ks, g.d_trivial, in, in_idx, ksk, p.big_n, p.small_n,
                                                       p.ks_base_log, p.ks_level, c);
    }
    HX_RANGE("bootstrap (%u blocks, %u functions, gpu %u)", c, many, g.gpu);
    if (p.grouping)
      cuda_multi_bit_programmable_bootstrap_64_async(st, g.gpu, out, out_idx, g.d_luts, lut_idx, g.d_ks, g.d_trivial, bsk,
                                                     g.pbs_buf, p.small_n, p.k, p.N, p.grouping, p.pbs_base_log,
                                                     p.pbs_level, c, many, stride);
    else
      cuda_programmable_bootstrap_64_async(st, g.gpu, out, out_idx, g.d_luts, lut_idx, g.d_ks, g.d_trivial, bsk, g.pbs_buf,
                                           p.small_n, p.k, p.N, p.pbs_base_log, p.pbs_level, c, many, stride);
  }

  // One keyswitch, one PBS that extracts `many` functions out of the same accumulator (sample extraction at
  // coefficients t * stride, integer.cuh:1002-1110): function t of block s lands in out block t * count + s.  The
  // PBS kernels place function t at t * (samples of the launch), so the round is ONE launch on the first GPU of the
  // set (count <= cap by construction of the scratch); a stream set with several GPUs is accepted, the others idle.
  void round_many(const CudaStreamsFFI &s, uint64_t *out, const uint64_t *in, const uint64_t *lut_idx, uint32_t count,
                  void *const *ksks, void *const *bsks, uint32_t many, uint32_t stride) const {
    HX_PANIC_IF_FALSE(count <= cap, "apply_many_univariate_lut: more blocks than the scratch was created for");
    HX_CHECK(hipSetDevice((int)gpus[0].gpu));
    ks_pbs((hipStream_t)s.streams[0], gpus[0], out, gpus[0].d_trivial, in, gpus[0].d_trivial, lut_idx, count, ksks[0],
           bsks[0], many, stride);
  }

  // Copies between this GPU and the first one.  direct: hipMemcpyPeerAsync on this GPU's stream.  Host-staged (no peer
  // access): the device-to-host leg on the stream of the device that owns the source, the host-to-device leg on the stream
  // of the device that owns the destination, ordered by the round's events (staged / done) — a stream never touches another
  // device's memory.  Not reachable on an MI355X node (every pair of GPUs is an xGMI peer): exercised on the
  // CPU tier's device model only.
  uint64_t *h_in(const PerGpu &g) const { return g.h_stage; }
  uint64_t *h_idx(const PerGpu &g) const { return g.h_stage + (size_t)cap * ((size_t)p.big_n + 1); }
  uint64_t *h_out(const PerGpu &g) const { return h_idx(g) + cap; }

  // one round, split into launches of at most `cap` blocks; a null in_idx / out_idx means "block s".
  // many > 1: every bootstrap extracts `many` functions of its accumulator (coefficients t * stride); out_idx then
  // holds many * count entries, function t of block s goes to out[out_idx[t * count + s]] (the PBS kernels place
  // function t behind the launch's own samples, so these rounds go through a dense buffer and a scatter).
  void round(const CudaStreamsFFI &s, uint64_t *out, const uint64_t *out_idx, const uint64_t *in, const uint64_t *in_idx,
             const uint64_t *lut_idx, uint32_t count, void *const *ksks, void *const *bsks, uint32_t many = 1,
             uint32_t stride = 0) const {
    const size_t w = (size_t)p.big_n + 1;
    HX_PANIC_IF_FALSE(many >= 1 && many <= many_max && (many == 1 || out_idx != nullptr),
                      "radix layer: a round of %u functions per bootstrap on a driver created for %u", many, many_max);
    const uint32_t avail = std::min<uint32_t>(s.gpu_count, (uint32_t)gpus.size());
    const hipStream_t st0 = (hipStream_t)s.streams[0];
    HX_RANGE("apply lut round (%u blocks)", count);  // integer.cuh:874
    for (uint32_t off = 0; off < count; off += cap) {
      const uint32_t c = std::min(cap, count - off);
      // helper_multi_gpu.cu:39-48 (get_active_gpu_count): as many GPUs as the round can keep busy
      const uint32_t min_blocks = multi_gpu_threshold(p, gpus[0].gpu);
      const uint32_t active = std::max(1u, std::min(avail, (c + min_blocks - 1) / min_blocks));
      const uint64_t *ii = in_idx ? in_idx + off : nullptr, *oi = out_idx ? out_idx + off : nullptr;
      const uint64_t *in0 = in_idx ? in : in + off * w;
      uint64_t *out0 = out_idx ? out : out + off * w;
      uint32_t first = num_inputs_on_gpu(c, 0, active);
      // shards of the other GPUs: gather on the first GPU, ship, compute, ship back (all asynchronous)
      uint32_t begin = first;
      for (uint32_t i = 1; i < active; ++i) {
        const PerGpu &g = gpus[i];
        const hipStream_t sti = (hipStream_t)s.streams[i];
        const uint32_t ci = num_inputs_on_gpu(c, i, active);
        if (ci == 0) continue;
        HX_RANGE("scatter %u blocks to gpu %u", ci, g.gpu);  // integer.cuh:958 (the range covers the shard's launches too)
        HX_CHECK(hipSetDevice((int)gpus[0].gpu));
        if (ii)
          axpy(st0, g.d0_in, nullptr, in0, ii + begin, 1, nullptr, nullptr, (uint32_t)w, ci);
        else  // trivial input indexing: the shard starts at block `begin`
          HX_CHECK(hipMemcpyAsync(g.d0_in, in0 + (size_t)begin * w, (size_t)ci * w * sizeof(uint64_t),
                                  hipMemcpyDeviceToDevice, st0));
        const size_t in_bytes = (size_t)ci * w * sizeof(uint64_t), idx_bytes = (size_t)ci * sizeof(uint64_t),
                     out_bytes = (size_t)many * ci * w * sizeof(uint64_t);
        if (!g.direct) {  // first legs on the first GPU's stream, which owns the sources
          HX_CHECK(hipMemcpyAsync(h_in(g), g.d0_in, in_bytes, hipMemcpyDeviceToHost, st0));
          HX_CHECK(hipMemcpyAsync(h_idx(g), lut_idx + off + begin, idx_bytes, hipMemcpyDeviceToHost, st0));
        }
        HX_CHECK(hipEventRecord(g.staged, st0));
        HX_CHECK(hipSetDevice((int)g.gpu));
        HX_CHECK(hipStreamWaitEvent(sti, g.staged, 0));
        if (g.direct) {
          HX_CHECK(hipMemcpyPeerAsync(g.d_in, (int)g.gpu, g.d0_in, (int)gpus[0].gpu, in_bytes, sti));
          HX_CHECK(hipMemcpyPeerAsync(g.d_lut_idx, (int)g.gpu, lut_idx + off + begin, (int)gpus[0].gpu, idx_bytes, sti));
        } else {
          HX_CHECK(hipMemcpyAsync(g.d_in, h_in(g), in_bytes, hipMemcpyHostToDevice, sti));
          HX_CHECK(hipMemcpyAsync(g.d_lut_idx, h_idx(g), idx_bytes, hipMemcpyHostToDevice, sti));
        }
        ks_pbs(sti, g, g.d_out, g.d_trivial, g.d_in, g.d_trivial, g.d_lut_idx, ci, ksks[i], bsks[i], many, stride);
        if (g.direct) HX_CHECK(hipMemcpyPeerAsync(g.d0_out, (int)gpus[0].gpu, g.d_out, (int)g.gpu, out_bytes, sti));
        else HX_CHECK(hipMemcpyAsync(h_out(g), g.d_out, out_bytes, hipMemcpyDeviceToHost, sti));  // second leg below, on st0
        HX_CHECK(hipEventRecord(g.done, sti));
        begin += ci;
      }
      // the first GPU's own shard, in place
      HX_CHECK(hipSetDevice((int)gpus[0].gpu));
      if (many == 1) {
        ks_pbs(st0, gpus[0], out0, oi ? oi : gpus[0].d_trivial, in0, ii ? ii : gpus[0].d_trivial, lut_idx + off, first,
               ksks[0], bsks[0]);
      } else {
        ks_pbs(st0, gpus[0], gpus[0].d_many, gpus[0].d_trivial, in0, ii ? ii : gpus[0].d_trivial, lut_idx + off, first,
               ksks[0], bsks[0], many, stride);
        for (uint32_t t = 0; t < many; ++t)
          axpy(st0, out, out_idx + (size_t)t * count + off, gpus[0].d_many + (size_t)t * first * w, nullptr, 1, nullptr,
               nullptr, (uint32_t)w, first);
      }
      // results of the other GPUs: scatter on the first GPU
      begin = first;
      for (uint32_t i = 1; i < active; ++i) {
        const PerGpu &g = gpus[i];
        const uint32_t ci = num_inputs_on_gpu(c, i, active);
        if (ci == 0) continue;
        HX_RANGE("gather %u blocks from gpu %u", ci, g.gpu);  // integer.cuh:981
        HX_CHECK(hipStreamWaitEvent(st0, g.done, 0));
        if (!g.direct)
          HX_CHECK(hipMemcpyAsync(g.d0_out, h_out(g), (size_t)many * ci * w * sizeof(uint64_t), hipMemcpyHostToDevice, st0));
        if (many > 1)
          for (uint32_t t = 0; t < many; ++t)
            axpy(st0, out, out_idx + (size_t)t * count + off + begin, g.d0_out + (size_t)t * ci * w, nullptr, 1, nullptr,
                 nullptr, (uint32_t)w, ci);
        else if (oi)
          axpy(st0, out0, oi + begin, g.d0_out, nullptr, 1, nullptr, nullptr, (uint32_t)w, ci);
        else
          HX_CHECK(hipMemcpyAsync(out0 + (size_t)begin * w, g.d0_out, (size_t)ci * w * sizeof(uint64_t),
                                  hipMemcpyDeviceToDevice, st0));
        // the next use of this GPU's staging buffers must wait for the scatter
        HX_CHECK(hipEventRecord(g.copied, st0));
        HX_CHECK(hipSetDevice((int)g.gpu));
        HX_CHECK(hipStreamWaitEvent((hipStream_t)s.streams[i], g.copied, 0));
        HX_CHECK(hipSetDevice((int)gpus[0].gpu));
        begin += ci;
      }
    }
  }

  void release(const CudaStreamsFFI &s) {
    for (uint32_t i = 0; i < (uint32_t)gpus.size(); ++i) {
      PerGpu &g = gpus[i];
      const hipStream_t st = (hipStream_t)s.streams[i < s.gpu_count ? i : 0];
      HX_CHECK(hipSetDevice((int)g.gpu));
      HX_CHECK(hipStreamSynchronize(st));
      if (g.pbs_buf) {
        if (p.grouping)
          cleanup_cuda_multi_bit_programmable_bootstrap_64(st, g.gpu, &g.pbs_buf);
        else
          cleanup_cuda_programmable_bootstrap_64(st, g.gpu, &g.pbs_buf);
      }
      for (uint64_t *d : {g.d_ks, g.d_luts, g.d_trivial, g.d_in, g.d_out, g.d_lut_idx, g.d_many})
        if (d) scratch_free(d);
      for (hipEvent_t e : {g.staged, g.done, g.copied})
        if (e) HX_CHECK(hipEventDestroy(e));
      if (g.h_stage) HX_CHECK(hipHostFree(g.h_stage));
      HX_CHECK(hipSetDevice((int)gpus[0].gpu));
      for (uint64_t *d : {g.d0_in, g.d0_out})
        if (d) scratch_free(d);
    }
    gpus.clear();
    magic = 0;
  }
};

static hipStream_t S0(const CudaStreamsFFI &s) {
  HX_PANIC_IF_FALSE(s.gpu_count >= 1 && s.streams != nullptr, "radix layer: empty stream set");
  return (hipStream_t)s.streams[0];
}
static uint32_t G0(const CudaStreamsFFI &s) { return s.gpu_indexes ? s.gpu_indexes[0] : 0; }
// every entry point of the radix layer starts on the first GPU of its stream set: scratch_alloc / scratch_free and the index
// uploads act on the thread's CURRENT device, which a host alternating between GPUs may have left elsewhere
static void first_gpu(const CudaStreamsFFI &s) { HX_CHECK(hipSetDevice((int)G0(s))); }

// ------------------------------------------------------------------ apply a univariate LUT
struct ApplyLutMem {
  static constexpr uint32_t kMagic = 0x4C555431;  // "LUT1"
  uint32_t magic = kMagic;
  bool size_only = false;
  LutDriver drv;
  uint64_t *d_lut_idx = nullptr;  // all zero
  uint64_t degree = 0;
  uint32_t num_many_lut = 1;      // apply_many_univariate_lut: functions packed in the accumulator
};

// ------------------------------------------------------------------ carry propagation
// In place on blocks whose value is <= 2*msg - 2 (the sum of two clean blocks): at most one carry leaves a
// block.  Block states as in radix_parallel/add.rs (OutputCarry): none, generated, propagated.
//
// Carry look-ahead as a TREE OF BINARY ADDITIONS (needs msg*carry >= 16; the same family as the reference's
// advanced_add_assign_with_carry_at_least_4_bits, add.rs:828-1044, which also takes its block states from
// many-LUT bootstraps).  Among up to three neighbours the carries are resolved by an ordinary binary addition of
// their states, which is LINEAR in the ciphertexts:
//   w_q = (generated ? 2 : propagated ? 1 : 0) << q          for the neighbour at position q = 0..2
//   S_q = carry_in + w_0 + ... + w_(q-1)                       LWE additions, no PBS
//   carry into neighbour q = bit q of S_q                      (one PBS, q >= 1)
// because a propagating neighbour (bit q set) forwards an incoming carry to bit q+1, a generating one has bit
// q+1 set whatever arrives, and an absorbing one stops it.  The state of the three TOGETHER is read off
// U = w_0 + w_1 + w_2 (bit 3 of U: a carry leaves without an incoming one; bit 3 of U + 1: with one) by one PBS,
// which emits it already shifted for its own position one level up.  So: blocks -> groups of 3 -> groups of 9
// -> ... until at most 4 elements are left, whose carries are bits 1..3 of the partial sums; then the carries go
// back down, one level per round (the first element of a group takes the carry of its group as it is).
// The blocks themselves take no round of their own on the way down: the first bootstrap of a block (its value is
// below msg*carry / 2, so the accumulator holds two functions) also returns 4 * (value % msg), and the last one
// reads the result off  4 * (value % msg) + z  with z = (state of the blocks before it in its group of three:
// 0 none, 1 propagated, 2 generated) + (carry into the group): the block receives a carry iff z >= 2.
// PBS for 32 blocks: 32 first bootstraps + 10 + 3 group states + 10 states of the first two blocks of a group
// + 3 top carries + 7 carries into the groups + 32 results = 97 in 6 rounds (round 3, before: 107 in 7 with a
// round for the carries into the blocks; a Hillis-Steele scan over groups of four 116 in 8; one per block 224).
enum : uint64_t {
  LUT_FIRST2 = 0,    // many-LUT accumulators [f, 4 (x % msg)]: + q: f = state << q (q = 0..2);
                     // + 3: block 0 of an integer, f = state with "propagated" dropped (nothing can arrive);
                     // + 4: last block of an integer, f = 4 * state (only the output carry reads it)
  LUT_GROUP = 5,     // + 3 (len - 1) + q: U of a group of len = 1..3 -> state of the group << q
  LUT_BIT = 13,      // + q (q = 1..3): S -> bit q of S
  LUT_RES_BIT = 17,  // 4 m + z -> (m + z) % msg, z = the carry itself (first block of a group, or the top level)
  LUT_RES_Z = 18,    // 4 m + z -> (m + (z >= 2)) % msg
  LUT_OUT_BIT = 19,  // 4 state + z -> output carry of the integer, z = the carry into its last block
  LUT_OUT_Z = 20,    // 4 state + z -> the same, the carry into the last block is z >= 2
  LUT_MSG = 21,      // x -> x % msg                  (single-block integers)
  LUT_CARRY = 22,    // x -> x / msg
  // signed overflow of an addition (FLAG_OVERFLOW, integer_utilities.h:2311-2357, :2383-2412): the last blocks of the
  // two operands, packed msg * lhs + rhs BEFORE the addition, give 8 (overflow if a carry arrives) + 4 (if none);
  // the flag is read off that + z, z as in LUT_OUT_*
  LUT_OVF_PREP = 23,
  LUT_OVF_BIT = 24,  // 8 o1 + 4 o0 + z -> z ? o1 : o0      (z = the carry into the last block itself)
  LUT_OVF_Z = 25,    // 8 o1 + 4 o0 + z -> (z >= 2) ? o1 : o0
  LUT_PROP_COUNT = 26
};

struct PropagateMem {
  static constexpr uint32_t kMagic = 0x50524F50;  // "PROP"
  static constexpr uint32_t G = 3, TOP = 4;       // children per group; elements the top level resolves directly
  uint32_t magic = kMagic;
  bool size_only = false;
  LutDriver drv;
  uint32_t blocks = 0;   // blocks per integer
  uint32_t max_cts = 0;  // integers the scratch was sized for
  // scratch ciphertexts: pool = [W of level 0 (the blocks), W of level 1, ... | C of level 0 (top level only), C of
  // level 1, ... | M: 4 (value % msg) of every block | P: state of the first two blocks of every full group | O: the
  // overflow preparation of every integer (FLAG_OVERFLOW)]
  // (shifted states; carries into the elements); P2: dense sums of one round
  uint64_t *d_pool = nullptr, *d_p = nullptr;
  uint32_t cached_cts = 0;
  std::vector<uint64_t *> dev_arrays;
  struct Idx {
    uint64_t *a = nullptr, *b = nullptr, *o = nullptr, *lut = nullptr;
    uint32_t count = 0;
  };
  Idx rA, rTop, rRes, rOut, rOvf, rOvfPrep, rIO, rOne, rLast, rOne1;
  uint64_t *d_pack = nullptr;  // msg * (last block of lhs) + (last block of rhs), one per integer (FLAG_OVERFLOW)
  std::vector<Idx> up, down;  // up[l]: states of level l + 1 from level l; down[l] (l >= 1): carries into level l

  // elements per level: n[0] = blocks, n[l + 1] = ceil(n[l] / 3) until at most TOP are left
  static std::vector<uint32_t> level_sizes(uint32_t L) {
    std::vector<uint32_t> n{L};
    while (n.back() > TOP) n.push_back((n.back() + G - 1) / G);
    return n;
  }
  static uint64_t pool_slots(uint32_t L) {  // per integer
    const auto n = level_sizes(L);
    uint64_t s = L + (n.size() > 1 ? n[1] : 0) + 1;  // M, P, O (overflow preparation of the integer)
    for (uint32_t v : n) s += 2 * (uint64_t)v;   // W, C
    return s;
  }
  // PBS one propagation issues per integer
  static uint64_t pbs_count(uint32_t L) {
    if (L == 1) return 1;
    const auto n = level_sizes(L);
    const size_t top = n.size() - 1;
    uint64_t c = 2 * (uint64_t)L;                     // first bootstraps, results
    for (size_t l = 1; l <= top; ++l) c += n[l] - 1;  // group states (every group but the last of its level)
    if (top >= 1) c += L / G;                         // state of the first two blocks of every full group
    c += n[top] - 1;                                  // carries into the top elements
    for (size_t l = 1; l < top; ++l) c += n[l] - (n[l] + G - 1) / G;  // carries into the groups that are not first in theirs
    return c;
  }

  Idx make(hipStream_t st, const std::vector<uint64_t> &a, const std::vector<uint64_t> &b,
           const std::vector<uint64_t> &o, const std::vector<uint64_t> &l) {
    auto up_ = [&](const std::vector<uint64_t> &h) {
      uint64_t *d = dev_upload(st, h);
      if (d) dev_arrays.push_back(d);
      return d;
    };
    Idx r;
    r.a = up_(a);
    r.b = up_(b);
    r.o = up_(o);
    r.lut = up_(l);
    r.count = (uint32_t)l.size();
    return r;
  }

  void build_indexes(hipStream_t st, uint32_t cts) {
    if (!dev_arrays.empty()) HX_CHECK(hipStreamSynchronize(st));  // launches of the previous shape may still read them
    for (auto *d : dev_arrays) scratch_free(d);
    dev_arrays.clear();
    up.clear();
    down.clear();
    const uint32_t L = blocks;
    const auto n = level_sizes(L);
    const size_t top = n.size() - 1;
    // pool slots: W[l] at wbase[l] + c n[l] + e, C[l] likewise; M at mbase + c L + j; P at pbase + c n[1] + group
    std::vector<uint64_t> wbase(n.size()), cbase(n.size());
    uint64_t next = 0;
    for (size_t l = 0; l <= top; ++l) wbase[l] = next, next += (uint64_t)cts * n[l];
    for (size_t l = 0; l <= top; ++l) cbase[l] = next, next += (uint64_t)cts * n[l];
    const uint64_t mbase = next;
    next += (uint64_t)cts * L;
    const uint64_t pbase = next;
    next += (uint64_t)cts * (n.size() > 1 ? n[1] : 0);
    const uint64_t obase = next;
    auto O = [&](uint32_t c) { return obase + c; };
    auto W = [&](size_t l, uint32_t c, uint32_t e) { return wbase[l] + (uint64_t)c * n[l] + e; };
    auto M = [&](uint32_t c, uint32_t j) { return mbase + (uint64_t)c * L + j; };
    auto P = [&](uint32_t c, uint32_t g) { return pbase + (uint64_t)c * n[1] + g; };
    auto T = [&](uint32_t c, uint32_t j) { return (uint64_t)c * L + j; };
    auto shift = [&](size_t l, uint32_t e) { return l == top ? e : e % G; };
    // pool slot of the carry into element e of level l (the slot of its group's carry for the first element of
    // a group); e = 0 of an integer receives none
    std::function<int64_t(size_t, uint32_t, uint32_t)> carry_of = [&](size_t l, uint32_t c, uint32_t e) -> int64_t {
      if (e == 0) return -1;
      if (l < top && e % G == 0) return carry_of(l + 1, c, e / G);
      return (int64_t)(cbase[l] + (uint64_t)c * n[l] + e);
    };
    std::vector<uint64_t> a, b, o, lut;
    auto reset = [&]() { a.clear(), b.clear(), o.clear(), lut.clear(); };
    // A (two functions per bootstrap): v[t] -> W[0][t], M[t]; out_idx = [function 0 of every block | function 1]
    for (uint32_t c = 0; c < cts; ++c)
      for (uint32_t j = 0; j < L; ++j) {
        o.push_back(W(0, c, j));
        lut.push_back(LUT_FIRST2 + (j + 1 == L ? 4 : j == 0 ? 3 : shift(0, j)));
      }
    for (uint32_t c = 0; c < cts; ++c)
      for (uint32_t j = 0; j < L; ++j) o.push_back(M(c, j));
    rA = make(st, {}, {}, o, lut);
    // up: U of the children -> shifted state of the parent (CSR over the pool; a = offsets, b = members); with the
    // blocks' round also the state of the first two blocks of every full group of three
    for (size_t l = 0; l < top; ++l) {
      reset();
      a.push_back(0);
      for (uint32_t c = 0; c < cts; ++c) {
        for (uint32_t pp = 0; pp + 1 < n[l + 1]; ++pp) {
          const uint32_t len = std::min(G, n[l] - pp * G);
          for (uint32_t q = 0; q < len; ++q) b.push_back(W(l, c, pp * G + q));
          a.push_back(b.size());
          o.push_back(W(l + 1, c, pp));
          lut.push_back(LUT_GROUP + 3 * (len - 1) + shift(l + 1, pp));
        }
        for (uint32_t pp = 0; l == 0 && pp * G + 2 < L; ++pp) {
          b.push_back(W(0, c, pp * G));
          b.push_back(W(0, c, pp * G + 1));
          a.push_back(b.size());
          o.push_back(P(c, pp));
          lut.push_back(LUT_GROUP + 3 * (2 - 1) + 0);
        }
      }
      up.push_back(make(st, a, b, o, lut));
    }
    // top: carry into element j = bit j of w_0 + .. + w_(j-1)
    reset();
    a.push_back(0);
    for (uint32_t c = 0; c < cts; ++c)
      for (uint32_t j = 1; j < n[top]; ++j) {
        for (uint32_t i = 0; i < j; ++i) b.push_back(W(top, c, i));
        a.push_back(b.size());
        o.push_back((uint64_t)carry_of(top, c, j));
        lut.push_back(LUT_BIT + j);
      }
    rTop = make(st, a, b, o, lut);
    // down (levels >= 1): carry into element e = 3 p + q (q >= 1) = bit q of (carry into p) + w_(3p) + .. + w_(e-1)
    down.resize(top);
    for (size_t l = top; l-- > 1;) {
      reset();
      a.push_back(0);
      for (uint32_t c = 0; c < cts; ++c)
        for (uint32_t e = 0; e < n[l]; ++e) {
          const uint32_t pp = e / G, q = e % G;
          if (q == 0) continue;
          const int64_t cin = carry_of(l + 1, c, pp);
          if (cin >= 0) b.push_back((uint64_t)cin);
          for (uint32_t i = 0; i < q; ++i) b.push_back(W(l, c, pp * G + i));
          a.push_back(b.size());
          o.push_back((uint64_t)carry_of(l, c, e));
          lut.push_back(LUT_BIT + q);
        }
      down[l] = make(st, a, b, o, lut);
    }
    // results: 4 m + z; and the output carry of every integer from 4 (state of the last block) + z
    std::vector<uint64_t> a2{0}, b2, lut2, a3{0}, b3, lut3, o3, l3;
    reset();
    a.push_back(0);
    for (uint32_t c = 0; c < cts; ++c)
      for (uint32_t j = 0; j < L; ++j) {
        const size_t before = b.size();
        bool z_is_bit = true;
        if (top == 0) {
          if (j > 0) b.push_back((uint64_t)carry_of(0, c, j));
        } else {
          const uint32_t pp = j / G, q = j % G;
          const int64_t cin = carry_of(1, c, pp);
          if (cin >= 0) b.push_back((uint64_t)cin);
          if (q == 1) b.push_back(W(0, c, pp * G));
          if (q == 2) b.push_back(P(c, pp));
          z_is_bit = q == 0;
        }
        if (j + 1 == L) {  // the same z next to the last block's state
          b2.insert(b2.end(), b.begin() + (std::ptrdiff_t)before, b.end());
          b2.push_back(W(0, c, j));
          a2.push_back(b2.size());
          lut2.push_back(z_is_bit ? LUT_OUT_BIT : LUT_OUT_Z);
          // ... and next to the overflow preparation
          b3.insert(b3.end(), b.begin() + (std::ptrdiff_t)before, b.end());
          b3.push_back(O(c));
          a3.push_back(b3.size());
          lut3.push_back(z_is_bit ? LUT_OVF_BIT : LUT_OVF_Z);
          o3.push_back(O(c));
          l3.push_back(LUT_OVF_PREP);
        }
        b.push_back(M(c, j));
        a.push_back(b.size());
        lut.push_back(z_is_bit ? LUT_RES_BIT : LUT_RES_Z);
      }
    rRes = make(st, a, b, {}, lut);
    rOut = make(st, a2, b2, {}, lut2);
    rOvf = make(st, a3, b3, {}, lut3);
    rOvfPrep = make(st, {}, {}, o3, l3);
    // optional input carry (added to block 0 of every integer); single-block integers: message and carry of v
    reset();
    std::vector<uint64_t> l1;
    for (uint32_t c = 0; c < cts; ++c) {
      a.push_back(T(c, 0));
      lut.push_back(LUT_CARRY);
      l1.push_back(LUT_MSG);
    }
    rIO = make(st, a, {}, {}, lut);
    rOne = make(st, {}, {}, {}, l1);
    {  // last block of every integer (operand views); single-block integers: the preparation alone as a "group"
      std::vector<uint64_t> last, a1{0}, b1, lo;
      for (uint32_t c = 0; c < cts; ++c) {
        last.push_back(T(c, L - 1));
        b1.push_back(O(c));
        a1.push_back(b1.size());
        lo.push_back(LUT_OVF_BIT);
      }
      rLast = make(st, last, {}, {}, {});
      rOne1 = make(st, a1, b1, {}, lo);
    }
    cached_cts = cts;
  }

  void init(const CudaStreamsFFI &ss, const Params &p, uint32_t num_blocks, uint32_t cts) {
    blocks = num_blocks;
    max_cts = cts;
    // 4 bits per block: the binary additions of three shifted states; an accumulator of two functions for values up
    // to 2 msg - 1; results packed as 4 (msg - 1) + 3
    HX_PANIC_IF_FALSE(p.msg * p.carry >= 16 && p.carry >= p.msg && 4 * p.msg <= p.msg * p.carry,
                      "carry propagation needs at least 4 bits per block (message_modulus * carry_modulus >= 16)");
    // MESSAGE_1_CARRY_3-class sets (msg = 2) are not covered by any test of this layer: refused, not guessed
    HX_PANIC_IF_FALSE(p.msg >= 3, "carry propagation: message_modulus %u < 3 is not supported", p.msg);
    const uint64_t m = p.msg;
    using F = std::function<uint64_t(uint64_t)>;
    const size_t lw = (size_t)(p.k + 1) * p.N;
    std::vector<std::vector<uint64_t>> luts(LUT_PROP_COUNT, std::vector<uint64_t>(lw, 0));
    auto one = [&](uint64_t id, const F &f) { generate_lut(p, luts[id].data(), f); };
    const F low4 = [m](uint64_t x) -> uint64_t { return 4 * (x % m); };
    for (uint64_t q = 0; q < G; ++q)
      generate_many_lut(p, luts[LUT_FIRST2 + q].data(),
                        {[m, q](uint64_t x) -> uint64_t { return (x >= m ? 2 : (x == m - 1 ? 1 : 0)) << q; }, low4});
    generate_many_lut(p, luts[LUT_FIRST2 + 3].data(), {[m](uint64_t x) -> uint64_t { return x >= m ? 2 : 0; }, low4});
    generate_many_lut(p, luts[LUT_FIRST2 + 4].data(),
                      {[m](uint64_t x) -> uint64_t { return 4 * (x >= m ? 2 : (x == m - 1 ? 1 : 0)); }, low4});
    for (uint64_t len = 1; len <= G; ++len)
      for (uint64_t q = 0; q < G; ++q)
        one(LUT_GROUP + 3 * (len - 1) + q, [len, q](uint64_t u) -> uint64_t {
          const uint64_t h0 = (u >> len) & 1, h1 = ((u + 1) >> len) & 1;  // a carry leaves without / with one coming in
          return (h0 ? 2 : (h1 ? 1 : 0)) << q;
        });
    for (uint64_t q = 1; q <= G; ++q) one(LUT_BIT + q, [q](uint64_t x) -> uint64_t { return (x >> q) & 1; });
    one(LUT_RES_BIT, [m](uint64_t x) -> uint64_t { return ((x >> 2) + (x & 3)) % m; });
    one(LUT_RES_Z, [m](uint64_t x) -> uint64_t { return ((x >> 2) + ((x & 3) >= 2)) % m; });
    // state 2 generated, 1 propagated: a carry leaves if it is generated, or propagated and one arrives
    one(LUT_OUT_BIT, [](uint64_t x) -> uint64_t { return ((x >> 2) + (x & 3)) >= 2; });
    one(LUT_OUT_Z, [](uint64_t x) -> uint64_t { return ((x >> 2) + ((x & 3) >= 2)) >= 2; });
    one(LUT_MSG, [m](uint64_t x) -> uint64_t { return x % m; });
    one(LUT_CARRY, [m](uint64_t x) -> uint64_t { return x / m; });
    {
      uint32_t bits = 0;
      while ((1ull << bits) < m) ++bits;
      // integer_utilities.h:2326-2352 (f_overflow_fp): the carry into the sign bit against the carry out of the block
      one(LUT_OVF_PREP, [m, bits](uint64_t x) -> uint64_t {
        const uint64_t lhs = x / m, rhs = x % m, mask = (1ull << (bits - 1)) - 1;
        uint64_t r = 0;
        for (uint64_t cin = 0; cin < 2; ++cin) {
          const uint64_t out_c = ((lhs + rhs + cin) >> bits) & 1;
          const uint64_t in_c = (((lhs & mask) + (rhs & mask) + cin) >> (bits - 1)) & 1;
          r |= (uint64_t)(in_c != out_c) << (2 + cin);
        }
        return r;
      });
      one(LUT_OVF_BIT, [](uint64_t x) -> uint64_t { return (x & 3) ? (x >> 3) & 1 : (x >> 2) & 1; });
      one(LUT_OVF_Z, [](uint64_t x) -> uint64_t { return (x & 3) >= 2 ? (x >> 3) & 1 : (x >> 2) & 1; });
    }
    const uint32_t T = cts * num_blocks;
    drv.init(ss, p, std::min<uint32_t>(T, 1u << 16), luts, 2);
    const size_t w = p.big_n + 1;
    radix_alloc((void **)&d_pool, (size_t)cts * pool_slots(num_blocks) * w * sizeof(uint64_t));
    radix_alloc((void **)&d_p, (size_t)T * w * sizeof(uint64_t));
    radix_alloc((void **)&d_pack, (size_t)cts * w * sizeof(uint64_t));
    if (!t_dry) build_indexes(S0(ss), cts);  // a call with fewer integers rebuilds them
  }

  // dense sums of a round's CSR groups, then one KS -> PBS round on them
  void summed_round(const CudaStreamsFFI &ss, uint64_t *out, const Idx &r, uint32_t w, void *const *ksks,
                    void *const *bsks) {
    if (r.count == 0) return;
    HX_LAUNCH(lwe_group_sum_kernel, dim3(r.count), dim3(256), 0, S0(ss), d_p, d_pool, r.a, r.b, w, r.count);
    drv.round(ss, out, r.o, d_p, nullptr, r.lut, r.count, ksks, bsks);
  }

  // in place on v (cts integers of `blocks` blocks).  carry_in (one block per integer, value 0/1) is added to
  // block 0 first — a first block of value <= 2 msg - 1 still emits at most one carry and receives none;
  // carry_out (one block per integer) receives the carry leaving the last block.  Either may be null.
  // overflow_out (one block per integer, add_and_propagate only): the signed-overflow flag of the addition whose last
  // operand blocks the caller packed into d_pack before adding (pack_last_blocks).
  void pack_last_blocks(const CudaStreamsFFI &ss, const uint64_t *lhs, const uint64_t *rhs, uint32_t cts) {
    const Params &p = drv.p;
    const uint32_t w = p.big_n + 1;
    // block (c + 1) * blocks - 1 of both operands: strided views, one launch per integer batch through index arrays
    if (cached_cts != cts) build_indexes(S0(ss), cts);
    axpy(S0(ss), d_pack, nullptr, lhs, rLast.a, p.msg, rhs, rLast.a, w, cts);
  }
  void run(const CudaStreamsFFI &ss, uint64_t *v, uint32_t cts, void *const *ksks, void *const *bsks,
           const uint64_t *carry_in = nullptr, uint64_t *carry_out = nullptr, uint64_t *overflow_out = nullptr,
           bool results = true) {  // results = false: only the flags are wanted (comparisons), v is left with its sums
    const hipStream_t st = S0(ss);  // linear operations and index uploads: first GPU only, like the reference
    HX_PANIC_IF_FALSE(cts >= 1 && cts <= max_cts, "carry propagation: %u integers exceed the scratch capacity %u", cts,
                      max_cts);
    if (cached_cts != cts) build_indexes(st, cts);
    const Params &p = drv.p;
    const uint32_t w = p.big_n + 1, T = cts * blocks;
    const size_t top = up.size();
    HX_RANGE("carry propagation: %u integers of %u blocks", cts, blocks);
    if (carry_in) axpy(st, v, rIO.a, v, rIO.a, 1, carry_in, nullptr, w, cts);
    if (overflow_out)  // 8 o1 + 4 o0 of every integer, from the operands' last blocks
      drv.round(ss, d_pool, rOvfPrep.o, d_pack, nullptr, rOvfPrep.lut, cts, ksks, bsks);
    if (blocks == 1) {  // nothing to propagate: the carry leaves the integer
      if (overflow_out) {  // the carry into the only block is the input carry itself
        HX_LAUNCH(lwe_group_sum_kernel, dim3(cts), dim3(256), 0, st, d_p, d_pool, rOne1.a, rOne1.b, w, cts);
        if (carry_in) axpy(st, d_p, nullptr, d_p, nullptr, 1, carry_in, nullptr, w, cts);
        drv.round(ss, overflow_out, nullptr, d_p, nullptr, rOne1.lut, cts, ksks, bsks);
      }
      if (carry_out) drv.round(ss, carry_out, nullptr, v, nullptr, rIO.lut, cts, ksks, bsks);
      if (results) drv.round(ss, v, nullptr, v, nullptr, rOne.lut, cts, ksks, bsks);
      return;
    }
    // A: shifted state and 4 (value % msg) of every block; up: shifted states of the groups, level by level
    drv.round(ss, d_pool, rA.o, v, nullptr, rA.lut, T, ksks, bsks, 2, many_lut_stride(p, 2));
    for (const Idx &r : up) summed_round(ss, d_pool, r, w, ksks, bsks);
    // carries: top level, then down to the groups of three blocks
    summed_round(ss, d_pool, rTop, w, ksks, bsks);
    for (size_t l = top; l-- > 1;) summed_round(ss, d_pool, down[l], w, ksks, bsks);
    // results (and the output carry) from 4 m + z
    if (carry_out) summed_round(ss, carry_out, rOut, w, ksks, bsks);
    if (overflow_out) summed_round(ss, overflow_out, rOvf, w, ksks, bsks);
    if (results) summed_round(ss, v, rRes, w, ksks, bsks);
  }

  void release(const CudaStreamsFFI &ss) {
    drv.release(ss);
    for (auto *d : dev_arrays) scratch_free(d);
    dev_arrays.clear();
    for (uint64_t *d : {d_pool, d_p, d_pack})
      if (d) scratch_free(d);
    magic = 0;
  }
};

// ------------------------------------------------------------------ multiplication
// radix_parallel/mul.rs: block products (low / high halves through bivariate LUTs), column sums in
// groups that fit the carry space, final carry propagation.
struct MulMem {
  static constexpr uint32_t kMagic = 0x4D554C31;  // "MUL1"
  uint32_t magic = kMagic;
  bool size_only = false;
  LutDriver drv;       // LUTs: 0 product low, 1 product high, 2 message, 3 carry
  PropagateMem prop;
  uint32_t blocks = 0, max_cts = 0, sub = 0;  // sub = integers per pass
  uint32_t slots = 0;                          // pool slots per integer
  uint64_t *d_pool = nullptr, *d_pack = nullptr, *d_sum = nullptr;

  struct Step {  // one reduction step, ciphertext-relative
    std::vector<uint64_t> offsets, members;      // CSR of groups (pool slots)
    std::vector<uint64_t> msg_slot, carry_slot;  // output slots per group (carry_slot = ~0 when dropped)
  };
  std::vector<uint64_t> prod_lhs, prod_rhs, prod_slot, prod_lut;  // products
  std::vector<Step> steps;
  std::vector<std::vector<uint64_t>> final_cols;  // what is left of every column: degrees adding up to 2 msg - 2 at most

  // The column sums are planned on the largest value every term can take (its degree): a low half of a block
  // product is at most msg - 1, a high half at most (msg - 1)^2 / msg (2 for msg = 4), the carry of a sum S at most
  // S / msg.  A group holds at most max_terms = (msg carry - 1) / (msg - 1) = 5 terms — every term is a fresh
  // bootstrap output, and that count is the noise level the parameter sets are made for (shortint MaxNoiseLevel;
  // it is also the reference's fixed chunk in sum_ciphertexts) — whose degrees add up to msg * carry - 1 at most;
  // a group whose degrees stay below msg emits no carry, and a column is finished once it has at most max_terms
  // terms whose degrees add up to 2 msg - 2, what one carry propagation accepts.
  void plan() {
    const uint32_t L = blocks, m = drv.p.msg;
    const uint32_t cap = drv.p.msg * drv.p.carry - 1, final_cap = 2 * m - 2, max_terms = cap / (m - 1);
    std::vector<std::vector<uint64_t>> cols(L);
    std::vector<uint32_t> deg;  // by pool slot
    uint64_t next = 0;
    auto new_slot = [&](uint32_t d) {
      deg.push_back(d);
      return next++;
    };
    for (uint32_t i = 0; i < L; ++i)
      for (uint32_t j = 0; i + j < L; ++j) {
        prod_lhs.push_back(j);
        prod_rhs.push_back(i);
        prod_lut.push_back(0);
        prod_slot.push_back(next);
        cols[i + j].push_back(new_slot(m - 1));
        if (i + j + 1 < L) {
          prod_lhs.push_back(j);
          prod_rhs.push_back(i);
          prod_lut.push_back(1);
          prod_slot.push_back(next);
          cols[i + j + 1].push_back(new_slot((m - 1) * (m - 1) / m));
        }
      }
    auto total = [&](const std::vector<uint64_t> &c) {
      uint32_t t = 0;
      for (uint64_t x : c) t += deg[x];
      return t;
    };
    auto finished = [&](const std::vector<uint64_t> &c) { return total(c) <= final_cap && c.size() <= max_terms; };
    auto unfinished = [&]() {
      for (auto &c : cols)
        if (!finished(c)) return true;
      return false;
    };
    // Many integers per call (throughput): a group costs two PBS whatever it holds, so only well-filled groups
    // (max_terms terms, or four with degrees adding up to cap - 2 at least) are summed while any column can form one —
    // what is left of a column waits for the next step (1,762 PBS per 32-block multiplication, three more but small
    // rounds).  Few integers (latency): every term is grouped at once, which needs the fewest rounds (1,804).
    const bool wide = max_cts >= 8;
    while (unfinished()) {
      // first-fit decreasing: the terms of an unfinished column, largest degree first, into groups of capacity cap
      std::vector<std::vector<std::vector<uint64_t>>> groups(L);
      bool any_good = false;
      auto good = [&](const std::vector<uint64_t> &g) {
        return g.size() >= max_terms || (g.size() >= 4 && total(g) + 2 >= cap);
      };
      for (uint32_t c = 0; c < L; ++c) {
        if (finished(cols[c])) continue;
        std::vector<uint64_t> sorted = cols[c];
        std::stable_sort(sorted.begin(), sorted.end(), [&](uint64_t x, uint64_t y) { return deg[x] > deg[y]; });
        for (uint64_t x : sorted) {
          bool placed = false;
          for (auto &g : groups[c])
            if (g.size() < max_terms && total(g) + deg[x] <= cap) {
              g.push_back(x);
              placed = true;
              break;
            }
          if (!placed) groups[c].push_back({x});
        }
        for (auto &g : groups[c]) any_good = any_good || good(g);
      }
      Step s;
      s.offsets.push_back(0);
      std::vector<std::vector<uint64_t>> nc(L);
      for (uint32_t c = 0; c < L; ++c) {
        if (groups[c].empty()) {  // finished column: stays as it is
          nc[c].insert(nc[c].end(), cols[c].begin(), cols[c].end());
          continue;
        }
        for (auto &g : groups[c]) {
          const bool emit = g.size() >= 2 && (!wide || (any_good ? good(g) : g.size() >= 3));
          if (!emit) {
            nc[c].insert(nc[c].end(), g.begin(), g.end());
            continue;
          }
          const uint32_t sum = total(g);
          s.members.insert(s.members.end(), g.begin(), g.end());
          s.offsets.push_back(s.members.size());
          const uint64_t ms = new_slot(std::min(m - 1, sum));
          s.msg_slot.push_back(ms);
          nc[c].push_back(ms);
          if (c + 1 < L && sum >= m) {
            const uint64_t cs = new_slot(sum / m);
            s.carry_slot.push_back(cs);
            nc[c + 1].push_back(cs);
          } else {
            s.carry_slot.push_back(~(uint64_t)0);
          }
        }
      }
      HX_PANIC_IF_FALSE(!s.msg_slot.empty(), "multiplication plan: no progress");
      cols.swap(nc);
      steps.push_back(std::move(s));
    }
    final_cols = cols;
    slots = (uint32_t)next;
  }

  void init(const CudaStreamsFFI &ss, const Params &p, uint32_t num_blocks, uint32_t cts) {
    blocks = num_blocks;
    max_cts = cts;
    const uint64_t m = p.msg;
    std::vector<std::function<uint64_t(uint64_t)>> fs = {
        [m](uint64_t x) -> uint64_t { return ((x / m) * (x % m)) % m; },
        [m](uint64_t x) -> uint64_t { return ((x / m) * (x % m)) / m; },
        [m](uint64_t x) -> uint64_t { return x % m; },
        [m](uint64_t x) -> uint64_t { return x / m; },
    };
    std::vector<std::vector<uint64_t>> luts;
    for (auto &f : fs) {
      luts.emplace_back((size_t)(p.k + 1) * p.N);
      generate_lut(p, luts.back().data(), f);
    }
    drv.p = p;
    plan();
    // integers per pass: the term pool of one pass may take 12 GiB of the 288 GB (32-block integers: 37 MB
    // each, ~340 per pass) — the later reduction rounds shrink fast, so passes should be as wide as possible
    const size_t w = p.big_n + 1;
    const size_t per_ct = (size_t)slots * w * sizeof(uint64_t);
    sub = (uint32_t)std::max<size_t>(1, std::min<size_t>(cts, ((size_t)12 << 30) / per_ct));
    drv.init(ss, p, 1u << 16, luts);
    radix_alloc((void **)&d_pool, (size_t)sub * per_ct);
    const size_t n_prod = prod_slot.size();
    size_t max_groups = 0;
    for (auto &s : steps) max_groups = std::max(max_groups, s.msg_slot.size());
    radix_alloc((void **)&d_pack, (size_t)sub * n_prod * w * sizeof(uint64_t));
    radix_alloc((void **)&d_sum, std::max<size_t>(1, (size_t)sub * max_groups) * w * sizeof(uint64_t));
    prop.init(ss, p, num_blocks, sub);
    if (!t_dry) {
      build_pass(S0(ss), std::min(sub, cts));
      if (std::min(sub, cts) != sub) prop.build_indexes(S0(ss), std::min(sub, cts));
    }
  }

  // device index arrays of one pass over nb integers (built when the scratch is created, rebuilt if a call
  // brings another count): block products, every column-sum step, the final additions
  struct PassIdx {
    uint32_t nb = 0;
    std::vector<uint64_t *> owned;
    uint64_t *pa = nullptr, *pb = nullptr, *po = nullptr, *pl = nullptr;
    struct StepIdx {
      uint64_t *off = nullptr, *mem = nullptr, *in = nullptr, *out = nullptr, *lut = nullptr;
      uint32_t groups = 0, count = 0;
    };
    std::vector<StepIdx> steps;
    uint64_t *foff = nullptr, *fmem = nullptr;
  } pass;

  void free_pass() {
    for (auto *d : pass.owned) scratch_free(d);
    pass = PassIdx();
  }

  void build_pass(hipStream_t st, uint32_t nb) {
    if (!pass.owned.empty()) HX_CHECK(hipStreamSynchronize(st));  // launches of the previous shape may still read them
    free_pass();
    pass.nb = nb;
    const uint32_t L = blocks;
    const size_t n_prod = prod_slot.size();
    auto up = [&](const std::vector<uint64_t> &h) {
      uint64_t *d = dev_upload(st, h);
      if (d) pass.owned.push_back(d);
      return d;
    };
    {  // block products
      std::vector<uint64_t> a(nb * n_prod), b(nb * n_prod), o(nb * n_prod), l(nb * n_prod);
      for (uint32_t c = 0; c < nb; ++c)
        for (size_t q = 0; q < n_prod; ++q) {
          a[c * n_prod + q] = (uint64_t)c * L + prod_lhs[q];
          b[c * n_prod + q] = (uint64_t)c * L + prod_rhs[q];
          o[c * n_prod + q] = (uint64_t)c * slots + prod_slot[q];
          l[c * n_prod + q] = prod_lut[q];
        }
      pass.pa = up(a), pass.pb = up(b), pass.po = up(o), pass.pl = up(l);
    }
    for (const Step &s : steps) {  // column sums
      PassIdx::StepIdx si;
      const size_t G = s.msg_slot.size();
      if (G != 0) {
        std::vector<uint64_t> off(nb * G + 1), mem(nb * s.members.size());
        std::vector<uint64_t> in, out, lut;
        for (uint32_t c = 0; c < nb; ++c) {
          for (size_t g = 0; g < G; ++g) {
            off[c * G + g] = c * s.members.size() + s.offsets[g];
            in.push_back(c * G + g);
            out.push_back((uint64_t)c * slots + s.msg_slot[g]);
            lut.push_back(2);
            if (s.carry_slot[g] != ~(uint64_t)0) {
              in.push_back(c * G + g);
              out.push_back((uint64_t)c * slots + s.carry_slot[g]);
              lut.push_back(3);
            }
          }
          for (size_t m = 0; m < s.members.size(); ++m) mem[c * s.members.size() + m] = (uint64_t)c * slots + s.members[m];
        }
        off[nb * G] = nb * s.members.size();
        si.off = up(off), si.mem = up(mem), si.in = up(in), si.out = up(out), si.lut = up(lut);
        si.groups = (uint32_t)(nb * G);
        si.count = (uint32_t)in.size();
      }
      pass.steps.push_back(si);
    }
    {  // what is left of every column (degrees adding up to 2 msg - 2 at most): CSR over the pool, in lhs order
      std::vector<uint64_t> off{0}, mem;
      for (uint32_t c = 0; c < nb; ++c)
        for (uint32_t col = 0; col < L; ++col) {
          for (uint64_t x : final_cols[col]) mem.push_back((uint64_t)c * slots + x);
          off.push_back(mem.size());
        }
      pass.foff = up(off), pass.fmem = up(mem);
    }
  }

  // lhs <- lhs * rhs for `cts` integers
  void run(const CudaStreamsFFI &ss, uint64_t *lhs, const uint64_t *rhs, uint32_t cts, void *const *ksks,
           void *const *bsks) {
    const hipStream_t st = S0(ss);
    HX_PANIC_IF_FALSE(cts >= 1 && cts <= max_cts, "multiplication: %u integers exceed the scratch capacity %u", cts,
                      max_cts);
    const Params &p = drv.p;
    const uint32_t L = blocks, w = p.big_n + 1;
    const size_t n_prod = prod_slot.size();
    for (uint32_t c0 = 0; c0 < cts; c0 += sub) {
      const uint32_t nb = std::min(sub, cts - c0);
      uint64_t *l0 = lhs + (size_t)c0 * L * w;
      const uint64_t *r0 = rhs + (size_t)c0 * L * w;
      if (pass.nb != nb) {
        HX_CHECK(hipStreamSynchronize(st));  // the previous pass may still read its arrays
        build_pass(st, nb);
      }
      // block products
      HX_RANGE("integer mul: %u integers of %u blocks", nb, L);
      {
        HX_RANGE("block products");
        axpy(st, d_pack, nullptr, l0, pass.pa, p.msg, r0, pass.pb, w, (uint32_t)(nb * n_prod));
        drv.round(ss, d_pool, pass.po, d_pack, nullptr, pass.pl, (uint32_t)(nb * n_prod), ksks, bsks);
      }
      for (const auto &si : pass.steps) {  // column sums
        if (si.groups == 0) continue;
        HX_RANGE("column sums (%u groups)", si.groups);
        HX_LAUNCH(lwe_group_sum_kernel, dim3(si.groups), dim3(256), 0, st, d_sum, d_pool, si.off, si.mem, w, si.groups);
        drv.round(ss, d_pool, si.out, d_sum, si.in, si.lut, si.count, ksks, bsks);
      }
      // the remaining terms of every column added into lhs, then one carry propagation
      HX_LAUNCH(lwe_group_sum_kernel, dim3(nb * L), dim3(256), 0, st, l0, d_pool, pass.foff, pass.fmem, w, nb * L);
      prop.run(ss, l0, nb, ksks, bsks);
    }
  }

  void release(const CudaStreamsFFI &ss) {
    HX_CHECK(hipStreamSynchronize(S0(ss)));
    free_pass();
    drv.release(ss);
    prop.release(ss);
    if (d_pool) scratch_free(d_pool);
    if (d_pack) scratch_free(d_pack);
    if (d_sum) scratch_free(d_sum);
    magic = 0;
  }
};

// Multiplication by an encrypted boolean (integer.h:173-177 is_boolean_left / is_boolean_right; multiplication.h:28-49,
// cmux.cuh:13-46 zero_out_if): every block of the integer packed with the boolean's single block, one bivariate
// table (condition == 0 ? 0 : block): one KS -> PBS round of `blocks` bootstraps per integer.
struct BoolMulMem {
  static constexpr uint32_t kMagic = 0x424D554C;  // "BMUL"
  uint32_t magic = kMagic;
  bool size_only = false;
  LutDriver drv;
  uint32_t blocks = 0, max_cts = 0;
  uint64_t *d_pack = nullptr, *d_cond_idx = nullptr, *d_lut_idx = nullptr;

  void init(const CudaStreamsFFI &ss, const Params &p, uint32_t num_blocks, uint32_t cts) {
    blocks = num_blocks;
    max_cts = cts;
    const uint64_t m = p.msg;
    std::vector<std::vector<uint64_t>> luts(1, std::vector<uint64_t>((size_t)(p.k + 1) * p.N));
    generate_lut(p, luts[0].data(), [m](uint64_t x) -> uint64_t { return (x % m) == 0 ? 0 : x / m; });
    const uint32_t T = cts * num_blocks;
    drv.init(ss, p, std::min<uint32_t>(T, 1u << 16), luts);
    radix_alloc((void **)&d_pack, (size_t)T * (p.big_n + 1) * sizeof(uint64_t));
    if (!t_dry) {
      std::vector<uint64_t> ci(T), li(T, 0);
      for (uint32_t i = 0; i < T; ++i) ci[i] = i / num_blocks;  // the boolean of integer c is block c of its operand
      d_cond_idx = dev_upload(S0(ss), ci);
      d_lut_idx = dev_upload(S0(ss), li);
    }
  }
  // out <- condition ? value : 0; `cond` holds one block per integer
  void run(const CudaStreamsFFI &ss, uint64_t *out, const uint64_t *value, const uint64_t *cond, uint32_t cts,
           void *const *ksks, void *const *bsks) {
    HX_PANIC_IF_FALSE(cts >= 1 && cts <= max_cts, "boolean multiplication: %u integers exceed the scratch capacity %u",
                      cts, max_cts);
    const Params &p = drv.p;
    const uint32_t T = cts * blocks;
    axpy(S0(ss), d_pack, nullptr, value, nullptr, p.msg, cond, d_cond_idx, p.big_n + 1, T);
    drv.round(ss, out, nullptr, d_pack, nullptr, d_lut_idx, T, ksks, bsks);
  }
  void release(const CudaStreamsFFI &ss) {
    HX_CHECK(hipStreamSynchronize(S0(ss)));
    drv.release(ss);
    for (uint64_t *d : {d_pack, d_cond_idx, d_lut_idx})
      if (d) scratch_free(d);
    magic = 0;
  }
};

// ------------------------------------------------------------------ bitwise operations, subtraction, full propagation
// bitwise_ops.h:101-178 (int_bitop_buffer): one bivariate table of (lhs, rhs) packed msg * lhs + rhs for the ciphertext
// forms, one univariate table per clear block value for the scalar forms (the clear blocks are the LUT indexes)
struct BitopMem {
  static constexpr uint32_t kMagic = 0x42495431;  // "BIT1"
  uint32_t magic = kMagic;
  bool size_only = false;
  LutDriver drv;
  uint32_t op = 0;
  uint64_t *d_pack = nullptr, *d_lut_idx = nullptr;  // packed operands; all-zero LUT indexes (ciphertext forms)
};
static uint64_t bitop_apply(uint32_t op, uint64_t a, uint64_t b) {
  switch (op % 3) {
    case 0: return a & b;
    case 1: return a | b;
    default: return a ^ b;
  }
}
// bitwise_ops.cu:133-185 (update_degrees_after_*): the largest value the operation can return on operands up to (a, b)
static uint64_t bitop_degree(uint32_t op, uint64_t a, uint64_t b) {
  const uint64_t hi = std::max(a, b), lo = std::min(a, b);
  if (op % 3 == 0) return lo;
  uint64_t r = hi;
  for (uint64_t j = 0; j <= lo; ++j) r = std::max(r, op % 3 == 1 ? (hi | j) : (hi ^ j));
  return r;
}

// subtraction.cuh:31-47: the negation of rhs with its correcting term, then the addition's carry propagation
struct SubMem {
  static constexpr uint32_t kMagic = 0x53554231;  // "SUB1"
  uint32_t magic = kMagic;
  PropagateMem prop;
  uint64_t *d_neg = nullptr;
};

// integer.cuh:1924-1983 (host_full_propagate_inplace): block after block, message and carry of the block by one
// two-input bootstrap round, the carry added to the next block
struct FullPropMem {
  static constexpr uint32_t kMagic = 0x46505031;  // "FPP1"
  uint32_t magic = kMagic;
  bool size_only = false;
  LutDriver drv;  // LUTs: 0 message, 1 carry
  uint64_t *d_two = nullptr, *d_lut_idx = nullptr;  // the block twice -> (message, carry); indexes {0, 1}
};

// comparison.cuh / integer.h:246-276, unsigned operands.  Orderings ride on the subtraction's carry tree:
//   a >= b  <=>  a + (2^bits - b) carries out of the last block
// so GE is the output carry of the subtraction WITHOUT its result round (65 of the 97 bootstraps of a 32-block
// subtraction, 5 rounds), LE swaps the operands, LT / GT are 1 - that (levelled).  The reference reduces block orderings
// pairwise (63 bootstraps in 6 rounds for 32 blocks: comparison.cuh).  Equality: one bivariate round, then sums of up to
// (msg * carry - 1) / (msg - 1) = 5 block results against their count, level by level (32 + 7 + 2 + 1 bootstraps in 4 rounds).
struct CompareMem {
  static constexpr uint32_t kMagic = 0x434D5031;  // "CMP1"
  uint32_t magic = kMagic;
  bool size_only = false;
  uint32_t op = 0, blocks = 0;
  PropagateMem prop;  // orderings
  LutDriver eq;       // equality: LUT 0 a == b, 1 a != b (packed pairs); 2 + (c - 1): sum == c; 2 + G + (c - 1): sum != c
  uint32_t G = 0;     // largest group: (msg * carry - 1) / (msg - 1)
  uint64_t *d_tmp = nullptr, *d_neg = nullptr, *d_bool = nullptr;  // orderings: lhs copy, negated rhs, the flag
  uint64_t *d_pack = nullptr, *d_pool = nullptr, *d_sum = nullptr, *d_lut0 = nullptr;
  struct Level {
    uint32_t groups = 0, in_off = 0, out_off = 0;
    uint64_t *off = nullptr, *mem = nullptr, *lut = nullptr;
  };
  std::vector<Level> levels;
  std::vector<uint64_t *> dev_arrays;
  uint32_t last_slot = 0;

  bool ordering() const { return op >= GT && op <= LE; }

  void init(const CudaStreamsFFI &ss, const Params &p, uint32_t L, uint32_t operation) {
    op = operation;
    blocks = L;
    const size_t w = (size_t)p.big_n + 1, lw = (size_t)(p.k + 1) * p.N;
    if (ordering()) {
      prop.init(ss, p, L, 1);
      radix_alloc((void **)&d_tmp, L * w * sizeof(uint64_t));
      radix_alloc((void **)&d_neg, L * w * sizeof(uint64_t));
      radix_alloc((void **)&d_bool, w * sizeof(uint64_t));
      return;
    }
    // comparison.cuh are_all_comparisons_block_true: chunks of (msg * carry - 1) / (msg - 1) block results — 5 at 2_2, which is also
    // what tfhe-rs's noise-level bookkeeping allows for a sum of fresh bootstrap outputs (levels add, MaxNoiseLevel 5).  A sum of
    // 15 fits the plaintext space, but no caller of the reference ever bootstraps one
    G = (p.msg * p.carry - 1) / (p.msg - 1);
    const uint32_t msg = p.msg;
    std::vector<std::vector<uint64_t>> luts(2 + 2 * G, std::vector<uint64_t>(lw));
    generate_lut(p, luts[0].data(), [msg](uint64_t x) -> uint64_t { return x / msg == x % msg; });
    generate_lut(p, luts[1].data(), [msg](uint64_t x) -> uint64_t { return x / msg != x % msg; });
    for (uint32_t c = 1; c <= G; ++c) {
      generate_lut(p, luts[2 + c - 1].data(), [c](uint64_t x) -> uint64_t { return x == c; });
      generate_lut(p, luts[2 + G + c - 1].data(), [c](uint64_t x) -> uint64_t { return x != c; });
    }
    eq.init(ss, p, L, luts);
    radix_alloc((void **)&d_pack, L * w * sizeof(uint64_t));
    // pool: the block results, then every level's results
    uint32_t slots = L, n = L;
    std::vector<uint32_t> sizes;
    while (n > 1) {
      n = (n + G - 1) / G;
      sizes.push_back(n);
      slots += n;
    }
    radix_alloc((void **)&d_pool, (size_t)slots * w * sizeof(uint64_t));
    radix_alloc((void **)&d_sum, (size_t)std::max<uint32_t>(1, sizes.empty() ? 1 : sizes[0]) * w * sizeof(uint64_t));
    const bool ne = op == NE;
    d_lut0 = dev_upload(S0(ss), std::vector<uint64_t>(L, (L == 1 && ne) ? 1 : 0));
    dev_arrays.push_back(d_lut0);
    uint32_t in_off = 0, in_n = L, out_off = L;
    for (size_t l = 0; l < sizes.size(); ++l) {
      Level lv;
      lv.groups = sizes[l];
      lv.in_off = in_off;
      lv.out_off = out_off;
      std::vector<uint64_t> off(lv.groups + 1), mem(in_n), lut(lv.groups);
      for (uint32_t g = 0; g <= lv.groups; ++g) off[g] = std::min<uint64_t>((uint64_t)g * G, in_n);
      for (uint32_t i = 0; i < in_n; ++i) mem[i] = in_off + i;
      const bool last = l + 1 == sizes.size();
      for (uint32_t g = 0; g < lv.groups; ++g) {
        const uint32_t c = (uint32_t)(off[g + 1] - off[g]);
        lut[g] = (last && ne) ? 2 + G + c - 1 : 2 + c - 1;
      }
      lv.off = dev_upload(S0(ss), off);
      lv.mem = dev_upload(S0(ss), mem);
      lv.lut = dev_upload(S0(ss), lut);
      for (uint64_t *d : {lv.off, lv.mem, lv.lut}) dev_arrays.push_back(d);
      levels.push_back(lv);
      in_off = out_off;
      in_n = lv.groups;
      out_off += lv.groups;
    }
    last_slot = in_off;  // slot of the single result (0 when L == 1)
  }

  // flag (one block) <- op(a, b); a and b: `blocks` clean blocks each
  void run(const CudaStreamsFFI &ss, uint64_t *flag, const uint64_t *a, const uint64_t *b, void *const *ksks,
           void *const *bsks) {
    const hipStream_t st = S0(ss);
    const uint32_t L = blocks;
    if (ordering()) {
      const Params &p = prop.drv.p;
      const uint32_t w = p.big_n + 1;
      const uint64_t delta = ((uint64_t)1 << 63) / ((uint64_t)p.msg * p.carry);
      const bool swap = op == LE || op == GT;       // LE(a, b) = GE(b, a); GT(a, b) = LT(b, a)
      const bool invert = op == LT || op == GT;     // LT = 1 - GE
      const uint64_t *x = swap ? b : a, *y = swap ? a : b;
      HX_CHECK(hipMemcpyAsync(d_tmp, x, (size_t)L * w * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
      HX_LAUNCH(lwe_negate_const_kernel, dim3(L), dim3(256), 0, st, d_neg, y, w, L, L, (uint64_t)p.msg * delta,
                (uint64_t)(p.msg - 1) * delta);
      axpy(st, d_tmp, nullptr, d_tmp, nullptr, 1, d_neg, nullptr, w, L);
      prop.run(ss, d_tmp, 1, ksks, bsks, nullptr, invert ? d_bool : flag, nullptr, false);
      if (invert) HX_LAUNCH(lwe_negate_const_kernel, dim3(1), dim3(256), 0, st, flag, d_bool, w, 1, 1, delta, delta);
      return;
    }
    const Params &p = eq.p;
    const uint32_t w = p.big_n + 1;
    axpy(st, d_pack, nullptr, a, nullptr, p.msg, b, nullptr, w, L);
    eq.round(ss, d_pool, nullptr, d_pack, nullptr, d_lut0, L, ksks, bsks);
    for (const Level &lv : levels) {
      HX_LAUNCH(lwe_group_sum_kernel, dim3(lv.groups), dim3(256), 0, st, d_sum, d_pool, lv.off, lv.mem, w, lv.groups);
      eq.round(ss, d_pool + (size_t)lv.out_off * w, nullptr, d_sum, nullptr, lv.lut, lv.groups, ksks, bsks);
    }
    HX_CHECK(hipMemcpyAsync(flag, d_pool + (size_t)last_slot * w, (size_t)w * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
  }

  void release(const CudaStreamsFFI &ss) {
    HX_CHECK(hipStreamSynchronize(S0(ss)));
    if (ordering()) prop.release(ss); else eq.release(ss);
    for (uint64_t *d : dev_arrays)
      if (d) scratch_free(d);
    dev_arrays.clear();
    for (uint64_t *d : {d_tmp, d_neg, d_bool, d_pack, d_pool, d_sum})
      if (d) scratch_free(d);
    magic = 0;
  }
};

// cmux.cuh / integer.h:349-365: out <- condition ? true : false.  Both branches packed with the condition's single block
// (block + msg * condition), one round over the 2 L packed blocks keeps a branch's block or zeroes it, the halves are added
// and a message-extraction round returns the sum to nominal noise (the reference's int_cmux_buffer does the same three steps)
struct CmuxMem {
  static constexpr uint32_t kMagic = 0x434D5831;  // "CMX1"
  uint32_t magic = kMagic;
  bool size_only = false;
  LutDriver drv;  // LUTs: 0 keep if the condition is 1, 1 keep if it is 0, 2 message
  uint32_t blocks = 0;
  uint64_t *d_pack = nullptr, *d_zero_idx = nullptr, *d_lut_idx = nullptr, *d_lut2 = nullptr;

  void init(const CudaStreamsFFI &ss, const Params &p, uint32_t L) {
    blocks = L;
    const uint64_t m = p.msg;
    std::vector<std::vector<uint64_t>> luts(3, std::vector<uint64_t>((size_t)(p.k + 1) * p.N));
    generate_lut(p, luts[0].data(), [m](uint64_t x) -> uint64_t { return x >= m && x < 2 * m ? x - m : 0; });
    generate_lut(p, luts[1].data(), [m](uint64_t x) -> uint64_t { return x < m ? x : 0; });
    generate_lut(p, luts[2].data(), [m](uint64_t x) -> uint64_t { return x % m; });
    drv.init(ss, p, 2 * L, luts);
    radix_alloc((void **)&d_pack, (size_t)2 * L * (p.big_n + 1) * sizeof(uint64_t));
    std::vector<uint64_t> li(2 * L, 0);
    for (uint32_t i = L; i < 2 * L; ++i) li[i] = 1;
    d_zero_idx = dev_upload(S0(ss), std::vector<uint64_t>(L, 0));
    d_lut_idx = dev_upload(S0(ss), li);
    d_lut2 = dev_upload(S0(ss), std::vector<uint64_t>(L, 2));
  }
  void run(const CudaStreamsFFI &ss, uint64_t *out, const uint64_t *cond, const uint64_t *t, const uint64_t *f,
           void *const *ksks, void *const *bsks) {
    const Params &p = drv.p;
    const hipStream_t st = S0(ss);
    const uint32_t L = blocks, w = p.big_n + 1;
    axpy(st, d_pack, nullptr, cond, d_zero_idx, p.msg, t, nullptr, w, L);
    axpy(st, d_pack + (size_t)L * w, nullptr, cond, d_zero_idx, p.msg, f, nullptr, w, L);
    drv.round(ss, d_pack, nullptr, d_pack, nullptr, d_lut_idx, 2 * L, ksks, bsks);
    axpy(st, out, nullptr, d_pack, nullptr, 1, d_pack + (size_t)L * w, nullptr, w, L);
    drv.round(ss, out, nullptr, out, nullptr, d_lut2, L, ksks, bsks);
  }
  void release(const CudaStreamsFFI &ss) {
    HX_CHECK(hipStreamSynchronize(S0(ss)));
    drv.release(ss);
    for (uint64_t *d : {d_pack, d_zero_idx, d_lut_idx, d_lut2})
      if (d) scratch_free(d);
    magic = 0;
  }
};

// scalar_shifts.cuh / integer.h:200-228: logical shift of an unsigned integer by a clear amount = a move by whole blocks
// and, when bits remain, ONE bivariate round over (block, its lower / upper neighbour)
struct ScalarShiftMem {
  static constexpr uint32_t kMagic = 0x53484631;  // "SHF1"
  uint32_t magic = kMagic;
  bool size_only = false;
  LutDriver drv;  // LUT r - 1: the shift by r bits inside a block (r = 1 .. bits per block - 1), this scratch's direction
  uint32_t blocks = 0, bits = 0, left = 0;
  uint64_t *d_pack = nullptr, *d_lut_idx = nullptr;  // packed pairs / block copy; one constant index array per r
};

static uint32_t batch_of(const CudaRadixCiphertextFFI *ct, uint32_t blocks, const char *what) {
  HX_PANIC_IF_FALSE(ct != nullptr && ct->ptr != nullptr, "%s: null radix ciphertext", what);
  HX_PANIC_IF_FALSE(blocks != 0 && ct->num_radix_blocks % blocks == 0,
                    "%s: %u blocks is not a whole number of %u-block integers", what, ct->num_radix_blocks, blocks);
  return ct->num_radix_blocks / blocks;
}

}  // namespace radix
}  // namespace tfhe_hip

using namespace tfhe_hip;
using namespace tfhe_hip::radix;

extern "C" {

// ---- cuda/include/integer/integer.h:127-160 ------------------------------------------------
static uint64_t scratch_apply_lut(CudaStreamsFFI streams, int8_t **mem_ptr, void const *input_lut,
                                  CudaLweBootstrapKeyParamsFFI bsk_params, CudaLweKeyswitchKeyParamsFFI ksk_params,
                                  uint32_t count, uint32_t message_modulus, uint32_t carry_modulus, uint32_t num_many_lut,
                                  uint64_t lut_degree, bool allocate_gpu_memory, uint32_t noise_reduction_type) {
  HX_PANIC_IF_FALSE(input_lut != nullptr && mem_ptr != nullptr, "apply_univariate_lut: null pointer");
  HX_PANIC_IF_FALSE(num_many_lut >= 1, "apply_many_univariate_lut: num_many_lut must be at least 1");
  t_dry = !allocate_gpu_memory;
  t_bytes = 0;
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, noise_reduction_type);
  auto *m = new ApplyLutMem();
  const size_t lw = (size_t)(p.k + 1) * p.N;
  std::vector<std::vector<uint64_t>> luts(1);
  luts[0].assign((const uint64_t *)input_lut, (const uint64_t *)input_lut + lw);
  m->drv.init(streams, p, std::max<uint32_t>(1, count), luts);
  m->degree = lut_degree;
  m->num_many_lut = num_many_lut;
  radix_alloc((void **)&m->d_lut_idx, std::max<uint32_t>(1, count) * sizeof(uint64_t));
  if (!t_dry) HX_CHECK(hipMemsetAsync(m->d_lut_idx, 0, std::max<uint32_t>(1, count) * sizeof(uint64_t), S0(streams)));
  m->size_only = t_dry;
  t_dry = false;
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return t_bytes;
}

uint64_t scratch_cuda_apply_univariate_lut_64_async(CudaStreamsFFI streams, int8_t **mem_ptr, void const *input_lut,
                                                    CudaLweBootstrapKeyParamsFFI bsk_params,
                                                    CudaLweKeyswitchKeyParamsFFI ksk_params,
                                                    uint32_t input_lwe_ciphertext_count, uint32_t message_modulus,
                                                    uint32_t carry_modulus, uint64_t lut_degree,
                                                    bool allocate_gpu_memory,
                                                    enum PBS_MS_REDUCTION_T noise_reduction_type) {
  first_gpu(streams);
  return scratch_apply_lut(streams, mem_ptr, input_lut, bsk_params, ksk_params, input_lwe_ciphertext_count,
                           message_modulus, carry_modulus, 1, lut_degree, allocate_gpu_memory,
                           (uint32_t)noise_reduction_type);
}

// input_lut: ONE accumulator holding num_many_lut functions in sub-tables of lut_stride coefficients
// (tfhe/src/shortint/engine/mod.rs:169-254 fill_many_lut_accumulator)
uint64_t scratch_cuda_apply_many_univariate_lut_64_async(CudaStreamsFFI streams, int8_t **mem_ptr, void const *input_lut,
                                                         CudaLweBootstrapKeyParamsFFI bsk_params,
                                                         CudaLweKeyswitchKeyParamsFFI ksk_params,
                                                         uint32_t num_radix_blocks, uint32_t message_modulus,
                                                         uint32_t carry_modulus, uint32_t num_many_lut,
                                                         uint64_t lut_degree, bool allocate_gpu_memory,
                                                         enum PBS_MS_REDUCTION_T noise_reduction_type) {
  first_gpu(streams);
  return scratch_apply_lut(streams, mem_ptr, input_lut, bsk_params, ksk_params, num_radix_blocks, message_modulus,
                           carry_modulus, num_many_lut, lut_degree, allocate_gpu_memory, (uint32_t)noise_reduction_type);
}

void cuda_apply_univariate_lut_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *output_radix_lwe,
                                        CudaRadixCiphertextFFI const *input_radix_lwe, int8_t *mem_ptr,
                                        void *const *ksks, void *const *bsks) {
  first_gpu(streams);
  auto *m = reinterpret_cast<ApplyLutMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == ApplyLutMem::kMagic, "apply_univariate_lut: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->size_only, "apply_univariate_lut: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(output_radix_lwe && input_radix_lwe && ksks && bsks, "apply_univariate_lut: null pointer");
  HX_PANIC_IF_FALSE(output_radix_lwe->lwe_dimension == input_radix_lwe->lwe_dimension,
                    "input and output radix ciphertexts should have the same lwe dimension");
  const uint32_t n = input_radix_lwe->num_radix_blocks;
  HX_PANIC_IF_FALSE(n <= m->drv.cap && n <= output_radix_lwe->num_radix_blocks,
                    "num radix blocks on which lut is applied should be smaller or equal to the number of lut radix "
                    "blocks");
  m->drv.round(streams, (uint64_t *)output_radix_lwe->ptr, nullptr, (const uint64_t *)input_radix_lwe->ptr, nullptr,
               m->d_lut_idx, n, ksks, bsks);
  if (output_radix_lwe->degrees)
    for (uint32_t i = 0; i < n; ++i) output_radix_lwe->degrees[i] = m->degree;
  if (output_radix_lwe->noise_levels)
    for (uint32_t i = 0; i < n; ++i) output_radix_lwe->noise_levels[i] = 1;
}

void cleanup_cuda_apply_univariate_lut_64(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  auto *m = reinterpret_cast<ApplyLutMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == ApplyLutMem::kMagic, "cleanup apply_univariate_lut: foreign scratch pointer");
  m->drv.release(streams);
  if (m->d_lut_idx) scratch_free(m->d_lut_idx);
  m->magic = 0;
  delete m;
  *mem_ptr_void = nullptr;
}

// integer.cuh:1002-1110: the output holds num_many_lut * n blocks, function t of input block s in block t * n + s
void cuda_apply_many_univariate_lut_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *output_radix_lwe,
                                             CudaRadixCiphertextFFI const *input_radix_lwe, int8_t *mem_ptr,
                                             void *const *ksks, void *const *bsks, uint32_t num_luts,
                                             uint32_t lut_stride) {
  first_gpu(streams);
  auto *m = reinterpret_cast<ApplyLutMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == ApplyLutMem::kMagic, "apply_many_univariate_lut: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->size_only, "apply_many_univariate_lut: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(output_radix_lwe && input_radix_lwe && ksks && bsks, "apply_many_univariate_lut: null pointer");
  HX_PANIC_IF_FALSE(num_luts >= 1 && num_luts <= m->num_many_lut,
                    "apply_many_univariate_lut: more functions than the scratch was created for");
  const uint32_t n = input_radix_lwe->num_radix_blocks;
  HX_PANIC_IF_FALSE((uint64_t)output_radix_lwe->num_radix_blocks >= (uint64_t)n * num_luts,
                    "output radix ciphertext should have at least num_many_lut times the number of blocks of the input");
  HX_PANIC_IF_FALSE(output_radix_lwe->lwe_dimension == input_radix_lwe->lwe_dimension,
                    "input and output radix ciphertexts should have the same lwe dimension");
  HX_PANIC_IF_FALSE((uint64_t)(num_luts - 1) * lut_stride < m->drv.p.N,
                    "apply_many_univariate_lut: lut_stride * (num_luts - 1) reaches past the polynomial");
  m->drv.round_many(streams, (uint64_t *)output_radix_lwe->ptr, (const uint64_t *)input_radix_lwe->ptr, m->d_lut_idx, n,
                    ksks, bsks, num_luts, lut_stride);
  const uint32_t total = n * num_luts;
  if (output_radix_lwe->degrees)
    for (uint32_t i = 0; i < total; ++i) output_radix_lwe->degrees[i] = m->degree;
  if (output_radix_lwe->noise_levels)
    for (uint32_t i = 0; i < total; ++i) output_radix_lwe->noise_levels[i] = 1;
}

void cleanup_cuda_apply_many_univariate_lut_64(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  cleanup_cuda_apply_univariate_lut_64(streams, mem_ptr_void);
}

// ---- cuda/include/linear_algebra.h:26-28 -----------------------------------------------------
void cuda_add_lwe_ciphertext_vector_inplace_64(void *stream, uint32_t gpu_index,
                                               CudaRadixCiphertextFFI *lwe_array_inout,
                                               CudaRadixCiphertextFFI const *input_2) {
  (void)gpu_index;
  HX_PANIC_IF_FALSE(lwe_array_inout && input_2 && lwe_array_inout->num_radix_blocks == input_2->num_radix_blocks &&
                        lwe_array_inout->lwe_dimension == input_2->lwe_dimension,
                    "add: operands must have the same shape");
  axpy((hipStream_t)stream, (uint64_t *)lwe_array_inout->ptr, nullptr, (const uint64_t *)lwe_array_inout->ptr, nullptr,
       1, (const uint64_t *)input_2->ptr, nullptr, lwe_array_inout->lwe_dimension + 1,
       lwe_array_inout->num_radix_blocks);
  for (uint32_t i = 0; i < lwe_array_inout->num_radix_blocks; ++i) {
    if (lwe_array_inout->degrees && input_2->degrees) lwe_array_inout->degrees[i] += input_2->degrees[i];
    if (lwe_array_inout->noise_levels && input_2->noise_levels)
      lwe_array_inout->noise_levels[i] += input_2->noise_levels[i];
  }
}

// The carry tree's first bootstrap holds two functions in its accumulator, so a block's value must stay below
// msg * carry / 2: at most 2 msg - 2 (two clean blocks added), one more for block 0 of an integer when an input carry
// is added to it.  A caller that tracks degrees gets the call refused instead of a wrong result.
static void check_propagation_degrees(const CudaRadixCiphertextFFI *ct, uint32_t msg, bool /*with_carry_in*/,
                                      uint32_t /*blocks_per_integer*/, const char *who) {
  if (ct == nullptr || ct->degrees == nullptr) return;
  for (uint32_t i = 0; i < ct->num_radix_blocks; ++i) {
    const uint64_t limit = 2ull * msg - 2;  // (the input carry itself comes on top of block 0: 2 msg - 1 there is fine)
    HX_PANIC_IF_FALSE(ct->degrees[i] <= limit,
                      "%s: block %u has degree %llu, the carry propagation accepts at most %llu (propagate the operands first)",
                      who, i, (unsigned long long)ct->degrees[i], (unsigned long long)limit);
  }
}

// ---- cuda/include/integer/integer.h:383-413 --------------------------------------------------
// num_blocks = blocks per integer; the ciphertexts handed to the launch may hold any whole number
// of integers up to the capacity given through hip_integer_scratch_batch (default 1).
// read by the NEXT scratch_* call of the same host thread (the radix scratch entry points have no batch parameter in
// the reference's prototypes): thread-local, so that concurrent host threads size their own scratches
static thread_local uint32_t g_scratch_batch = 1;
void hip_integer_scratch_batch(uint32_t num_integers) { g_scratch_batch = num_integers ? num_integers : 1; }
// blocks per GPU from which a KS -> PBS round spreads over one more GPU of the stream set; 0 restores the default, the
// reference's THRESHOLD_MULTI_GPU_* rule (helper_multi_gpu.cu:12-48); tests lower it
void hip_integer_set_multi_gpu_threshold(uint32_t blocks_per_gpu) {
  radix::g_multi_gpu_min_blocks.store(blocks_per_gpu);  // 0: the reference's rule (12 multi-bit, compute units + 1 classic)
}
// how many GPUs of a set of `gpu_count` a round of `num_blocks` blocks uses (get_active_gpu_count): for callers and tests
uint32_t hip_integer_active_gpu_count(uint32_t num_blocks, uint32_t gpu_count, uint32_t pbs_type, uint32_t first_gpu) {
  radix::Params p{};
  p.grouping = pbs_type == MULTI_BIT ? 1u : 0u;
  const uint32_t th = radix::multi_gpu_threshold(p, first_gpu);
  return std::max(1u, std::min(gpu_count, (num_blocks + th - 1) / th));
}

uint64_t scratch_cuda_propagate_single_carry_64_inplace_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                              CudaLweBootstrapKeyParamsFFI bsk_params,
                                                              CudaLweKeyswitchKeyParamsFFI ksk_params,
                                                              uint32_t num_blocks, uint32_t message_modulus,
                                                              uint32_t carry_modulus, uint32_t requested_flag,
                                                              bool allocate_gpu_memory,
                                                              enum PBS_MS_REDUCTION_T noise_reduction_type) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(requested_flag <= 2, "propagate_single_carry: unknown output flag %u", requested_flag);
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, (uint32_t)noise_reduction_type);
  t_dry = !allocate_gpu_memory;
  t_bytes = 0;
  auto *m = new PropagateMem();
  m->init(streams, p, num_blocks, g_scratch_batch);
  m->size_only = t_dry;
  t_dry = false;
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return t_bytes;
}
uint64_t scratch_cuda_add_and_propagate_single_carry_64_inplace_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks, uint32_t message_modulus, uint32_t carry_modulus,
    uint32_t requested_flag, bool allocate_gpu_memory, enum PBS_MS_REDUCTION_T noise_reduction_type) {
  return scratch_cuda_propagate_single_carry_64_inplace_async(streams, mem_ptr, bsk_params, ksk_params, num_blocks,
                                                              message_modulus, carry_modulus, requested_flag,
                                                              allocate_gpu_memory, noise_reduction_type);
}

void cuda_propagate_single_carry_64_inplace_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array,
                                                  CudaRadixCiphertextFFI *carry_out,
                                                  const CudaRadixCiphertextFFI *carry_in, int8_t *mem_ptr,
                                                  void *const *bsks, void *const *ksks, uint32_t requested_flag,
                                                  uint32_t uses_carry) {
  first_gpu(streams);
  auto *m = reinterpret_cast<PropagateMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == PropagateMem::kMagic, "propagate_single_carry: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->size_only, "propagate_single_carry: scratch was created with allocate_gpu_memory=false");
  // integer.cuh:2368-2370: the overflow flag needs the operands of the addition
  HX_PANIC_IF_FALSE(requested_flag != 1, "single carry propagation is not supported for overflow, try using "
                                         "add_and_propagate_single_carry");
  HX_PANIC_IF_FALSE(requested_flag == 0 || requested_flag == 2, "propagate_single_carry: unknown output flag %u",
                    requested_flag);
  const uint32_t cts = batch_of(lwe_array, m->blocks, "propagate_single_carry");
  check_propagation_degrees(lwe_array, m->drv.p.msg, uses_carry != 0, m->blocks, "propagate_single_carry");
  // the reference's Rust caller always hands over carry_in / carry_out structs (integer/gpu/ffi.rs:2213-2237);
  // they are read / written only when uses_carry / requested_flag say so
  const uint64_t *cin = nullptr;
  uint64_t *cout = nullptr;
  if (uses_carry != 0) {
    HX_PANIC_IF_FALSE(carry_in && carry_in->ptr && carry_in->num_radix_blocks >= cts &&
                          carry_in->lwe_dimension == lwe_array->lwe_dimension,
                      "propagate_single_carry: uses_carry needs one input carry block per integer");
    cin = (const uint64_t *)carry_in->ptr;
  }
  if (requested_flag == 2) {
    HX_PANIC_IF_FALSE(carry_out && carry_out->ptr && carry_out->num_radix_blocks >= cts &&
                          carry_out->lwe_dimension == lwe_array->lwe_dimension,
                      "propagate_single_carry: FLAG_CARRY needs one output carry block per integer");
    cout = (uint64_t *)carry_out->ptr;
  }
  m->run(streams, (uint64_t *)lwe_array->ptr, cts, ksks, bsks, cin, cout);
  if (cout)
    for (uint32_t i = 0; i < cts; ++i) {
      if (carry_out->degrees) carry_out->degrees[i] = 1;
      if (carry_out->noise_levels) carry_out->noise_levels[i] = 1;
    }
  for (uint32_t i = 0; i < lwe_array->num_radix_blocks; ++i) {
    if (lwe_array->degrees) lwe_array->degrees[i] = m->drv.p.msg - 1;
    if (lwe_array->noise_levels) lwe_array->noise_levels[i] = 1;
  }
}

void cuda_add_and_propagate_single_carry_64_inplace_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lhs_array,
                                                          const CudaRadixCiphertextFFI *rhs_array,
                                                          CudaRadixCiphertextFFI *carry_out,
                                                          const CudaRadixCiphertextFFI *carry_in, int8_t *mem_ptr,
                                                          void *const *bsks, void *const *ksks,
                                                          uint32_t requested_flag, uint32_t uses_carry) {
  first_gpu(streams);
  if (requested_flag != 1) {
    cuda_add_lwe_ciphertext_vector_inplace_64(streams.streams[0], G0(streams), lhs_array, rhs_array);
    cuda_propagate_single_carry_64_inplace_async(streams, lhs_array, carry_out, carry_in, mem_ptr, bsks, ksks,
                                                 requested_flag, uses_carry);
    return;
  }
  // FLAG_OVERFLOW (integer.cuh:2493-2520, integer_utilities.h:2311-2357): carry_out receives the signed-overflow flag of
  // lhs + rhs (+ carry_in) — the carry into the sign bit against the carry out of the last block.  The operands' last
  // blocks are read before the addition; both operands are clean (degrees <= msg - 1: the preparation is a bivariate
  // table of the two last blocks).
  auto *m = reinterpret_cast<PropagateMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == PropagateMem::kMagic, "add_and_propagate_single_carry: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->size_only, "add_and_propagate_single_carry: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(lhs_array && rhs_array && lhs_array->num_radix_blocks == rhs_array->num_radix_blocks &&
                        lhs_array->lwe_dimension == rhs_array->lwe_dimension,
                    "add_and_propagate_single_carry: operands must have the same shape");
  const uint32_t cts = batch_of(lhs_array, m->blocks, "add_and_propagate_single_carry");
  HX_PANIC_IF_FALSE(carry_out && carry_out->ptr && carry_out->num_radix_blocks >= cts &&
                        carry_out->lwe_dimension == lhs_array->lwe_dimension,
                    "when requesting FLAG_CARRY or FLAG_OVERFLOW, carry_out must be a valid pointer (one block per integer)");
  const uint32_t msg = m->drv.p.msg;
  for (const CudaRadixCiphertextFFI *op : {(const CudaRadixCiphertextFFI *)lhs_array, rhs_array})
    if (op->degrees)
      for (uint32_t c = 0; c < cts; ++c)
        HX_PANIC_IF_FALSE(op->degrees[(size_t)(c + 1) * m->blocks - 1] <= msg - 1,
                          "add_and_propagate_single_carry: FLAG_OVERFLOW needs clean last blocks (degree %llu > %u)",
                          (unsigned long long)op->degrees[(size_t)(c + 1) * m->blocks - 1], msg - 1);
  const uint64_t *cin = nullptr;
  if (uses_carry != 0) {
    HX_PANIC_IF_FALSE(carry_in && carry_in->ptr && carry_in->num_radix_blocks >= cts &&
                          carry_in->lwe_dimension == lhs_array->lwe_dimension,
                      "add_and_propagate_single_carry: uses_carry needs one input carry block per integer");
    cin = (const uint64_t *)carry_in->ptr;
  }
  m->pack_last_blocks(streams, (const uint64_t *)lhs_array->ptr, (const uint64_t *)rhs_array->ptr, cts);
  cuda_add_lwe_ciphertext_vector_inplace_64(streams.streams[0], G0(streams), lhs_array, rhs_array);
  check_propagation_degrees(lhs_array, msg, uses_carry != 0, m->blocks, "add_and_propagate_single_carry");
  m->run(streams, (uint64_t *)lhs_array->ptr, cts, ksks, bsks, cin, nullptr, (uint64_t *)carry_out->ptr);
  for (uint32_t i = 0; i < cts; ++i) {
    if (carry_out->degrees) carry_out->degrees[i] = 1;
    if (carry_out->noise_levels) carry_out->noise_levels[i] = 1;
  }
  for (uint32_t i = 0; i < lhs_array->num_radix_blocks; ++i) {
    if (lhs_array->degrees) lhs_array->degrees[i] = msg - 1;
    if (lhs_array->noise_levels) lhs_array->noise_levels[i] = 1;
  }
}

void cleanup_cuda_propagate_single_carry_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  auto *m = reinterpret_cast<PropagateMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == PropagateMem::kMagic, "cleanup propagate_single_carry: foreign scratch pointer");
  m->release(streams);
  delete m;
  *mem_ptr_void = nullptr;
}
void cleanup_cuda_add_and_propagate_single_carry_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  cleanup_cuda_propagate_single_carry_64_inplace(streams, mem_ptr_void);
}

// ---- cuda/include/integer/integer.h:173-187 --------------------------------------------------
uint64_t scratch_cuda_integer_mult_inplace_64_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                    bool const is_boolean_left, bool const is_boolean_right,
                                                    uint32_t message_modulus, uint32_t carry_modulus,
                                                    CudaLweBootstrapKeyParamsFFI bsk_params,
                                                    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks,
                                                    bool allocate_gpu_memory,
                                                    enum PBS_MS_REDUCTION_T noise_reduction_type) {
  first_gpu(streams);
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, (uint32_t)noise_reduction_type);
  t_dry = !allocate_gpu_memory;
  t_bytes = 0;
  if (is_boolean_left || is_boolean_right) {  // one operand is an encrypted boolean: a select, not a product
    auto *bm = new BoolMulMem();
    bm->init(streams, p, num_blocks, g_scratch_batch);
    bm->size_only = t_dry;
    t_dry = false;
    *mem_ptr = reinterpret_cast<int8_t *>(bm);
    return t_bytes;
  }
  auto *m = new MulMem();
  m->init(streams, p, num_blocks, g_scratch_batch);
  m->size_only = t_dry;
  t_dry = false;
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return t_bytes;
}

void cuda_integer_mult_inplace_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *radix_lwe_inout,
                                        bool const is_bool_left, CudaRadixCiphertextFFI const *radix_lwe_right,
                                        bool const is_bool_right, void *const *bsks, void *const *ksks,
                                        int8_t *mem_ptr, uint32_t polynomial_size, uint32_t num_blocks) {
  first_gpu(streams);
  if (is_bool_left || is_bool_right) {
    // multiplication.cuh:508-520: the boolean operand is ONE block (here: one per integer of the batch, packed at the
    // start of its ciphertext); the other operand's blocks are kept or zeroed.  In place on radix_lwe_inout, which
    // holds the full-width integers either way (as the left operand, or as the destination when it is the boolean).
    auto *bm = reinterpret_cast<BoolMulMem *>(mem_ptr);
    HX_PANIC_IF_FALSE(bm && bm->magic == BoolMulMem::kMagic,
                      "integer_mult: boolean operands need a scratch created with is_boolean_left / is_boolean_right");
    HX_PANIC_IF_FALSE(!bm->size_only, "integer_mult: scratch was created with allocate_gpu_memory=false");
    HX_PANIC_IF_FALSE(polynomial_size == bm->drv.p.N && num_blocks == bm->blocks, "integer_mult: call does not match the scratch");
    HX_PANIC_IF_FALSE(radix_lwe_inout && radix_lwe_right && radix_lwe_inout->lwe_dimension == radix_lwe_right->lwe_dimension,
                      "integer_mult: input and output lwe dimensions should be the same");
    const uint32_t cts = batch_of(radix_lwe_inout, bm->blocks, "integer_mult");
    const CudaRadixCiphertextFFI *cond = is_bool_right ? radix_lwe_right : radix_lwe_inout;
    const CudaRadixCiphertextFFI *value = is_bool_right ? (const CudaRadixCiphertextFFI *)radix_lwe_inout : radix_lwe_right;
    HX_PANIC_IF_FALSE(cond->num_radix_blocks >= cts && value->num_radix_blocks >= cts * bm->blocks,
                      "integer_mult: input or output does not have enough radix blocks");
    // the packing msg * value + condition must stay below the padding bit: clean value blocks, a boolean that IS one
    if (value->degrees)
      for (uint32_t i = 0; i < cts * bm->blocks; ++i)
        HX_PANIC_IF_FALSE(value->degrees[i] <= bm->drv.p.msg - 1,
                          "integer_mult: block %u of the non-boolean operand has degree %llu, at most %u is accepted (propagate it first)",
                          i, (unsigned long long)value->degrees[i], bm->drv.p.msg - 1);
    if (cond->degrees)
      for (uint32_t c = 0; c < cts; ++c)
        HX_PANIC_IF_FALSE(cond->degrees[c] <= 1, "integer_mult: the boolean operand's block %u has degree %llu", c,
                          (unsigned long long)cond->degrees[c]);
    // (a boolean that sits in the destination is read by the packing launch before the bootstrap overwrites it)
    bm->run(streams, (uint64_t *)radix_lwe_inout->ptr, (const uint64_t *)value->ptr, (const uint64_t *)cond->ptr, cts,
            ksks, bsks);
    for (uint32_t i = 0; i < cts * bm->blocks; ++i) {
      if (radix_lwe_inout->degrees) radix_lwe_inout->degrees[i] = bm->drv.p.msg - 1;
      if (radix_lwe_inout->noise_levels) radix_lwe_inout->noise_levels[i] = 1;
    }
    return;
  }
  auto *m = reinterpret_cast<MulMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == MulMem::kMagic, "integer_mult: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->size_only, "integer_mult: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(polynomial_size == m->drv.p.N && num_blocks == m->blocks, "integer_mult: call does not match the scratch");
  const uint32_t cts = batch_of(radix_lwe_inout, m->blocks, "integer_mult");
  HX_PANIC_IF_FALSE(radix_lwe_right && radix_lwe_right->num_radix_blocks == radix_lwe_inout->num_radix_blocks,
                    "integer_mult: operands must have the same shape");
  m->run(streams, (uint64_t *)radix_lwe_inout->ptr, (const uint64_t *)radix_lwe_right->ptr, cts, ksks, bsks);
  for (uint32_t i = 0; i < radix_lwe_inout->num_radix_blocks; ++i) {
    if (radix_lwe_inout->degrees) radix_lwe_inout->degrees[i] = m->drv.p.msg - 1;
    if (radix_lwe_inout->noise_levels) radix_lwe_inout->noise_levels[i] = 1;
  }
}

void cleanup_cuda_integer_mult_inplace_64(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  if (*mem_ptr_void && reinterpret_cast<BoolMulMem *>(*mem_ptr_void)->magic == BoolMulMem::kMagic) {
    auto *bm = reinterpret_cast<BoolMulMem *>(*mem_ptr_void);
    bm->release(streams);
    delete bm;
    *mem_ptr_void = nullptr;
    return;
  }
  auto *m = reinterpret_cast<MulMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == MulMem::kMagic, "cleanup integer_mult: foreign scratch pointer");
  m->release(streams);
  delete m;
  *mem_ptr_void = nullptr;
}

// ---- cuda/include/integer/integer.h:189-198, :312-316 ------------------------------------------
// negation.cuh:85-155 (host_negation_with_correcting_term): block 0 becomes z - b, every later block z - (b + z / msg)
// with z = msg, so that the blocks stay non-negative and the borrowed unit is handed to the next block; the degrees
// follow the reference's loop (including its integer division inside the ceil)
void cuda_negate_ciphertext_64(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_out,
                               CudaRadixCiphertextFFI const *lwe_array_in, uint32_t message_modulus,
                               uint32_t carry_modulus, uint32_t num_radix_blocks) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(lwe_array_out != lwe_array_in, "Output and input pointers must be different for out-of-place operations");
  HX_PANIC_IF_FALSE(lwe_array_out && lwe_array_in && lwe_array_out->ptr && lwe_array_in->ptr, "negate: null pointer");
  HX_PANIC_IF_FALSE(lwe_array_out->num_radix_blocks >= num_radix_blocks && lwe_array_in->num_radix_blocks >= num_radix_blocks,
                    "lwe_array_in and lwe_array_out num radix blocks must be greater or equal to the number of blocks to negate");
  HX_PANIC_IF_FALSE(lwe_array_out->lwe_dimension == lwe_array_in->lwe_dimension,
                    "lwe_array_in and lwe_array_out lwe_dimension must be the same");
  HX_PANIC_IF_FALSE(message_modulus >= 2 && carry_modulus >= 1, "negate: bad moduli");
  if (num_radix_blocks == 0) return;
  const uint64_t delta = ((uint64_t)1 << 63) / ((uint64_t)message_modulus * carry_modulus);
  const uint64_t z = ((2ull * message_modulus - 1) / message_modulus) * message_modulus;
  HX_LAUNCH(lwe_negate_const_kernel, dim3(num_radix_blocks), dim3(256), 0, S0(streams), (uint64_t *)lwe_array_out->ptr,
            (const uint64_t *)lwe_array_in->ptr, lwe_array_in->lwe_dimension + 1, num_radix_blocks, num_radix_blocks,
            z * delta, (z - z / message_modulus) * delta);
  uint64_t zb = 0;
  for (uint32_t i = 0; i < lwe_array_out->num_radix_blocks; ++i) {
    const uint64_t d = (lwe_array_in->degrees ? lwe_array_in->degrees[i] : message_modulus - 1) + zb;
    const uint64_t zz = std::max<uint64_t>(1, d / message_modulus) * message_modulus;
    if (lwe_array_out->degrees) lwe_array_out->degrees[i] = zz - zb;
    if (lwe_array_out->noise_levels && lwe_array_in->noise_levels) lwe_array_out->noise_levels[i] = lwe_array_in->noise_levels[i];
    zb = zz / message_modulus;
  }
}

// scalar_addition.cuh:27-55: scalar_input (device) and h_scalar_input (host) hold the same num_scalars clear blocks
void cuda_scalar_addition_ciphertext_64_inplace(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array,
                                                void const *scalar_input, void const *h_scalar_input,
                                                uint32_t num_scalars, uint32_t message_modulus, uint32_t carry_modulus) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(lwe_array && lwe_array->ptr && (num_scalars == 0 || (scalar_input && h_scalar_input)),
                    "scalar_addition: null pointer");
  HX_PANIC_IF_FALSE(lwe_array->num_radix_blocks >= num_scalars,
                    "num scalars should be smaller or equal to input num radix blocks");
  if (num_scalars == 0) return;
  const uint64_t delta = ((uint64_t)1 << 63) / ((uint64_t)message_modulus * carry_modulus);
  HX_LAUNCH(lwe_body_add_scalars_kernel, dim3((num_scalars + 255) / 256), dim3(256), 0, S0(streams),
            (uint64_t *)lwe_array->ptr, (const uint64_t *)scalar_input, lwe_array->lwe_dimension + 1, num_scalars, delta);
  if (lwe_array->degrees)
    for (uint32_t i = 0; i < num_scalars; ++i) lwe_array->degrees[i] += ((const uint64_t *)h_scalar_input)[i];
}

// bitwise_ops.cuh:164-191 (host_bitnot): (ct_message_modulus - 1) - block, as a plain negation plus a constant
void cuda_bitnot_ciphertext_64(CudaStreamsFFI streams, CudaRadixCiphertextFFI *radix_ciphertext,
                               uint32_t ct_message_modulus, uint32_t param_message_modulus,
                               uint32_t param_carry_modulus) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(radix_ciphertext && radix_ciphertext->ptr, "bitnot: null pointer");
  HX_PANIC_IF_FALSE(ct_message_modulus >= 1 && param_message_modulus >= 2 && param_carry_modulus >= 1, "bitnot: bad moduli");
  const uint32_t nb = radix_ciphertext->num_radix_blocks;
  if (nb == 0) return;
  const uint64_t delta = (uint64_t)1 << (63 - __builtin_ctzll((uint64_t)param_message_modulus * param_carry_modulus));
  const uint64_t enc = delta * (ct_message_modulus - 1);
  HX_LAUNCH(lwe_negate_const_kernel, dim3(nb), dim3(256), 0, S0(streams), (uint64_t *)radix_ciphertext->ptr,
            (const uint64_t *)radix_ciphertext->ptr, radix_ciphertext->lwe_dimension + 1, nb, nb, enc, enc);
  if (radix_ciphertext->degrees)
    for (uint32_t i = 0; i < nb; ++i) radix_ciphertext->degrees[i] = ct_message_modulus - 1;
}

// ---- cuda/include/integer/integer.h:318-347 -----------------------------------------------------
static uint64_t scratch_bitop(CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
                              CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t count, uint32_t message_modulus,
                              uint32_t carry_modulus, uint32_t op, bool scalar, bool allocate_gpu_memory,
                              uint32_t noise_reduction_type) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(mem_ptr != nullptr, "bitop: null pointer");
  HX_PANIC_IF_FALSE(scalar ? (op >= SCALAR_BITAND && op <= SCALAR_BITXOR) : op <= BITXOR,
                    "bitop: operation %u does not belong to this entry point", op);
  t_dry = !allocate_gpu_memory;
  t_bytes = 0;
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, noise_reduction_type);
  auto *m = new BitopMem();
  m->op = op;
  const size_t lw = (size_t)(p.k + 1) * p.N;
  const uint32_t cap = std::max<uint32_t>(1, count) * g_scratch_batch;
  std::vector<std::vector<uint64_t>> luts(scalar ? p.msg : 1, std::vector<uint64_t>(lw));
  if (scalar) {
    for (uint32_t c = 0; c < p.msg; ++c)
      generate_lut(p, luts[c].data(), [op, c](uint64_t x) { return bitop_apply(op, x, c); });
  } else {
    const uint32_t msg = p.msg;
    generate_lut(p, luts[0].data(), [op, msg](uint64_t x) { return bitop_apply(op, x / msg, x % msg); });
  }
  m->drv.init(streams, p, cap, luts);
  if (!scalar) {
    radix_alloc((void **)&m->d_pack, (size_t)cap * (p.big_n + 1) * sizeof(uint64_t));
    radix_alloc((void **)&m->d_lut_idx, (size_t)cap * sizeof(uint64_t));
    if (!t_dry) HX_CHECK(hipMemsetAsync(m->d_lut_idx, 0, (size_t)cap * sizeof(uint64_t), S0(streams)));
  }
  m->size_only = t_dry;
  t_dry = false;
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return t_bytes;
}
uint64_t scratch_cuda_integer_bitop_inplace_64_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                     CudaLweBootstrapKeyParamsFFI bsk_params,
                                                     CudaLweKeyswitchKeyParamsFFI ksk_params,
                                                     uint32_t lwe_ciphertext_count, uint32_t message_modulus,
                                                     uint32_t carry_modulus, enum BITOP_TYPE op_type,
                                                     bool allocate_gpu_memory,
                                                     enum PBS_MS_REDUCTION_T noise_reduction_type) {
  return scratch_bitop(streams, mem_ptr, bsk_params, ksk_params, lwe_ciphertext_count, message_modulus, carry_modulus,
                       (uint32_t)op_type, false, allocate_gpu_memory, (uint32_t)noise_reduction_type);
}
uint64_t scratch_cuda_integer_scalar_bitop_inplace_64_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                            CudaLweBootstrapKeyParamsFFI bsk_params,
                                                            CudaLweKeyswitchKeyParamsFFI ksk_params,
                                                            uint32_t lwe_ciphertext_count, uint32_t message_modulus,
                                                            uint32_t carry_modulus, enum BITOP_TYPE op_type,
                                                            bool allocate_gpu_memory,
                                                            enum PBS_MS_REDUCTION_T noise_reduction_type) {
  return scratch_bitop(streams, mem_ptr, bsk_params, ksk_params, lwe_ciphertext_count, message_modulus, carry_modulus,
                       (uint32_t)op_type, true, allocate_gpu_memory, (uint32_t)noise_reduction_type);
}

// bitwise_ops.cuh:230-271 (host_bitop): one bivariate round over all blocks
void cuda_integer_bitop_inplace_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_inout,
                                         CudaRadixCiphertextFFI const *lwe_array_2, int8_t *mem_ptr,
                                         void *const *bsks, void *const *ksks) {
  first_gpu(streams);
  auto *m = reinterpret_cast<BitopMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == BitopMem::kMagic && m->op <= BITXOR, "integer_bitop: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->size_only, "integer_bitop: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(lwe_array_inout && lwe_array_2 && lwe_array_inout->ptr && lwe_array_2->ptr && bsks && ksks,
                    "integer_bitop: null pointer");
  HX_PANIC_IF_FALSE(lwe_array_inout->num_radix_blocks == lwe_array_2->num_radix_blocks,
                    "input and output num radix blocks must be equal");
  HX_PANIC_IF_FALSE(lwe_array_inout->lwe_dimension == lwe_array_2->lwe_dimension,
                    "input and output lwe dimension must be equal");
  const uint32_t nb = lwe_array_inout->num_radix_blocks;
  HX_PANIC_IF_FALSE(nb <= m->drv.cap, "integer_bitop: %u blocks exceed the scratch capacity %u", nb, m->drv.cap);
  const Params &p = m->drv.p;
  if (lwe_array_inout->degrees && lwe_array_2->degrees)
    for (uint32_t i = 0; i < nb; ++i)
      HX_PANIC_IF_FALSE(lwe_array_inout->degrees[i] < p.msg && lwe_array_2->degrees[i] < p.msg,
                        "integer_bitop: block %u carries a degree of %llu / %llu, the packed pair needs clean blocks", i,
                        (unsigned long long)lwe_array_inout->degrees[i], (unsigned long long)lwe_array_2->degrees[i]);
  uint64_t *v = (uint64_t *)lwe_array_inout->ptr;
  axpy(S0(streams), m->d_pack, nullptr, v, nullptr, p.msg, (const uint64_t *)lwe_array_2->ptr, nullptr, p.big_n + 1, nb);
  m->drv.round(streams, v, nullptr, m->d_pack, nullptr, m->d_lut_idx, nb, ksks, bsks);
  for (uint32_t i = 0; i < nb; ++i) {
    if (lwe_array_inout->degrees)
      lwe_array_inout->degrees[i] = bitop_degree(m->op, lwe_array_inout->degrees[i],
                                                 lwe_array_2->degrees ? lwe_array_2->degrees[i] : p.msg - 1);
    if (lwe_array_inout->noise_levels) lwe_array_inout->noise_levels[i] = 1;
  }
}

// scalar_bitops.cuh:6-66 (host_scalar_bitop): clear_blocks (device) index the per-value tables; the blocks past the
// clear ones are ANDed with zero / left as they are
void cuda_integer_scalar_bitop_inplace_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_inout,
                                                void const *clear_blocks, void const *h_clear_blocks,
                                                uint32_t num_clear_blocks, int8_t *mem_ptr, void *const *bsks,
                                                void *const *ksks) {
  first_gpu(streams);
  auto *m = reinterpret_cast<BitopMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == BitopMem::kMagic && m->op >= SCALAR_BITAND, "integer_scalar_bitop: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->size_only, "integer_scalar_bitop: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(lwe_array_inout && lwe_array_inout->ptr && bsks && ksks, "integer_scalar_bitop: null pointer");
  const uint32_t nb = lwe_array_inout->num_radix_blocks;
  HX_PANIC_IF_FALSE(num_clear_blocks <= nb && num_clear_blocks <= m->drv.cap,
                    "integer_scalar_bitop: %u clear blocks for %u radix blocks (scratch capacity %u)", num_clear_blocks, nb,
                    m->drv.cap);
  HX_PANIC_IF_FALSE(num_clear_blocks == 0 || (clear_blocks && h_clear_blocks), "integer_scalar_bitop: null clear blocks");
  const Params &p = m->drv.p;
  const size_t w = (size_t)p.big_n + 1;
  uint64_t *v = (uint64_t *)lwe_array_inout->ptr;
  const uint64_t *hc = (const uint64_t *)h_clear_blocks;
  if (num_clear_blocks) {
    for (uint32_t i = 0; i < num_clear_blocks; ++i)
      HX_PANIC_IF_FALSE(hc[i] < p.msg, "integer_scalar_bitop: clear block %u is %llu, not below the message modulus", i,
                        (unsigned long long)hc[i]);
    m->drv.round(streams, v, nullptr, v, nullptr, (const uint64_t *)clear_blocks, num_clear_blocks, ksks, bsks);
    for (uint32_t i = 0; i < num_clear_blocks; ++i) {
      if (lwe_array_inout->degrees) lwe_array_inout->degrees[i] = bitop_degree(m->op, hc[i], lwe_array_inout->degrees[i]);
      if (lwe_array_inout->noise_levels) lwe_array_inout->noise_levels[i] = 1;
    }
  }
  if (m->op == SCALAR_BITAND && num_clear_blocks < nb) {
    HX_CHECK(hipMemsetAsync(v + (size_t)num_clear_blocks * w, 0, (size_t)(nb - num_clear_blocks) * w * sizeof(uint64_t),
                            S0(streams)));
    for (uint32_t i = num_clear_blocks; i < nb; ++i) {
      if (lwe_array_inout->degrees) lwe_array_inout->degrees[i] = 0;
      if (lwe_array_inout->noise_levels) lwe_array_inout->noise_levels[i] = 0;
    }
  }
}

static void cleanup_bitop(CudaStreamsFFI streams, int8_t **mem_ptr_void, const char *who) {
  first_gpu(streams);
  auto *m = reinterpret_cast<BitopMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == BitopMem::kMagic, "%s: foreign scratch pointer", who);
  m->drv.release(streams);
  for (uint64_t *d : {m->d_pack, m->d_lut_idx})
    if (d) scratch_free(d);
  m->magic = 0;
  delete m;
  *mem_ptr_void = nullptr;
}
void cleanup_cuda_integer_bitop_inplace_64(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  cleanup_bitop(streams, mem_ptr_void, "cleanup integer_bitop");
}
void cleanup_cuda_integer_scalar_bitop_inplace_64(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  cleanup_bitop(streams, mem_ptr_void, "cleanup integer_scalar_bitop");
}

// ---- cuda/include/integer/integer.h:559-573 -----------------------------------------------------
// lhs - rhs = lhs + (negation of rhs with its correcting term), then the addition's carry propagation.  Block 0 of the
// sum reaches 2 msg - 1 (msg - b0 on top of a clean block): exactly what block 0 takes with an input carry, so an input
// carry on top of it is refused, and so is FLAG_OVERFLOW (the reference's callers — integer/gpu/server_key/radix/sub.rs:
// 222, :347-400 — pass neither).
uint64_t scratch_cuda_sub_and_propagate_single_carry_64_inplace_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks, uint32_t message_modulus, uint32_t carry_modulus,
    uint32_t requested_flag, bool allocate_gpu_memory, enum PBS_MS_REDUCTION_T noise_reduction_type) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(mem_ptr != nullptr, "sub_and_propagate_single_carry: null pointer");
  HX_PANIC_IF_FALSE(requested_flag == 0 || requested_flag == 2,
                    "sub_and_propagate_single_carry: output flag %u is not wired (0 = none, 2 = carry)", requested_flag);
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, (uint32_t)noise_reduction_type);
  t_dry = !allocate_gpu_memory;
  t_bytes = 0;
  auto *m = new SubMem();
  m->prop.init(streams, p, num_blocks, g_scratch_batch);
  radix_alloc((void **)&m->d_neg, (size_t)num_blocks * g_scratch_batch * (p.big_n + 1) * sizeof(uint64_t));
  m->prop.size_only = t_dry;
  t_dry = false;
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return t_bytes;
}

void cuda_sub_and_propagate_single_carry_64_inplace_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lhs_array,
                                                          const CudaRadixCiphertextFFI *rhs_array,
                                                          CudaRadixCiphertextFFI *carry_out,
                                                          const CudaRadixCiphertextFFI *carry_in, int8_t *mem_ptr,
                                                          void *const *bsks, void *const *ksks,
                                                          uint32_t requested_flag, uint32_t uses_carry) {
  first_gpu(streams);
  (void)carry_in;
  auto *m = reinterpret_cast<SubMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == SubMem::kMagic, "sub_and_propagate_single_carry: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->prop.size_only, "sub_and_propagate_single_carry: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(requested_flag == 0 || requested_flag == 2,
                    "sub_and_propagate_single_carry: output flag %u is not wired (0 = none, 2 = carry)", requested_flag);
  HX_PANIC_IF_FALSE(uses_carry == 0, "sub_and_propagate_single_carry: an input carry is not wired (block 0 already holds the "
                                     "borrowed unit of the negation)");
  HX_PANIC_IF_FALSE(lhs_array && rhs_array && lhs_array->ptr && rhs_array->ptr && bsks && ksks &&
                        lhs_array->num_radix_blocks == rhs_array->num_radix_blocks &&
                        lhs_array->lwe_dimension == rhs_array->lwe_dimension,
                    "sub_and_propagate_single_carry: operands must have the same shape");
  const uint32_t L = m->prop.blocks, cts = batch_of(lhs_array, L, "sub_and_propagate_single_carry");
  const Params &p = m->prop.drv.p;
  for (const CudaRadixCiphertextFFI *op : {(const CudaRadixCiphertextFFI *)lhs_array, rhs_array})
    if (op->degrees)
      for (uint32_t i = 0; i < op->num_radix_blocks; ++i)
        HX_PANIC_IF_FALSE(op->degrees[i] <= p.msg - 1,
                          "sub_and_propagate_single_carry: block %u has degree %llu, the subtraction takes clean operands "
                          "(propagate them first)", i, (unsigned long long)op->degrees[i]);
  uint64_t *cout = nullptr;
  if (requested_flag == 2) {
    HX_PANIC_IF_FALSE(carry_out && carry_out->ptr && carry_out->num_radix_blocks >= cts &&
                          carry_out->lwe_dimension == lhs_array->lwe_dimension,
                      "sub_and_propagate_single_carry: FLAG_CARRY needs one output carry block per integer");
    cout = (uint64_t *)carry_out->ptr;
  }
  const hipStream_t st = S0(streams);
  const uint32_t w = p.big_n + 1, T = cts * L;
  const uint64_t delta = ((uint64_t)1 << 63) / ((uint64_t)p.msg * p.carry);
  HX_LAUNCH(lwe_negate_const_kernel, dim3(T), dim3(256), 0, st, m->d_neg, (const uint64_t *)rhs_array->ptr, w, T, L,
            (uint64_t)p.msg * delta, (uint64_t)(p.msg - 1) * delta);
  uint64_t *v = (uint64_t *)lhs_array->ptr;
  axpy(st, v, nullptr, v, nullptr, 1, m->d_neg, nullptr, w, T);
  m->prop.run(streams, v, cts, ksks, bsks, nullptr, cout);
  if (cout)
    for (uint32_t i = 0; i < cts; ++i) {
      if (carry_out->degrees) carry_out->degrees[i] = 1;
      if (carry_out->noise_levels) carry_out->noise_levels[i] = 1;
    }
  for (uint32_t i = 0; i < lhs_array->num_radix_blocks; ++i) {
    if (lhs_array->degrees) lhs_array->degrees[i] = p.msg - 1;
    if (lhs_array->noise_levels) lhs_array->noise_levels[i] = 1;
  }
}

void cleanup_cuda_sub_and_propagate_single_carry_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  auto *m = reinterpret_cast<SubMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == SubMem::kMagic, "cleanup sub_and_propagate_single_carry: foreign scratch pointer");
  m->prop.release(streams);
  if (m->d_neg) scratch_free(m->d_neg);
  m->magic = 0;
  delete m;
  *mem_ptr_void = nullptr;
}

// ---- cuda/include/integer/integer.h:415-431 -----------------------------------------------------
// lhs -= rhs with the borrow: the subtraction above with FLAG_CARRY; the carry of lhs + (2^bits - rhs) is 1 exactly when
// nothing was borrowed, so the overflow block is 1 - carry (levelled).  An input borrow is not wired (the reference's unsigned
// overflowing_sub passes none: integer/gpu/server_key/radix/sub.rs unsigned_overflowing_sub).
uint64_t scratch_cuda_integer_overflowing_sub_64_inplace_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                               CudaLweBootstrapKeyParamsFFI bsk_params,
                                                               CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks,
                                                               uint32_t message_modulus, uint32_t carry_modulus,
                                                               uint32_t compute_overflow, bool allocate_gpu_memory,
                                                               enum PBS_MS_REDUCTION_T noise_reduction_type) {
  (void)compute_overflow;
  return scratch_cuda_sub_and_propagate_single_carry_64_inplace_async(streams, mem_ptr, bsk_params, ksk_params, num_blocks,
                                                                      message_modulus, carry_modulus, 2, allocate_gpu_memory,
                                                                      noise_reduction_type);
}
void cuda_integer_overflowing_sub_64_inplace_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lhs_array,
                                                   const CudaRadixCiphertextFFI *rhs_array,
                                                   CudaRadixCiphertextFFI *overflow_block,
                                                   const CudaRadixCiphertextFFI *input_borrow, int8_t *mem_ptr,
                                                   void *const *bsks, void *const *ksks, uint32_t compute_overflow,
                                                   uint32_t uses_input_borrow) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(uses_input_borrow == 0, "integer_overflowing_sub: an input borrow is not wired");
  if (compute_overflow == 0) {
    cuda_sub_and_propagate_single_carry_64_inplace_async(streams, lhs_array, rhs_array, overflow_block, input_borrow, mem_ptr, bsks,
                                                         ksks, 0, 0);
    return;
  }
  HX_PANIC_IF_FALSE(overflow_block && overflow_block->ptr && lhs_array, "integer_overflowing_sub: the overflow block is missing");
  cuda_sub_and_propagate_single_carry_64_inplace_async(streams, lhs_array, rhs_array, overflow_block, input_borrow, mem_ptr, bsks,
                                                       ksks, 2, 0);
  auto *m = reinterpret_cast<SubMem *>(mem_ptr);
  const Params &p = m->prop.drv.p;
  const uint32_t cts = lhs_array->num_radix_blocks / m->prop.blocks;
  const uint64_t delta = ((uint64_t)1 << 63) / ((uint64_t)p.msg * p.carry);
  HX_LAUNCH(lwe_negate_const_kernel, dim3(cts), dim3(256), 0, S0(streams), (uint64_t *)overflow_block->ptr,
            (const uint64_t *)overflow_block->ptr, p.big_n + 1, cts, 1, delta, delta);  // 1 - carry
}
void cleanup_cuda_integer_overflowing_sub_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  cleanup_cuda_sub_and_propagate_single_carry_64_inplace(streams, mem_ptr_void);
}

// ---- cuda/include/integer/integer.h:159-171 -----------------------------------------------------
uint64_t scratch_cuda_full_propagation_64_inplace_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                        CudaLweBootstrapKeyParamsFFI bsk_params,
                                                        CudaLweKeyswitchKeyParamsFFI ksk_params,
                                                        uint32_t message_modulus, uint32_t carry_modulus,
                                                        bool allocate_gpu_memory,
                                                        enum PBS_MS_REDUCTION_T noise_reduction_type) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(mem_ptr != nullptr, "full_propagation: null pointer");
  t_dry = !allocate_gpu_memory;
  t_bytes = 0;
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, (uint32_t)noise_reduction_type);
  auto *m = new FullPropMem();
  const size_t lw = (size_t)(p.k + 1) * p.N;
  std::vector<std::vector<uint64_t>> luts(2, std::vector<uint64_t>(lw));
  const uint32_t msg = p.msg;
  generate_lut(p, luts[0].data(), [msg](uint64_t x) { return x % msg; });
  generate_lut(p, luts[1].data(), [msg](uint64_t x) { return x / msg; });
  m->drv.init(streams, p, 2, luts);
  radix_alloc((void **)&m->d_two, 2 * (size_t)(p.big_n + 1) * sizeof(uint64_t));
  m->d_lut_idx = dev_upload(S0(streams), std::vector<uint64_t>{0, 1});
  m->size_only = t_dry;
  t_dry = false;
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return t_bytes;
}

void cuda_full_propagation_64_inplace_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *input_blocks,
                                            int8_t *mem_ptr, void *const *ksks, void *const *bsks,
                                            uint32_t num_blocks) {
  first_gpu(streams);
  auto *m = reinterpret_cast<FullPropMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == FullPropMem::kMagic, "full_propagation: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->size_only, "full_propagation: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(input_blocks && input_blocks->ptr && ksks && bsks && num_blocks <= input_blocks->num_radix_blocks,
                    "full_propagation: null pointer or more blocks than the ciphertext holds");
  const Params &p = m->drv.p;
  const hipStream_t st = S0(streams);
  const size_t w = (size_t)p.big_n + 1;
  uint64_t *v = (uint64_t *)input_blocks->ptr;
  for (uint32_t i = 0; i < num_blocks; ++i) {
    uint64_t *blk = v + (size_t)i * w;
    HX_CHECK(hipMemcpyAsync(m->d_two, blk, w * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    HX_CHECK(hipMemcpyAsync(m->d_two + w, blk, w * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    m->drv.round(streams, m->d_two, nullptr, m->d_two, nullptr, m->d_lut_idx, 2, ksks, bsks);
    HX_CHECK(hipMemcpyAsync(blk, m->d_two, w * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    if (input_blocks->degrees) input_blocks->degrees[i] = p.msg - 1;
    if (input_blocks->noise_levels) input_blocks->noise_levels[i] = 1;
    if (i + 1 < num_blocks) {
      axpy(st, blk + w, nullptr, blk + w, nullptr, 1, m->d_two + w, nullptr, (uint32_t)w, 1);
      if (input_blocks->degrees) input_blocks->degrees[i + 1] += p.carry - 1;
      if (input_blocks->noise_levels) input_blocks->noise_levels[i + 1] += 1;
    }
  }
}

void cleanup_cuda_full_propagation_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  auto *m = reinterpret_cast<FullPropMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == FullPropMem::kMagic, "cleanup full_propagation: foreign scratch pointer");
  m->drv.release(streams);
  for (uint64_t *d : {m->d_two, m->d_lut_idx})
    if (d) scratch_free(d);
  m->magic = 0;
  delete m;
  *mem_ptr_void = nullptr;
}

// ---- cuda/include/integer/integer.h:246-276 (comparison), :349-365 (cmux) ----------------------------
// Unsigned operands; the boolean lands in block 0 of lwe_array_out (its other blocks are cleared), as the single-block
// result of the reference's EQ ... LE.  MAX / MIN: the ordering's flag drives a cmux of the operands.
struct CompareScratch {
  static constexpr uint32_t kMagic = 0x43535231;  // "CSR1"
  uint32_t magic = kMagic;
  uint32_t op = 0;
  CompareMem cmp;
  CmuxMem mux;  // MAX / MIN only
  uint64_t *d_flag = nullptr;
};
uint64_t scratch_cuda_integer_comparison_64_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                  CudaLweBootstrapKeyParamsFFI bsk_params,
                                                  CudaLweKeyswitchKeyParamsFFI ksk_params,
                                                  uint32_t lwe_ciphertext_count, uint32_t message_modulus,
                                                  uint32_t carry_modulus, enum COMPARISON_TYPE op_type, bool is_signed,
                                                  bool allocate_gpu_memory,
                                                  enum PBS_MS_REDUCTION_T noise_reduction_type) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(mem_ptr != nullptr && lwe_ciphertext_count >= 1, "integer_comparison: null pointer or no blocks");
  HX_PANIC_IF_FALSE(!is_signed, "integer_comparison: signed operands are not wired");
  HX_PANIC_IF_FALSE((uint32_t)op_type <= MIN, "integer_comparison: unknown operation %u", (uint32_t)op_type);
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, (uint32_t)noise_reduction_type);
  t_dry = !allocate_gpu_memory;
  t_bytes = 0;
  auto *m = new CompareScratch();
  m->op = (uint32_t)op_type;
  const bool select = op_type == MAX || op_type == MIN;
  m->cmp.init(streams, p, lwe_ciphertext_count, select ? (uint32_t)GE : (uint32_t)op_type);
  if (select) {
    m->mux.init(streams, p, lwe_ciphertext_count);
    radix_alloc((void **)&m->d_flag, (size_t)(p.big_n + 1) * sizeof(uint64_t));
  }
  m->cmp.size_only = t_dry;
  t_dry = false;
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return t_bytes;
}

void cuda_integer_comparison_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_out,
                                      CudaRadixCiphertextFFI const *lwe_array_1,
                                      CudaRadixCiphertextFFI const *lwe_array_2, int8_t *mem_ptr, void *const *bsks,
                                      void *const *ksks) {
  first_gpu(streams);
  auto *m = reinterpret_cast<CompareScratch *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == CompareScratch::kMagic, "integer_comparison: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->cmp.size_only, "integer_comparison: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(lwe_array_out && lwe_array_1 && lwe_array_2 && lwe_array_out->ptr && lwe_array_1->ptr &&
                        lwe_array_2->ptr && bsks && ksks,
                    "integer_comparison: null pointer");
  const uint32_t L = m->cmp.blocks;
  HX_PANIC_IF_FALSE(lwe_array_1->num_radix_blocks == L && lwe_array_2->num_radix_blocks == L,
                    "integer_comparison: the operands must hold the %u blocks the scratch was made for", L);
  HX_PANIC_IF_FALSE(lwe_array_1->lwe_dimension == lwe_array_2->lwe_dimension &&
                        lwe_array_out->lwe_dimension == lwe_array_1->lwe_dimension,
                    "input and output lwe dimension must be equal");
  const Params &p = m->cmp.ordering() ? m->cmp.prop.drv.p : m->cmp.eq.p;
  for (const CudaRadixCiphertextFFI *op : {lwe_array_1, lwe_array_2})
    if (op->degrees)
      for (uint32_t i = 0; i < L; ++i)
        HX_PANIC_IF_FALSE(op->degrees[i] <= p.msg - 1, "integer_comparison: block %u has degree %llu, the comparison takes "
                          "clean operands (propagate them first)", i, (unsigned long long)op->degrees[i]);
  const size_t w = (size_t)p.big_n + 1;
  uint64_t *out = (uint64_t *)lwe_array_out->ptr;
  const uint64_t *a = (const uint64_t *)lwe_array_1->ptr, *b = (const uint64_t *)lwe_array_2->ptr;
  if (m->op == MAX || m->op == MIN) {
    HX_PANIC_IF_FALSE(lwe_array_out->num_radix_blocks >= L, "integer_comparison: MAX / MIN need %u output blocks", L);
    m->cmp.run(streams, m->d_flag, a, b, ksks, bsks);  // a >= b
    m->mux.run(streams, out, m->d_flag, m->op == MAX ? a : b, m->op == MAX ? b : a, ksks, bsks);
    for (uint32_t i = 0; i < L; ++i) {
      if (lwe_array_out->degrees) lwe_array_out->degrees[i] = p.msg - 1;
      if (lwe_array_out->noise_levels) lwe_array_out->noise_levels[i] = 1;
    }
    return;
  }
  HX_PANIC_IF_FALSE(lwe_array_out->num_radix_blocks >= 1, "integer_comparison: the output needs a block");
  m->cmp.run(streams, out, a, b, ksks, bsks);
  if (lwe_array_out->num_radix_blocks > 1)
    HX_CHECK(hipMemsetAsync(out + w, 0, (size_t)(lwe_array_out->num_radix_blocks - 1) * w * sizeof(uint64_t), S0(streams)));
  for (uint32_t i = 0; i < lwe_array_out->num_radix_blocks; ++i) {
    if (lwe_array_out->degrees) lwe_array_out->degrees[i] = i == 0 ? 1 : 0;
    if (lwe_array_out->noise_levels) lwe_array_out->noise_levels[i] = i == 0 ? 1 : 0;
  }
}

// integer.h:254-274: the scalar comes as its clear blocks (one per radix block, least significant first, the caller stops at
// the last non-zero one: integer/gpu/server_key/radix/scalar_comparison.rs:129-157).  They become trivial ciphertexts (zero
// masks) in the scratch and take the ciphertext path: the same rounds, the same tables.
struct ScalarCompareScratch {
  static constexpr uint32_t kMagic = 0x53435231;  // "SCR1"
  uint32_t magic = kMagic;
  int8_t *inner = nullptr;  // a CompareScratch
  uint64_t *d_triv = nullptr;
  uint32_t blocks = 0, big_n = 0, msg = 0, carry = 0;
};
uint64_t scratch_cuda_integer_scalar_comparison_64_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                         CudaLweBootstrapKeyParamsFFI bsk_params,
                                                         CudaLweKeyswitchKeyParamsFFI ksk_params,
                                                         uint32_t lwe_ciphertext_count, uint32_t message_modulus,
                                                         uint32_t carry_modulus, enum COMPARISON_TYPE op_type, bool is_signed,
                                                         bool allocate_gpu_memory,
                                                         enum PBS_MS_REDUCTION_T noise_reduction_type) {
  HX_PANIC_IF_FALSE(mem_ptr != nullptr, "integer_scalar_comparison: null pointer");
  auto *m = new ScalarCompareScratch();
  uint64_t bytes = scratch_cuda_integer_comparison_64_async(streams, &m->inner, bsk_params, ksk_params, lwe_ciphertext_count,
                                                            message_modulus, carry_modulus, op_type, is_signed,
                                                            allocate_gpu_memory, noise_reduction_type);
  m->blocks = lwe_ciphertext_count;
  m->big_n = ksk_params.input_lwe_dimension;
  m->msg = message_modulus;
  m->carry = carry_modulus;
  const size_t sz = (size_t)lwe_ciphertext_count * (m->big_n + 1) * sizeof(uint64_t);
  if (allocate_gpu_memory) m->d_triv = (uint64_t *)scratch_alloc(sz);
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return bytes + sz;
}

void cuda_integer_scalar_comparison_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_out,
                                             CudaRadixCiphertextFFI const *lwe_array_in, void const *scalar_blocks,
                                             void const *h_scalar_blocks, int8_t *mem_ptr, void *const *bsks,
                                             void *const *ksks, uint32_t num_scalar_blocks) {
  first_gpu(streams);
  auto *m = reinterpret_cast<ScalarCompareScratch *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == ScalarCompareScratch::kMagic, "integer_scalar_comparison: foreign scratch pointer");
  HX_PANIC_IF_FALSE(m->d_triv != nullptr, "integer_scalar_comparison: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(lwe_array_in && lwe_array_in->ptr && (num_scalar_blocks == 0 || (scalar_blocks && h_scalar_blocks)),
                    "integer_scalar_comparison: null pointer");
  HX_PANIC_IF_FALSE(num_scalar_blocks <= m->blocks, "integer_scalar_comparison: %u scalar blocks for %u radix blocks (the caller "
                    "truncates the scalar to the ciphertext's length)", num_scalar_blocks, m->blocks);
  for (uint32_t i = 0; i < num_scalar_blocks; ++i)
    HX_PANIC_IF_FALSE(((const uint64_t *)h_scalar_blocks)[i] < m->msg, "integer_scalar_comparison: scalar block %u is %llu, not below "
                      "the message modulus", i, (unsigned long long)((const uint64_t *)h_scalar_blocks)[i]);
  const hipStream_t st = S0(streams);
  const uint32_t w = m->big_n + 1;
  HX_CHECK(hipMemsetAsync(m->d_triv, 0, (size_t)m->blocks * w * sizeof(uint64_t), st));
  if (num_scalar_blocks)
    HX_LAUNCH(lwe_body_add_scalars_kernel, dim3((num_scalar_blocks + 255) / 256), dim3(256), 0, st, m->d_triv,
              (const uint64_t *)scalar_blocks, w, num_scalar_blocks, ((uint64_t)1 << 63) / ((uint64_t)m->msg * m->carry));
  std::vector<uint64_t> deg(m->blocks, 0), noise(m->blocks, 0);
  for (uint32_t i = 0; i < num_scalar_blocks; ++i) deg[i] = ((const uint64_t *)h_scalar_blocks)[i];
  CudaRadixCiphertextFFI triv{m->d_triv, deg.data(), noise.data(), m->blocks, m->blocks, m->big_n};
  cuda_integer_comparison_64_async(streams, lwe_array_out, lwe_array_in, &triv, m->inner, bsks, ksks);
}

void cleanup_cuda_integer_scalar_comparison_64(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  auto *m = reinterpret_cast<ScalarCompareScratch *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == ScalarCompareScratch::kMagic, "cleanup integer_scalar_comparison: foreign scratch pointer");
  cleanup_cuda_integer_comparison_64(streams, &m->inner);
  if (m->d_triv) scratch_free(m->d_triv);
  m->magic = 0;
  delete m;
  *mem_ptr_void = nullptr;
}

void cleanup_cuda_integer_comparison_64(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  auto *m = reinterpret_cast<CompareScratch *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == CompareScratch::kMagic, "cleanup integer_comparison: foreign scratch pointer");
  m->cmp.release(streams);
  if (m->op == MAX || m->op == MIN) m->mux.release(streams);
  if (m->d_flag) scratch_free(m->d_flag);
  m->magic = 0;
  delete m;
  *mem_ptr_void = nullptr;
}

uint64_t scratch_cuda_cmux_64_async(CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
                                    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t lwe_ciphertext_count,
                                    uint32_t message_modulus, uint32_t carry_modulus, bool allocate_gpu_memory,
                                    enum PBS_MS_REDUCTION_T noise_reduction_type) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(mem_ptr != nullptr && lwe_ciphertext_count >= 1, "cmux: null pointer or no blocks");
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, (uint32_t)noise_reduction_type);
  t_dry = !allocate_gpu_memory;
  t_bytes = 0;
  auto *m = new CmuxMem();
  m->init(streams, p, lwe_ciphertext_count);
  m->size_only = t_dry;
  t_dry = false;
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return t_bytes;
}

void cuda_cmux_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array_out,
                        CudaRadixCiphertextFFI const *lwe_condition, CudaRadixCiphertextFFI const *lwe_array_true,
                        CudaRadixCiphertextFFI const *lwe_array_false, int8_t *mem_ptr, void *const *bsks,
                        void *const *ksks) {
  first_gpu(streams);
  auto *m = reinterpret_cast<CmuxMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == CmuxMem::kMagic, "cmux: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->size_only, "cmux: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(lwe_array_out && lwe_condition && lwe_array_true && lwe_array_false && lwe_array_out->ptr &&
                        lwe_condition->ptr && lwe_array_true->ptr && lwe_array_false->ptr && bsks && ksks,
                    "cmux: null pointer");
  const uint32_t L = m->blocks;
  HX_PANIC_IF_FALSE(lwe_array_true->num_radix_blocks == L && lwe_array_false->num_radix_blocks == L &&
                        lwe_array_out->num_radix_blocks >= L && lwe_condition->num_radix_blocks >= 1,
                    "cmux: the branches must hold the %u blocks the scratch was made for, the condition one block", L);
  HX_PANIC_IF_FALSE(lwe_array_true->lwe_dimension == lwe_array_false->lwe_dimension &&
                        lwe_array_out->lwe_dimension == lwe_array_true->lwe_dimension &&
                        lwe_condition->lwe_dimension == lwe_array_true->lwe_dimension,
                    "input and output lwe dimension must be equal");
  const Params &p = m->drv.p;
  for (const CudaRadixCiphertextFFI *op : {lwe_array_true, lwe_array_false})
    if (op->degrees)
      for (uint32_t i = 0; i < L; ++i)
        HX_PANIC_IF_FALSE(op->degrees[i] <= p.msg - 1, "cmux: block %u has degree %llu, the branches must be clean", i,
                          (unsigned long long)op->degrees[i]);
  if (lwe_condition->degrees)
    HX_PANIC_IF_FALSE(lwe_condition->degrees[0] <= 1, "cmux: the condition must be a boolean block (degree %llu)",
                      (unsigned long long)lwe_condition->degrees[0]);
  m->run(streams, (uint64_t *)lwe_array_out->ptr, (const uint64_t *)lwe_condition->ptr,
         (const uint64_t *)lwe_array_true->ptr, (const uint64_t *)lwe_array_false->ptr, ksks, bsks);
  for (uint32_t i = 0; i < L; ++i) {
    if (lwe_array_out->degrees) lwe_array_out->degrees[i] = p.msg - 1;
    if (lwe_array_out->noise_levels) lwe_array_out->noise_levels[i] = 1;
  }
}

void cleanup_cuda_cmux_64(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  auto *m = reinterpret_cast<CmuxMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == CmuxMem::kMagic, "cleanup cmux: foreign scratch pointer");
  m->release(streams);
  delete m;
  *mem_ptr_void = nullptr;
}

// ---- cuda/include/integer/integer.h:200-228 -------------------------------------------------------
uint64_t scratch_cuda_logical_scalar_shift_64_inplace_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                            CudaLweBootstrapKeyParamsFFI bsk_params,
                                                            CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks,
                                                            uint32_t message_modulus, uint32_t carry_modulus,
                                                            enum SHIFT_OR_ROTATE_TYPE shift_type, bool allocate_gpu_memory,
                                                            enum PBS_MS_REDUCTION_T noise_reduction_type) {
  first_gpu(streams);
  HX_PANIC_IF_FALSE(mem_ptr != nullptr && num_blocks >= 1, "logical_scalar_shift: null pointer or no blocks");
  HX_PANIC_IF_FALSE(shift_type == LEFT_SHIFT || shift_type == RIGHT_SHIFT,
                    "logical_scalar_shift: shift type %u is a rotation (scalar_rotate is not wired)", (uint32_t)shift_type);
  HX_PANIC_IF_FALSE((message_modulus & (message_modulus - 1)) == 0, "logical_scalar_shift: the message modulus must be a power of two");
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, (uint32_t)noise_reduction_type);
  t_dry = !allocate_gpu_memory;
  t_bytes = 0;
  auto *m = new ScalarShiftMem();
  m->blocks = num_blocks;
  m->bits = (uint32_t)__builtin_ctz(p.msg);
  m->left = shift_type == LEFT_SHIFT;
  const uint32_t msg = p.msg, bits = m->bits;
  const bool left = m->left != 0;
  std::vector<std::vector<uint64_t>> luts(std::max(1u, bits - 1), std::vector<uint64_t>((size_t)(p.k + 1) * p.N));
  for (uint32_t r = 1; r < bits; ++r)  // packed msg * current + neighbour (lower neighbour for a left shift, upper for a right one)
    generate_lut(p, luts[r - 1].data(), [msg, bits, r, left](uint64_t x) -> uint64_t {
      const uint64_t cur = x / msg, nb = x % msg;
      return (left ? ((cur << r) | (nb >> (bits - r))) : ((cur >> r) | (nb << (bits - r)))) % msg;
    });
  m->drv.init(streams, p, num_blocks, luts);
  radix_alloc((void **)&m->d_pack, (size_t)num_blocks * (p.big_n + 1) * sizeof(uint64_t));
  std::vector<uint64_t> li((size_t)std::max(1u, bits - 1) * num_blocks);
  for (size_t i = 0; i < li.size(); ++i) li[i] = i / num_blocks;
  m->d_lut_idx = dev_upload(S0(streams), li);
  m->size_only = t_dry;
  t_dry = false;
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return t_bytes;
}

void cuda_logical_scalar_shift_64_inplace_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array, uint32_t shift,
                                                int8_t *mem_ptr, void *const *bsks, void *const *ksks) {
  first_gpu(streams);
  auto *m = reinterpret_cast<ScalarShiftMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == ScalarShiftMem::kMagic, "logical_scalar_shift: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!m->size_only, "logical_scalar_shift: scratch was created with allocate_gpu_memory=false");
  HX_PANIC_IF_FALSE(lwe_array && lwe_array->ptr && bsks && ksks, "logical_scalar_shift: null pointer");
  const uint32_t L = m->blocks;
  HX_PANIC_IF_FALSE(lwe_array->num_radix_blocks >= L, "input does not have enough blocks");
  const Params &p = m->drv.p;
  if (lwe_array->degrees)
    for (uint32_t i = 0; i < L; ++i)
      HX_PANIC_IF_FALSE(lwe_array->degrees[i] <= p.msg - 1, "logical_scalar_shift: block %u has degree %llu, the shift takes "
                        "clean blocks", i, (unsigned long long)lwe_array->degrees[i]);
  if (shift == 0) return;
  const hipStream_t st = S0(streams);
  const uint32_t w = p.big_n + 1;
  uint64_t *v = (uint64_t *)lwe_array->ptr;
  auto clear = [&](uint32_t first, uint32_t count) {
    if (count == 0) return;
    HX_CHECK(hipMemsetAsync(v + (size_t)first * w, 0, (size_t)count * w * sizeof(uint64_t), st));
    for (uint32_t i = first; i < first + count; ++i) {
      if (lwe_array->degrees) lwe_array->degrees[i] = 0;
      if (lwe_array->noise_levels) lwe_array->noise_levels[i] = 0;
    }
  };
  if ((uint64_t)shift >= (uint64_t)m->bits * L) {  // every bit leaves
    clear(0, L);
    return;
  }
  const uint32_t q = shift / m->bits, r = shift % m->bits, n = L - q;  // n blocks survive
  const size_t bw = (size_t)w * sizeof(uint64_t);
  if (r == 0) {
    HX_CHECK(hipMemcpyAsync(m->d_pack, v + (m->left ? 0 : (size_t)q * w), n * bw, hipMemcpyDeviceToDevice, st));
    HX_CHECK(hipMemcpyAsync(v + (m->left ? (size_t)q * w : 0), m->d_pack, n * bw, hipMemcpyDeviceToDevice, st));
    if (lwe_array->degrees) {
      std::vector<uint64_t> d(lwe_array->degrees + (m->left ? 0 : q), lwe_array->degrees + (m->left ? 0 : q) + n);
      std::copy(d.begin(), d.end(), lwe_array->degrees + (m->left ? q : 0));
    }
    if (lwe_array->noise_levels) {
      std::vector<uint64_t> d(lwe_array->noise_levels + (m->left ? 0 : q), lwe_array->noise_levels + (m->left ? 0 : q) + n);
      std::copy(d.begin(), d.end(), lwe_array->noise_levels + (m->left ? q : 0));
    }
    clear(m->left ? 0 : n, q);
    return;
  }
  const uint64_t *lut = m->d_lut_idx + (size_t)(r - 1) * L;
  if (m->left) {
    // surviving block j (j = 0 .. n - 1) lands at position j + q: msg * b[j] + b[j - 1]
    axpy(st, m->d_pack, nullptr, v, nullptr, p.msg, nullptr, nullptr, w, 1);
    if (n > 1) axpy(st, m->d_pack + w, nullptr, v + w, nullptr, p.msg, v, nullptr, w, n - 1);
    m->drv.round(streams, v + (size_t)q * w, nullptr, m->d_pack, nullptr, lut, n, ksks, bsks);
  } else {
    // position j (j = 0 .. n - 1) takes msg * b[j + q] + b[j + q + 1]
    if (n > 1) axpy(st, m->d_pack, nullptr, v + (size_t)q * w, nullptr, p.msg, v + (size_t)(q + 1) * w, nullptr, w, n - 1);
    axpy(st, m->d_pack + (size_t)(n - 1) * w, nullptr, v + (size_t)(L - 1) * w, nullptr, p.msg, nullptr, nullptr, w, 1);
    m->drv.round(streams, v, nullptr, m->d_pack, nullptr, lut, n, ksks, bsks);
  }
  for (uint32_t i = (m->left ? q : 0); i < (m->left ? L : n); ++i) {
    if (lwe_array->degrees) lwe_array->degrees[i] = p.msg - 1;
    if (lwe_array->noise_levels) lwe_array->noise_levels[i] = 1;
  }
  clear(m->left ? 0 : n, q);
}

void cleanup_cuda_logical_scalar_shift_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  first_gpu(streams);
  auto *m = reinterpret_cast<ScalarShiftMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == ScalarShiftMem::kMagic, "cleanup logical_scalar_shift: foreign scratch pointer");
  HX_CHECK(hipStreamSynchronize(S0(streams)));
  m->drv.release(streams);
  for (uint64_t *d : {m->d_pack, m->d_lut_idx})
    if (d) scratch_free(d);
  m->magic = 0;
  delete m;
  *mem_ptr_void = nullptr;
}

// number of PBS one carry propagation / one multiplication of `num_blocks` blocks issues (for benches)
uint64_t hip_integer_propagate_pbs_count(uint32_t num_blocks) { return PropagateMem::pbs_count(num_blocks); }
uint64_t hip_integer_mult_pbs_count(int8_t *mem_ptr) {
  if (mem_ptr && reinterpret_cast<BoolMulMem *>(mem_ptr)->magic == BoolMulMem::kMagic)
    return reinterpret_cast<BoolMulMem *>(mem_ptr)->blocks;
  auto *m = reinterpret_cast<MulMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == MulMem::kMagic, "hip_integer_mult_pbs_count: foreign scratch pointer");
  uint64_t n = m->prod_slot.size();
  for (auto &s : m->steps)
    for (size_t g = 0; g < s.msg_slot.size(); ++g) n += 1 + (s.carry_slot[g] != ~(uint64_t)0);
  return n + PropagateMem::pbs_count(m->blocks);
}

}  // extern "C"
