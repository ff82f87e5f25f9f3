// arena.hip — the library's own stream-ordered allocator behind cuda_malloc_async / cuda_drop.
//
// The reference allocates and drops a CudaVec per operation (tfhe/src/core_crypto/gpu/vec.rs:94-150, Drop at :487-495)
// through cudaMallocAsync / cudaFree on the device's memory pool (tfhe-cuda-common/cuda/src/device.cu:176-226, 457-491):
// an allocation is an enqueue, not a device synchronisation.  hipMallocAsync's pool is not usable that way on this
// runtime (profiles/r04h_ks_cpp_diag*.txt), and a plain hipMalloc / hipFree pair synchronises the
// device twice per CudaVec.  This arena gives the reference's contract without the runtime's pool:
//
//   * blocks come from hipMalloc ONCE and are then recycled through per-device free lists by size class (powers of two
//     up to 1 MiB, multiples of 1 MiB above);
//   * a block dropped on the host is re-usable in STREAM ORDER: the drop records an event on the stream the block was
//     allocated for; the next owner's stream waits for that event (hipStreamWaitEvent) unless it is the same stream or
//     the event has already completed — nothing blocks the host, nothing synchronises the device.  As with
//     cudaFree of a cudaMallocAsync pointer, work on OTHER streams that still uses the block is the caller's to order
//     (the reference's wrappers synchronise their streams before a CudaVec is dropped);
//   * under stream capture no runtime allocator and no event of another timeline may be touched: a capturing stream
//     takes only blocks that were dropped on the same stream or are known to be idle, falls back to hipMalloc in
//     relaxed capture mode, and every block it touches stays with that stream for good (a graph replays its addresses),
//     re-usable only by later allocations of the same stream;
//   * memory goes back to the runtime when an allocation fails (trim and retry), on hip_backend_trim_allocator() and
//     when the owning stream is destroyed.
//
// Debug mode TFHE_HIP_ARENA_REDZONE=1 (the practice it stands in for: compute-sanitizer memcheck / racecheck over the
// reference's GPU tests, scripts/check_memory_errors.sh:1-60, Makefile:955-962): blocks are carved out of shared SLABS,
// so a block's neighbours are other live blocks of the library and its callers, not page padding; every block carries a
// 4 KiB canary in front of its payload and a canary from the LAST REQUESTED BYTE to 4 KiB past its size class; the canaries
// are written at every allocation and checked at every drop (cuda_drop, cleanup_*), a mismatch panics with the block's
// class, size, owner and the offset of the first foreign byte; a dropped payload is poisoned and the poison is checked
// when the block is handed out again (a write after the drop).  Every check synchronises the device: this mode is for
// the test suites, never for throughput.  The library's other device allocations (cuda_malloc, key staging, keyswitch
// planes and scratches) go through the arena too in this mode (device_alloc_sync).
#include "arena.h"
#include "kernels.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <map>
#include <deque>
#include <vector>

namespace tfhe_hip {
namespace {

struct FreeBlock {
  void *p;
  hipStream_t stream;  // the stream it was last owned on (nullptr: none / destroyed)
  hipEvent_t ready;    // recorded on `stream` at the drop; nullptr: idle (nothing can still be using it)
  bool pinned;         // touched by a capture of `stream`: that stream only, for good
  // events recorded at the drop on the OTHER streams cuda_create_stream_ffi made on this device: the reference's cuda_drop is a
  // cudaFree, which waits for the whole device — a vector that was also used on another stream of the library (another
  // stream_index of a CudaStreams set) is therefore ordered behind that work too, without blocking the host
  std::vector<std::pair<hipStream_t, hipEvent_t>> others;
};
struct LiveBlock {
  size_t cls, bytes;
  hipStream_t stream;
  bool pinned;
  bool scratch;  // the library's own scratch (no stream: idle when handed out, idle by contract when returned)
  bool armed;    // red-zone mode: canaries in place (not for a block handed out under stream capture: nothing may synchronise there)
};
struct Slab {
  char *base;
  size_t size, used;
};
constexpr size_t kRedzone = 4096, kSlabBytes = (size_t)64 << 20;
constexpr int kCanary = 0xA5, kPoison = 0xDD;
struct DeviceArena {
  std::mutex m;
  std::unordered_map<const void *, LiveBlock> live;
  std::map<size_t, std::vector<FreeBlock>> free_;
  std::unordered_set<hipStream_t> known_streams;  // made by cuda_create_stream_ffi and not destroyed yet: safe to touch at a drop
  std::deque<hipEvent_t> spare_events;  // taken from the front, returned to the back: a consumed event rests before its next record
  ArenaStats stats{};
  // red-zone mode
  std::vector<Slab> slabs;
  std::unordered_set<const void *> poisoned;  // free blocks whose payload holds the poison pattern
  unsigned long long *scan_result = nullptr;  // device word: offset of the first byte that is not the pattern
  uint64_t rz_checks = 0;
  uint64_t cache_cap = 0;  // bytes of idle blocks the arena keeps (0: not decided yet, see arena_free)
};
DeviceArena g_arena[16];

size_t size_class(size_t n) {
  if (n <= 256) return 256;
  if (n <= ((size_t)1 << 20)) {
    size_t c = 256;
    while (c < n) c <<= 1;
    return c;
  }
  return (n + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
}

hipEvent_t take_event(DeviceArena &a) {
  if (!a.spare_events.empty()) {
    hipEvent_t e = a.spare_events.front();
    a.spare_events.pop_front();
    return e;
  }
  hipEvent_t e;
  HX_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}

// hipMalloc, also from inside a stream capture (relaxed mode for the one call: the allocation is not part of the graph)
void *runtime_alloc(size_t bytes, bool capturing) {
  void *p = nullptr;
#if !defined(TFHE_HIPEMU)
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  if (capturing) HX_CHECK(hipThreadExchangeStreamCaptureMode(&mode));
  const hipError_t e = hipMalloc(&p, bytes);
  if (capturing) HX_CHECK(hipThreadExchangeStreamCaptureMode(&mode));
#else
  (void)capturing;
  const hipError_t e = hipMalloc(&p, bytes);
#endif
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

bool redzone_on() {
  static const bool on = [] {
    const char *e = std::getenv("TFHE_HIP_ARENA_REDZONE");
    return e != nullptr && std::atoi(e) != 0;
  }();
  return on;
}

// offset of the first byte of [p, p + n) that is not `pattern` (n if there is none)
__global__ void __launch_bounds__(256) arena_scan_kernel(const unsigned char *p, size_t n, unsigned pattern,
                                                         unsigned long long *first_bad) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if (p[i] != (unsigned char)pattern) {
      atomicMin(first_bad, (unsigned long long)i);
      break;
    }
}
size_t scan_for_foreign_byte(DeviceArena &a, const void *p, size_t n, int pattern) {
  if (n == 0) return 0;
  if (a.scan_result == nullptr) HX_CHECK(hipMalloc((void **)&a.scan_result, sizeof(unsigned long long)));
  unsigned long long v = (unsigned long long)n;
  HX_CHECK(hipMemcpy(a.scan_result, &v, sizeof(v), hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)((n + 256 * 64 - 1) / (256 * 64) < 4096 ? (n + 256 * 64 - 1) / (256 * 64) : 4096);
  HX_LAUNCH(arena_scan_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t) nullptr, (const unsigned char *)p, n,
            (unsigned)pattern, a.scan_result);
  HX_CHECK(hipDeviceSynchronize());
  HX_CHECK(hipMemcpy(&v, a.scan_result, sizeof(v), hipMemcpyDeviceToHost));
  return (size_t)v;
}
// a block of RZ + cls + RZ bytes out of the slabs (payload = block + RZ, 256-byte aligned)
void *slab_carve(DeviceArena &a, size_t cls, bool capturing) {
  const size_t need = kRedzone + cls + kRedzone;
  for (Slab &s : a.slabs)
    if (s.size - s.used >= need) {
      void *p = s.base + s.used + kRedzone;
      s.used += need;
      return p;
    }
  Slab s{nullptr, need > kSlabBytes ? need : kSlabBytes, 0};
  s.base = (char *)runtime_alloc(s.size, capturing);
  if (s.base == nullptr) return nullptr;
  s.used = need;
  a.slabs.push_back(s);
  return s.base + kRedzone;
}
// canaries around the `bytes` the caller asked for; synchronous (debug mode)
void redzone_arm(void *payload, size_t cls, size_t bytes) {
  HX_CHECK(hipDeviceSynchronize());
  HX_CHECK(hipMemsetAsync((char *)payload - kRedzone, kCanary, kRedzone, nullptr));
  HX_CHECK(hipMemsetAsync((char *)payload + bytes, kCanary, cls - bytes + kRedzone, nullptr));
  HX_CHECK(hipDeviceSynchronize());
}
void redzone_check(DeviceArena &a, const void *payload, const LiveBlock &lb, const char *when) {
  HX_CHECK(hipDeviceSynchronize());
  a.rz_checks++;
  const size_t front = scan_for_foreign_byte(a, (const char *)payload - kRedzone, kRedzone, kCanary);
  if (front != kRedzone)
    HX_PANIC("arena red zone (%s): block %p (class %zu, %zu bytes asked for, %s, stream %p) was written %zu bytes IN FRONT of "
             "its payload", when, payload, lb.cls, lb.bytes, lb.scratch ? "library scratch" : "cuda_malloc_async",
             (void *)lb.stream, kRedzone - front);
  const size_t tail = lb.cls - lb.bytes + kRedzone;
  const size_t back = scan_for_foreign_byte(a, (const char *)payload + lb.bytes, tail, kCanary);
  if (back != tail)
    HX_PANIC("arena red zone (%s): block %p (class %zu, %zu bytes asked for, %s, stream %p) was written %zu bytes PAST its "
             "last byte", when, payload, lb.cls, lb.bytes, lb.scratch ? "library scratch" : "cuda_malloc_async",
             (void *)lb.stream, back + 1);
}

// the events a drop recorded on the other streams of the device: completed ones go back to the spares; true if none is left
bool others_done(DeviceArena &a, FreeBlock &b, hipStream_t asking = nullptr) {
  for (size_t i = 0; i < b.others.size();) {
    if ((asking != nullptr && b.others[i].first == asking) || hipEventQuery(b.others[i].second) == hipSuccess) {
      a.spare_events.push_back(b.others[i].second);
      b.others.erase(b.others.begin() + i);
    } else {
      ++i;
    }
  }
  return b.others.empty();
}

// every idle, unpinned block goes back to the runtime; returns the bytes released
size_t trim_locked(DeviceArena &a) {
  size_t released = 0;
  if (redzone_on()) return 0;  // blocks are parts of slabs
  for (auto &kv : a.free_) {
    std::vector<FreeBlock> keep;
    for (FreeBlock &b : kv.second) {
      bool idle = b.ready == nullptr;
      if (!b.pinned && !others_done(a, b)) {
        keep.push_back(b);
        continue;
      }
      if (!idle && !b.pinned && hipEventQuery(b.ready) == hipSuccess) {
        a.spare_events.push_back(b.ready);
        b.ready = nullptr;
        idle = true;
      }
      if (idle && !b.pinned) {
        HX_CHECK(hipFree(b.p));
        released += kv.first;
      } else {
        keep.push_back(b);
      }
    }
    kv.second.swap(keep);
  }
  a.stats.cached_bytes -= released;
  return released;
}

}  // namespace

void *arena_alloc(int device, size_t bytes, hipStream_t stream, bool scratch) {
  HX_PANIC_IF_FALSE(device >= 0 && device < 16, "cuda_malloc_async: device index %d", device);
  DeviceArena &a = g_arena[device];
  const size_t cls = size_class(bytes ? bytes : 1);
  const bool capturing = stream != nullptr && stream_is_capturing(stream);
  std::lock_guard<std::mutex> lock(a.m);
  a.stats.allocations++;
  std::vector<FreeBlock> &fl = a.free_[cls];
  int pick = -1, pick_rank = 99;  // 0 same stream, 1 idle, 2 event completed, 3 must wait
  for (int i = (int)fl.size() - 1; i >= 0 && pick_rank > 0; --i) {
    FreeBlock &b = fl[i];
    int rank;
    if (!b.others.empty() && (capturing || !others_done(a, b, stream))) {
      if (capturing) continue;  // events of other timelines cannot be waited for inside a capture
      rank = 3;                 // the new owner waits for the other streams' work at the drop (and for the owner's, below)
    } else if (b.pinned) {
      if (!(capturing && b.stream == stream)) continue;  // a graph's address: its own stream's captures only
      rank = 0;
    } else if (b.stream == stream && b.stream != nullptr) {
      rank = 0;  // stream order does the waiting
    } else if (b.ready == nullptr) {
      rank = 1;
    } else if (capturing) {
      continue;  // an event of another timeline cannot be queried or waited for inside a capture
    } else {
      rank = hipEventQuery(b.ready) == hipSuccess ? 2 : 3;
    }
    if (rank < pick_rank) pick = i, pick_rank = rank;
  }
  void *p = nullptr;
  bool pinned = capturing;
  if (pick >= 0) {
    FreeBlock b = fl[pick];
    fl.erase(fl.begin() + pick);
    bool waited = false;
    if (b.ready != nullptr) {
      // rank 3 with the owner's own stream asking again: stream order does that part of the waiting
      if (pick_rank == 3 && !(b.stream == stream && b.stream != nullptr) && hipEventQuery(b.ready) != hipSuccess) {
        if (stream != nullptr) HX_CHECK(hipStreamWaitEvent(stream, b.ready, 0));  // the new owner's work queues behind the old owner's
        else HX_CHECK(hipEventSynchronize(b.ready));  // a scratch without a stream of its own: the host waits (rare: same-class block still busy)
        waited = true;
      }
      a.spare_events.push_back(b.ready);  // consumed (same stream: stream order did the waiting; a pinned block carries none)
    }
    for (const auto &se : b.others) {  // what the device's other streams had queued when the block was dropped
      if (se.first != stream && hipEventQuery(se.second) != hipSuccess) {  // (the asking stream's own event: stream order)
        if (stream != nullptr) HX_CHECK(hipStreamWaitEvent(stream, se.second, 0));
        else HX_CHECK(hipEventSynchronize(se.second));
        waited = true;
      }
      a.spare_events.push_back(se.second);
    }
    if (waited) a.stats.cross_stream_waits++;
    p = b.p;
    pinned = pinned || b.pinned;
    a.stats.cached_bytes -= cls;
    a.stats.reuses++;
  } else if (redzone_on()) {
    p = slab_carve(a, cls, capturing);
    HX_PANIC_IF_FALSE(p != nullptr, "cuda_malloc_async: out of device memory (%zu bytes requested, red-zone mode)", bytes);
    a.stats.runtime_allocations++;
  } else {
    p = runtime_alloc(cls, capturing);
    if (p == nullptr && !capturing) {  // out of memory: hand the cache back and try once more (hipFree is illegal under capture)
      trim_locked(a);
      p = runtime_alloc(cls, capturing);
    }
    HX_PANIC_IF_FALSE(p != nullptr, "cuda_malloc_async: out of device memory (%zu bytes requested)", bytes);
    a.stats.runtime_allocations++;
  }
  LiveBlock lb{cls, bytes, stream, pinned, scratch, false};
  if (redzone_on() && capturing) a.poisoned.erase(p);
  if (redzone_on() && !capturing) {
    lb.armed = true;
    if (a.poisoned.erase(p) != 0) {  // dropped earlier: nobody may have written it since
      HX_CHECK(hipDeviceSynchronize());
      const size_t bad = scan_for_foreign_byte(a, p, cls, kPoison);
      if (bad != cls)
        HX_PANIC("arena red zone: block %p (class %zu) was written at offset %zu AFTER it had been dropped", p, cls, bad);
    }
    redzone_arm(p, cls, bytes);
  }
  a.live[p] = lb;
  a.stats.live_bytes += cls;
  return p;
}

bool arena_free(int device, void *p, size_t *user_bytes) {
  if (device < 0 || device >= 16) return false;
  DeviceArena &a = g_arena[device];
  std::lock_guard<std::mutex> lock(a.m);
  auto it = a.live.find(p);
  if (it == a.live.end()) return false;
  const LiveBlock lb = it->second;
  a.live.erase(it);
  if (user_bytes) *user_bytes = lb.bytes;
  FreeBlock fb{p, lb.stream, nullptr, lb.pinned, {}};
  if (redzone_on() && !(lb.stream != nullptr && a.known_streams.count(lb.stream) != 0 && stream_is_capturing(lb.stream))) {
    if (lb.armed) redzone_check(a, p, lb, "drop");
    HX_CHECK(hipMemsetAsync(p, kPoison, lb.cls, nullptr));
    HX_CHECK(hipDeviceSynchronize());
    a.poisoned.insert(p);
  }
  // A drop has no stream argument: the block's owner stream is touched only if the library made it and still knows it alive
  // (cuda_create_stream_ffi / cuda_destroy_stream).  A caller's own stream may be gone by now — recording an event on a dead
  // handle is undefined in the runtime — so its blocks come back after ONE device synchronisation, idle and nobody's.
  const bool known = lb.stream != nullptr && a.known_streams.count(lb.stream) != 0;
  const bool capturing = known && stream_is_capturing(lb.stream);
  if (capturing) {
    fb.pinned = true;  // dropped inside the capture: free at that point of the graph's timeline, for this stream only
  } else if (lb.stream != nullptr && !known) {
    HX_CHECK(hipDeviceSynchronize());
    fb.stream = nullptr;
    fb.pinned = false;
  } else if (!lb.pinned && lb.stream != nullptr) {
    fb.ready = take_event(a);
    HX_CHECK(hipEventRecord(fb.ready, lb.stream));  // everything queued on the owner's stream so far may still use the block
  } else if (!lb.pinned && !lb.scratch) {
    // cuda_malloc_async(size, NULL, gpu): the caller's work runs on the legacy default stream — order behind it like behind
    // any other owner (a library scratch, also stream-less, is idle by contract and takes no event)
    fb.ready = take_event(a);
    HX_CHECK(hipEventRecord(fb.ready, nullptr));
  }
  if (!capturing && !fb.pinned && !lb.scratch && !(lb.stream != nullptr && !known)) {
    // ... and the device's other library streams (none of them capturing: an event recorded inside a capture belongs to it)
    for (hipStream_t s : a.known_streams) {
      if (s == lb.stream || stream_is_capturing(s)) continue;
      if (hipStreamQuery(s) == hipSuccess) continue;  // idle: nothing it has queued can still touch the block
      (void)hipGetLastError();                         // (hipErrorNotReady is not an error here)
      hipEvent_t e = take_event(a);
      HX_CHECK(hipEventRecord(e, s));
      fb.others.emplace_back(s, e);
    }
  }
  a.free_[lb.cls].push_back(fb);
  a.stats.live_bytes -= lb.cls;
  a.stats.cached_bytes += lb.cls;
  a.stats.frees++;
  // the cache is bounded (the reference's pool has a release threshold, device.cu:70-120): beyond the cap every idle block
  // goes back to the runtime (hipFree synchronises the device — rare by construction)
  // Default: a quarter of the device's memory, at least 16 GiB (72 GB on an MI355X) — sized for the machine: one batch of 128
  // FheUint64 multiplications leaves more than 16 GiB of scratch in the size classes, and with the round-5 cap of 16 GiB every
  // second multiplication of a loop paid the trim and the runtime allocations again (1.79 s instead of 1.21 s per 128,
  // docs/history/round_6_log.md section 9).  TFHE_HIP_ARENA_CACHE_MB overrides it.
  if (a.cache_cap == 0) {
    const char *e = std::getenv("TFHE_HIP_ARENA_CACHE_MB");
    if (e) {
      a.cache_cap = (uint64_t)std::strtoull(e, nullptr, 10) << 20;
    } else {
      hipDeviceProp_t prop;
      HX_CHECK(hipGetDeviceProperties(&prop, device));
      a.cache_cap = std::max<uint64_t>((uint64_t)16384 << 20, (uint64_t)prop.totalGlobalMem / 4);
    }
    if (a.cache_cap == 0) a.cache_cap = 1;  // TFHE_HIP_ARENA_CACHE_MB=0: nothing stays cached
  }
  if (a.stats.cached_bytes > a.cache_cap && !capturing) trim_locked(a);  // (hipFree is illegal while this thread's stream captures)
  return true;
}

void arena_register_stream(int device, hipStream_t stream) {
  if (device < 0 || device >= 16) return;
  DeviceArena &a = g_arena[device];
  std::lock_guard<std::mutex> lock(a.m);
  a.known_streams.insert(stream);
}

// the stream is about to be destroyed (already synchronised): its blocks are idle and nobody's
void arena_release_stream(int device, hipStream_t stream) {
  if (device < 0 || device >= 16) return;
  DeviceArena &a = g_arena[device];
  std::lock_guard<std::mutex> lock(a.m);
  a.known_streams.erase(stream);
  for (auto &kv : a.live)
    if (kv.second.stream == stream) kv.second.stream = nullptr, kv.second.pinned = false;
  for (auto &kv : a.free_)
    for (FreeBlock &b : kv.second)
      if (b.stream == stream) {
        if (b.ready != nullptr) a.spare_events.push_back(b.ready);
        b.ready = nullptr;
        b.stream = nullptr;
        b.pinned = false;  // the graphs of a destroyed stream cannot be launched on it again
      }
}

size_t arena_trim(int device) {
  HX_PANIC_IF_FALSE(device >= 0 && device < 16, "hip_backend_trim_allocator: device index %d", device);
  DeviceArena &a = g_arena[device];
  std::lock_guard<std::mutex> lock(a.m);
  return trim_locked(a);
}

bool arena_enabled() {
  static const bool on = [] {
    const char *e = std::getenv("TFHE_HIP_MALLOC_ASYNC");
    return e == nullptr || !std::strcmp(e, "arena");
  }();
  return on;
}

// The library's own scratch_* / cleanup_* pairs (the reference: cuda_malloc_with_size_tracking_async /
// cuda_drop_with_size_tracking_async on the scratch's stream, e.g. integer/integer_utilities.h).  Blocks of no particular
// stream: handed out only when idle, and idle by contract when they come back (every cleanup_* synchronises its stream first).
void *scratch_alloc(size_t bytes) {
  int dev = 0;
  HX_CHECK(hipGetDevice(&dev));
  if (arena_enabled()) return arena_alloc(dev, bytes, nullptr, true);
  void *p = nullptr;
  HX_CHECK(hipMalloc(&p, bytes));
  return p;
}
// The block is looked up on the current device first, then on every other one: a cleanup_* may run on a host thread whose
// current device is not the scratch's (a multi-GPU host alternating devices) — handing an arena block to hipFree would
// leave a stale entry in its owner's map.
void scratch_free(void *p) {
  if (p == nullptr) return;
  int dev = 0;
  HX_CHECK(hipGetDevice(&dev));
  if (arena_free(dev, p, nullptr)) return;
  for (int d = 0; d < 16; ++d)
    if (d != dev && arena_free(d, p, nullptr)) return;
  HX_CHECK(hipFree(p));
}
// hipMalloc / hipFree semantics (the free synchronises the device) for the library's staging buffers, keyswitch planes and
// scratches and for cuda_malloc: the runtime's own calls, except in red-zone mode, where they are arena blocks as well
void *device_alloc_sync(size_t bytes) {
  if (arena_enabled() && redzone_on()) return scratch_alloc(bytes);
  void *p = nullptr;
  HX_CHECK(hipMalloc(&p, bytes));
  return p;
}
void device_free_sync(void *p) {
  if (p == nullptr) return;
  if (arena_enabled() && redzone_on()) {
    HX_CHECK(hipDeviceSynchronize());
    scratch_free(p);
    return;
  }
  HX_CHECK(hipFree(p));
}
// red-zone mode: every process says at its end what it checked (a finding would have aborted it)
namespace {
struct RedzoneReport {
  ~RedzoneReport() {
    if (!redzone_on()) return;
    uint64_t checks = 0, slabs = 0, bytes = 0;
    for (DeviceArena &a : g_arena) {
      checks += a.rz_checks;
      slabs += a.slabs.size();
      for (const Slab &s : a.slabs) bytes += s.used;
    }
    std::fprintf(stderr, "[arena red zone] %llu blocks checked at their drop, %llu slabs, %llu bytes carved: no finding\n",
                 (unsigned long long)checks, (unsigned long long)slabs, (unsigned long long)bytes);
  }
} g_redzone_report;
}  // namespace
uint64_t arena_redzone_checks(int device) {
  if (device < 0 || device >= 16) return 0;
  DeviceArena &a = g_arena[device];
  std::lock_guard<std::mutex> lock(a.m);
  return redzone_on() ? a.rz_checks : 0;
}

ArenaStats arena_stats(int device) {
  HX_PANIC_IF_FALSE(device >= 0 && device < 16, "hip_backend_allocator_stats: device index %d", device);
  DeviceArena &a = g_arena[device];
  std::lock_guard<std::mutex> lock(a.m);
  return a.stats;
}

}  // namespace tfhe_hip
