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
};
struct LiveBlock {
  size_t cls, bytes;
  hipStream_t stream;
  bool pinned;
};
struct DeviceArena {
  std::mutex m;
  std::unordered_map<const void *, LiveBlock> live;
  std::map<size_t, std::vector<FreeBlock>> free_;
  std::unordered_set<hipStream_t> known_streams;  // made by cuda_create_stream_ffi and not destroyed yet: safe to touch at a drop
  std::deque<hipEvent_t> spare_events;  // taken from the front, returned to the back: a consumed event rests before its next record
  ArenaStats stats{};
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

// every idle, unpinned block goes back to the runtime; returns the bytes released
size_t trim_locked(DeviceArena &a) {
  size_t released = 0;
  for (auto &kv : a.free_) {
    std::vector<FreeBlock> keep;
    for (FreeBlock &b : kv.second) {
      bool idle = b.ready == nullptr;
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

void *arena_alloc(int device, size_t bytes, hipStream_t stream) {
  HX_PANIC_IF_FALSE(device >= 0 && device < 16, "cuda_malloc_async: device index %d", device);
  DeviceArena &a = g_arena[device];
  const size_t cls = size_class(bytes ? bytes : 1);
  const bool capturing = stream != nullptr && stream_is_capturing(stream);
  std::lock_guard<std::mutex> lock(a.m);
  a.stats.allocations++;
  std::vector<FreeBlock> &fl = a.free_[cls];
  int pick = -1, pick_rank = 99;  // 0 same stream, 1 idle, 2 event completed, 3 must wait
  for (int i = (int)fl.size() - 1; i >= 0 && pick_rank > 0; --i) {
    const FreeBlock &b = fl[i];
    int rank;
    if (b.pinned) {
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
    if (b.ready != nullptr) {
      if (pick_rank == 3) {
        if (stream != nullptr) HX_CHECK(hipStreamWaitEvent(stream, b.ready, 0));  // the new owner's work queues behind the old owner's
        else HX_CHECK(hipEventSynchronize(b.ready));  // a scratch without a stream of its own: the host waits (rare: same-class block still busy)
        a.stats.cross_stream_waits++;
      }
      a.spare_events.push_back(b.ready);  // consumed (same stream: stream order did the waiting; a pinned block carries none)
    }
    p = b.p;
    pinned = pinned || b.pinned;
    a.stats.cached_bytes -= cls;
    a.stats.reuses++;
  } else {
    p = runtime_alloc(cls, capturing);
    if (p == nullptr) {  // out of memory: hand the cache back and try once more
      trim_locked(a);
      p = runtime_alloc(cls, capturing);
      HX_PANIC_IF_FALSE(p != nullptr, "cuda_malloc_async: out of device memory (%zu bytes requested)", bytes);
    }
    a.stats.runtime_allocations++;
  }
  a.live[p] = LiveBlock{cls, bytes, stream, pinned};
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
  FreeBlock fb{p, lb.stream, nullptr, lb.pinned};
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
  }
  a.free_[lb.cls].push_back(fb);
  a.stats.live_bytes -= lb.cls;
  a.stats.cached_bytes += lb.cls;
  a.stats.frees++;
  // the cache is bounded (the reference's pool has a release threshold, device.cu:70-120): beyond the cap every idle block
  // goes back to the runtime (hipFree synchronises the device — rare by construction)
  static const uint64_t cap = [] {
    const char *e = std::getenv("TFHE_HIP_ARENA_CACHE_MB");
    return (uint64_t)(e ? std::strtoull(e, nullptr, 10) : 16384) << 20;
  }();
  if (a.stats.cached_bytes > cap) trim_locked(a);
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
  if (arena_enabled()) return arena_alloc(dev, bytes, nullptr);
  void *p = nullptr;
  HX_CHECK(hipMalloc(&p, bytes));
  return p;
}
void scratch_free(void *p) {
  if (p == nullptr) return;
  int dev = 0;
  HX_CHECK(hipGetDevice(&dev));
  if (!arena_free(dev, p, nullptr)) HX_CHECK(hipFree(p));
}

ArenaStats arena_stats(int device) {
  DeviceArena &a = g_arena[device];
  std::lock_guard<std::mutex> lock(a.m);
  return a.stats;
}

}  // namespace tfhe_hip
