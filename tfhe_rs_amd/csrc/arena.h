// arena.h — stream-ordered device allocator behind cuda_malloc_async / cuda_drop (arena.hip)
#pragma once
#include "hx.h"
#include <cstddef>
#include <cstdint>

namespace tfhe_hip {

struct ArenaStats {
  uint64_t allocations, reuses, runtime_allocations, frees, cross_stream_waits, live_bytes, cached_bytes;
};
// a block of at least `bytes` for work queued on `stream` from now on (never null: panics when the device is full)
void *arena_alloc(int device, size_t bytes, hipStream_t stream, bool scratch = false);
// false: `p` is not an arena block (the caller frees it its own way); *user_bytes = the size it was asked for
bool arena_free(int device, void *p, size_t *user_bytes);
void arena_register_stream(int device, hipStream_t stream);  // cuda_create_stream_ffi: a stream the arena may touch at a drop
void arena_release_stream(int device, hipStream_t stream);
size_t arena_trim(int device);  // idle blocks back to the runtime; bytes released
ArenaStats arena_stats(int device);
bool arena_enabled();  // TFHE_HIP_MALLOC_ASYNC unset or "arena"
// scratch of the library's own scratch_* / cleanup_* pairs, on the current device: arena blocks of no particular stream when
// the arena is on (idle when handed out, idle by contract when returned), hipMalloc / hipFree otherwise
void *scratch_alloc(size_t bytes);
void scratch_free(void *p);  // whichever device's arena the block belongs to
// hipMalloc / hipFree (a free that synchronises the device) — arena blocks too in red-zone mode (TFHE_HIP_ARENA_REDZONE=1)
void *device_alloc_sync(size_t bytes);
void device_free_sync(void *p);
uint64_t arena_redzone_checks(int device);  // canary checks made so far (0 when the mode is off)

}  // namespace tfhe_hip
