// profile.hip — roctx ranges behind TFHE_HIP_PROFILE=1 (profile.h)
#include "profile.h"
#include "hx.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>

namespace tfhe_hip {
namespace {
typedef int (*push_fn)(const char *);
typedef int (*pop_fn)();
push_fn g_push = nullptr;
pop_fn g_pop = nullptr;
std::once_flag g_once;
std::atomic<uint64_t> g_ranges{0};
void resolve() {
  // the SDK's marker library first (what rocprofv3 --marker-trace intercepts), the legacy roctx as a second choice
  for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
    void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) continue;
    g_push = (push_fn)dlsym(h, "roctxRangePushA");
    g_pop = (pop_fn)dlsym(h, "roctxRangePop");
    if (g_push != nullptr && g_pop != nullptr) return;
    g_push = nullptr;
    g_pop = nullptr;
  }
  if (std::getenv("TFHE_HIP_PROFILE_QUIET") == nullptr)
    std::fprintf(stderr, "TFHE_HIP_PROFILE=1: no roctx library found (ranges are counted, not emitted)\n");
}
}  // namespace

bool profile_on() {
  static const bool on = [] {
    const char *e = std::getenv("TFHE_HIP_PROFILE");
    return e != nullptr && std::atoi(e) != 0;
  }();
  return on;
}
void profile_push(const char *name) {
  std::call_once(g_once, resolve);
  g_ranges.fetch_add(1, std::memory_order_relaxed);
  if (g_push) g_push(name);
}
void profile_pop() {
  if (g_pop) g_pop();
}
uint64_t profile_range_count() { return g_ranges.load(std::memory_order_relaxed); }
}  // namespace tfhe_hip
