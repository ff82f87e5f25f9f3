// profile.h — named ranges around the radix layer's rounds for rocprofv3 --marker-trace (TFHE_HIP_PROFILE=1).
//
// The reference brackets apply-LUT / scatter / gather with NVTX ranges (tfhe-cuda-common/cuda/include/helper_profile.cuh:1-17,
// cuda/src/integer/integer.cuh:874,958,981): a trace of an FheUint64 multiplication is dozens of keyswitch + bootstrap
// launches, unreadable without them.  Here: roctx ranges (rocprofiler-sdk-roctx, looked up with dlopen the first time a range
// is pushed — the library does not link against the profiler), compiled in always, active only with TFHE_HIP_PROFILE=1 in the
// environment (one relaxed load per range otherwise).
#pragma once
#include <cstdint>
#include <cstdio>

namespace tfhe_hip {
bool profile_on();
void profile_push(const char *name);
void profile_pop();
uint64_t profile_range_count();
struct ProfileRange {
  bool on;
  explicit ProfileRange(const char *name) : on(profile_on()) {
    if (on) profile_push(name);
  }
  template <class... A>
  ProfileRange(const char *fmt, A... a) : on(profile_on()) {
    if (on) {
      char buf[128];
      std::snprintf(buf, sizeof(buf), fmt, a...);
      profile_push(buf);
    }
  }
  ~ProfileRange() {
    if (on) profile_pop();
  }
  ProfileRange(const ProfileRange &) = delete;
  ProfileRange &operator=(const ProfileRange &) = delete;
};
}  // namespace tfhe_hip
#define HX_RANGE_CAT2(a, b) a##b
#define HX_RANGE_CAT(a, b) HX_RANGE_CAT2(a, b)
#define HX_RANGE(...) tfhe_hip::ProfileRange HX_RANGE_CAT(hx_range_, __LINE__)(__VA_ARGS__)
