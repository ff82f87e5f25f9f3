// hx.h — platform header of the MI355X backend.
//
// Product build: hipcc --offload-arch=gfx950 (HIP runtime, CDNA4 device code).
// TFHE_HIPEMU build (tests/emu only): the same sources compiled by g++ against the host-side
// kernel-language emulation so the kernel logic can be checked without a GPU.  The emulation
// is test infrastructure; it is never a fallback of the product library.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(TFHE_HIPEMU)
#include "hipemu.h"
#include "hipemu_runtime.h"
#define HX_WAVE_SYNC() hx_wave_sync_emu()
#define HX_UNROLL _Pragma("GCC unroll 64")
#define HX_NO_UNROLL _Pragma("GCC unroll 1")
#define HX_SCHED_FENCE() do { } while (0)
#define HX_OPAQUE(v) do { } while (0)
#define HX_LAUNDER(v) do { } while (0)
#define HX_OPAQUE_S(v) do { } while (0)
#define HX_UNIFORM(v) (v)
// 16 bytes per lane from global memory straight into LDS at (wave-uniform base) + lane * 16
#define HX_GLOBAL_TO_LDS16(gsrc, lds_wave_base, lane) __builtin_memcpy((char *)(lds_wave_base) + (lane) * 16, (gsrc), 16)
#define HX_BLOCK_SYNC_LDS() __syncthreads()
#else
#include <hip/hip_runtime.h>
// a launch that the runtime rejects (dynamic LDS over the limit, bad grid, missing code object) aborts like every
// other misuse of the boundary (the reference: check_cuda_error(cudaGetLastError()) after each launch);
// hipGetLastError is legal during stream capture
#define HX_LAUNCH(kern, grid, block, smem, stream, ...)                 \
  do {                                                                  \
    hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__);   \
    HX_CHECK_LAUNCH(#kern);                                             \
  } while (0)
#define HX_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
// Lanes of one wave exchanging data through LDS need no s_barrier (they execute in lock
// step); they do need the compiler not to reorder the LDS accesses across this point.
#define HX_WAVE_SYNC()                                   \
  do {                                                   \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
#define HX_UNROLL _Pragma("unroll")
#define HX_NO_UNROLL _Pragma("unroll 1")
// compile-time scheduling fence: keeps the instruction scheduler from hoisting loads of a later
// chunk across this point (bounds live ranges, hence VGPR pressure)
#define HX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// makes the compiler forget what it knows about a VGPR value: address arithmetic derived from it
// is recomputed where it is used instead of being hoisted out of the loop and kept (or spilled)
#define HX_OPAQUE(v) asm volatile("" : "+v"(v))
// same, but free to move: only hides the value's origin from the optimiser's pattern matching
#define HX_LAUNDER(v) asm("" : "+v"(v))
// the same fence for a wave-uniform value that lives in a scalar register
#define HX_OPAQUE_S(v) asm volatile("" : "+s"(v))
// wave-uniform value into an SGPR
#define HX_UNIFORM(v) __builtin_amdgcn_readfirstlane(v)
// 16 bytes per lane from global memory straight into LDS at (wave-uniform base) + lane * 16 (global_load_lds_dwordx4:
// no staging registers, no ds_write; counted on vmcnt)
#define HX_GLOBAL_TO_LDS16(gsrc, lds_wave_base, lane)                                                       \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc),                  \
                                   (__attribute__((address_space(3))) void *)(lds_wave_base), 16, 0, 0)
// workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, not for its
// outstanding global loads (__syncthreads() drains vmcnt too, which would expose the latency of key
// loads that were deliberately issued early)
#define HX_BLOCK_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

#define HX_DEV __device__ __forceinline__

// ---- per-lane select by a wave-uniform lane mask (bit l <-> lane l) that lives in a scalar register pair:
// one v_cndmask, no vector compare
#if defined(TFHE_HIPEMU)
static inline uint32_t hx_select_by_lane_mask(uint64_t mask, uint32_t if_set, uint32_t if_clear) {
  return ((mask >> (threadIdx.x & 63)) & 1) ? if_set : if_clear;
}
#else
__device__ __forceinline__ uint32_t hx_select_by_lane_mask(uint64_t mask, uint32_t if_set, uint32_t if_clear) {
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
  return r;
}
#endif

// ---- loads through a buffer resource: wave-uniform base in scalar registers + a 32-bit per-lane byte offset.
// A per-lane index into a uniform array costs two vector instructions this way; as a 64-bit pointer the
// compiler builds every address with a shift-add and a carry pair.
#if defined(TFHE_HIPEMU)
struct HxBuffer {
  const char *base;
};
static inline HxBuffer hx_make_buffer(const void *base, uint32_t) { return HxBuffer{(const char *)base}; }
static inline uint64_t hx_buffer_load_u64(HxBuffer b, uint32_t byte_offset) {
  uint64_t v;
  __builtin_memcpy(&v, b.base + byte_offset, 8);
  return v;
}
struct hx_f64x2 {
  double x, y;
};
template <int AUX = 0>
static inline hx_f64x2 hx_buffer_load_f64x2(HxBuffer b, uint32_t lane_byte_offset, uint32_t uniform_byte_offset) {
  hx_f64x2 v;
  __builtin_memcpy(&v, b.base + lane_byte_offset + uniform_byte_offset, 16);
  return v;
}
template <int AUX = 0>
static inline void hx_buffer_load_u64x2(HxBuffer b, uint32_t lane_byte_offset, uint32_t uniform_byte_offset, uint64_t &x, uint64_t &y) {
  uint64_t v[2];
  __builtin_memcpy(v, b.base + lane_byte_offset + uniform_byte_offset, 16);
  x = v[0];
  y = v[1];
}
template <int AUX = 0>
static inline void hx_buffer_store_u64x2(HxBuffer b, uint32_t lane_byte_offset, uint32_t uniform_byte_offset, uint64_t x, uint64_t y) {
  const uint64_t v[2] = {x, y};
  __builtin_memcpy(const_cast<char *>(b.base) + lane_byte_offset + uniform_byte_offset, v, 16);
}
#else
struct HxBuffer {
  __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ HxBuffer hx_make_buffer(const void *base, uint32_t bytes) {
  // word 3 = 0x00020000: raw (untyped) buffer on gfx90a/gfx94x/gfx950; stride 0, range check against `bytes`
  return HxBuffer{__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ uint64_t hx_buffer_load_u64(HxBuffer b, uint32_t byte_offset) {
  typedef unsigned int hx_u32x2 __attribute__((ext_vector_type(2)));
  const hx_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(b.rsrc, (int)byte_offset, 0, 0);
  return ((uint64_t)v.y << 32) | v.x;
}
// 16 bytes per lane at (per-lane byte offset in ONE vector register) + (wave-uniform byte offset in a scalar
// register): the address of every load of an unrolled sweep costs at most one scalar add and no vector registers
struct hx_f64x2 {
  double x, y;
};
template <int AUX = 0>
__device__ __forceinline__ hx_f64x2 hx_buffer_load_f64x2(HxBuffer b, uint32_t lane_byte_offset,
                                                         uint32_t uniform_byte_offset) {
  typedef unsigned int hx_u32x4 __attribute__((ext_vector_type(4)));
  const hx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.rsrc, (int)lane_byte_offset, (int)uniform_byte_offset, AUX);
  hx_f64x2 r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}
// 16 bytes per lane as two 64-bit words, load and store (the split-key engine's accumulator round trip: one vector register
// of lane offsets for the whole sweep instead of a 64-bit pointer per 4 KB window of immediate offsets)
// AUX: the instruction's cache-policy bits (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ void hx_buffer_load_u64x2(HxBuffer b, uint32_t lane_byte_offset, uint32_t uniform_byte_offset, uint64_t &x,
                                                     uint64_t &y) {
  typedef unsigned int hx_u32x4 __attribute__((ext_vector_type(4)));
  const hx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.rsrc, (int)lane_byte_offset, (int)uniform_byte_offset, AUX);
  x = ((uint64_t)v.y << 32) | v.x;
  y = ((uint64_t)v.w << 32) | v.z;
}
template <int AUX = 0>
__device__ __forceinline__ void hx_buffer_store_u64x2(HxBuffer b, uint32_t lane_byte_offset, uint32_t uniform_byte_offset, uint64_t x,
                                                      uint64_t y) {
  typedef unsigned int hx_u32x4 __attribute__((ext_vector_type(4)));
  hx_u32x4 v;
  v.x = (uint32_t)x;
  v.y = (uint32_t)(x >> 32);
  v.z = (uint32_t)y;
  v.w = (uint32_t)(y >> 32);
  __builtin_amdgcn_raw_buffer_store_b128(v, b.rsrc, (int)lane_byte_offset, (int)uniform_byte_offset, AUX);
}
#endif

// ---- cross-lane swaps of gfx950: v_permlane32_swap exchanges the upper 32 lanes of `a` with the lower 32 lanes
// of `b`; v_permlane16_swap exchanges the odd 16-lane rows of `a` with the even rows of `b`.  Applied to the
// dwords of two registers they transpose (register select) x (lane bit 5, resp. bit 4).
#if defined(TFHE_HIPEMU)
static inline void hx_permlane_swap(uint32_t &a, uint32_t &b, int lane_bit) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *x = (uint32_t *)hipemu::g_wave_xchg[wave];
  x[lane * 2] = a;
  x[lane * 2 + 1] = b;
  hipemu::yield_barrier(2);
  const int bit = (lane >> lane_bit) & 1, partner = lane ^ (1 << lane_bit);
  const uint32_t na = bit ? x[partner * 2 + 1] : a;   // upper part of a <- lower part of b
  const uint32_t nb = bit ? b : x[partner * 2];       // lower part of b <- upper part of a
  hipemu::yield_barrier(2);
  a = na;
  b = nb;
}
static inline void hx_permlane32_swap(uint32_t &a, uint32_t &b) { hx_permlane_swap(a, b, 5); }
static inline void hx_permlane16_swap(uint32_t &a, uint32_t &b) { hx_permlane_swap(a, b, 4); }
#else
__device__ __forceinline__ void hx_permlane32_swap(uint32_t &a, uint32_t &b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
__device__ __forceinline__ void hx_permlane16_swap(uint32_t &a, uint32_t &b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
#endif

// ---- one lane's 32-bit value as a wave-uniform scalar (v_readlane_b32 with a literal lane: one instruction)
#if defined(TFHE_HIPEMU)
static inline uint32_t hx_readlane(uint32_t v, int src_lane) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *x = (uint32_t *)hipemu::g_wave_xchg[wave];
  x[lane] = v;
  hipemu::yield_barrier(2);
  const uint32_t r = x[src_lane];
  hipemu::yield_barrier(2);
  return r;
}
#else
__device__ __forceinline__ uint32_t hx_readlane(uint32_t v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
#endif

// ---- int8 matrix core: D(32x32, i32) = A(32x32, i8) * B(32x32, i8) + C, one instruction per wave.
// Lane l supplies 16 bytes of row (l & 31) of A and of column (l & 31) of B, both for the same 16 values of
// k (the half selected by l >> 5); it receives D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
// in v[r], r = 0..15 (cdna_hip_programming.md, "Fragment layout").
struct hx_i8x16 {
  int32_t w[4];
};
struct hx_i32x16 {
  int32_t v[16];
};
#if defined(TFHE_HIPEMU)
static inline hx_i32x16 hx_mfma_i32_32x32x32_i8(const hx_i8x16 a, const hx_i8x16 b, hx_i32x16 c) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t *x = hipemu::g_wave_xchg[wave];
  memcpy(&x[lane * 4], &a, 16);
  memcpy(&x[lane * 4 + 2], &b, 16);
  hipemu::yield_barrier(2);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    int32_t acc = c.v[r];
    for (int h = 0; h < 2; ++h) {
      const int8_t *pa = (const int8_t *)&x[(h * 32 + row) * 4];
      const int8_t *pb = (const int8_t *)&x[(h * 32 + col) * 4 + 2];
      for (int j = 0; j < 16; ++j) acc += (int32_t)pa[j] * (int32_t)pb[j];
    }
    c.v[r] = acc;
  }
  hipemu::yield_barrier(2);
  return c;
}
#else
__device__ __forceinline__ hx_i32x16 hx_mfma_i32_32x32x32_i8(const hx_i8x16 a, const hx_i8x16 b, hx_i32x16 c) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef int v16i __attribute__((ext_vector_type(16)));
  v4i va = {a.w[0], a.w[1], a.w[2], a.w[3]}, vb = {b.w[0], b.w[1], b.w[2], b.w[3]};
  v16i vc;
#pragma unroll
  for (int i = 0; i < 16; ++i) vc[i] = c.v[i];
  vc = __builtin_amdgcn_mfma_i32_32x32x32_i8(va, vb, vc, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) c.v[i] = vc[i];
  return c;
}
#endif

#include <cstdio>
#include <cstdlib>

// Error convention of the boundary we replace: misuse => message on stderr + abort()
// (backends/tfhe-cuda-common/cuda/include/device.h:13-41).
#define HX_PANIC(format, ...)                                                            \
  do {                                                                                   \
    std::fprintf(stderr, "%s::%d::%s: panic.\n" format "\n", __FILE__, __LINE__, __func__, \
                 ##__VA_ARGS__);                                                         \
    std::abort();                                                                        \
  } while (0)
#define HX_PANIC_IF_FALSE(cond, format, ...)                        \
  do {                                                              \
    if (!(cond)) HX_PANIC(format "\n\n %s\n", ##__VA_ARGS__, #cond); \
  } while (0)
#define HX_CHECK_LAUNCH(name)                                                                    \
  do {                                                                                           \
    hipError_t hx_le_ = hipGetLastError();                                                       \
    if (hx_le_ != hipSuccess) {                                                                  \
      std::fprintf(stderr, "HIP launch error: %s (%s) %s %d\n", hipGetErrorString(hx_le_), name, \
                   __FILE__, __LINE__);                                                          \
      std::abort();                                                                              \
    }                                                                                            \
  } while (0)
#define HX_CHECK(ans)                                                                     \
  do {                                                                                    \
    hipError_t hx_code_ = (ans);                                                          \
    if (hx_code_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "HIP error: %s %s %d\n", hipGetErrorString(hx_code_), __FILE__, \
                   __LINE__);                                                             \
      std::abort();                                                                       \
    }                                                                                     \
  } while (0)

// Dynamic-LDS limit of a kernel, raised once per (kernel, device) instead of on every launch: the attribute call
// takes the runtime's device lock and showed up as host time between back-to-back launches.
#include <atomic>
template <auto Kernel>
static inline void hx_set_dynamic_smem_once(size_t bytes) {
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  HX_CHECK(hipGetDevice(&dev));
  const uint64_t bit = (uint64_t)1 << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  HX_CHECK(hipFuncSetAttribute((const void *)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  done.fetch_or(bit, std::memory_order_release);
}
