// hipemu.h — minimal host-side emulation of the HIP kernel language, TEST INFRASTRUCTURE ONLY.
//
// Purpose: compile the product's kernel sources (tfhe_rs_amd/csrc/*.hip) with g++ and run
// them on the CPU so that `pytest -m "not gpu"` can check the kernels' LOGIC bit-for-bit
// against the oracle in a container without a GPU.  It is never linked into, loaded by, or
// used as a fallback for the product library (libtfhe_hip_backend.so aborts without a GPU).
//
// Model: one block at a time; the block's threads are ucontext fibers.  __syncthreads()
// and the wave-level sync (hx_wave_sync / shuffles) yield to a scheduler that releases a
// barrier once every live thread of the block (resp. wave) has arrived.  64-lane waves.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include <ucontext.h>
#include <vector>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static thread_local  // one block at a time per host thread: per-block storage

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipemu {
struct ThreadCtx {
  ucontext_t ctx;
  char *stack = nullptr;
  dim3 tid;
  int state = 0;  // 0 runnable, 1 at block barrier, 2 at wave barrier, 3 done
  void *asan_fake = nullptr;  // AddressSanitizer builds (make SAN=1): the fiber's fake-stack handle across switches
};
extern thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern thread_local ThreadCtx *g_cur;
extern thread_local ucontext_t g_sched;
extern thread_local char *g_dyn_smem;
extern thread_local uint64_t g_wave_xchg[32][64 * 4];  // per-wave exchange area for cross-lane collectives
void yield_barrier(int kind);
void run_grid(dim3 grid, dim3 block, size_t smem, const std::function<void()> &body);
}  // namespace hipemu

#define threadIdx (hipemu::g_threadIdx)
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)

static inline void __syncthreads() { hipemu::yield_barrier(1); }
static inline void hx_wave_sync_emu() { hipemu::yield_barrier(2); }
// blocks of a grid run on different host threads: a real atomic
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned int atomicAdd(unsigned int *p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) {
  unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
static inline void __threadfence_block() {}
static inline void __threadfence() {}

static inline uint64_t __umul64hi(uint64_t a, uint64_t b) {
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
}
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint32_t __brev(uint32_t x) {
  uint32_t r = 0;
  for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
  return r;
}

// kernel launch: HX_LAUNCH(kernel, grid, block, smem_bytes, stream, args...)
#define HX_LAUNCH(kern, grid, block, smem, stream, ...) \
  hipemu::run_grid((grid), (block), (smem), [&]() { kern(__VA_ARGS__); })
#define HX_DYN_SMEM(name) char *name = hipemu::g_dyn_smem
