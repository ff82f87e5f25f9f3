// hipemu.cpp — fiber scheduler for the host-side kernel emulation (TEST INFRASTRUCTURE ONLY,
// see hipemu.h).
#include "hipemu.h"
#include <omp.h>
// AddressSanitizer build of the emulation (make SAN=1: the CPU tier's stand-in for the reference's compute-sanitizer runs,
// scripts/check_memory_errors.sh): the sanitizer must be told about every fiber switch, or it takes a fiber's frames for
// garbage on the scheduler's stack
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define HIPEMU_ASAN 1
#else
#define HIPEMU_ASAN 0
#endif

namespace hipemu {
thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
thread_local ThreadCtx *g_cur = nullptr;
thread_local ucontext_t g_sched;
thread_local char *g_dyn_smem = nullptr;
thread_local uint64_t g_wave_xchg[32][64 * 4];
static thread_local const std::function<void()> *g_body = nullptr;

static constexpr size_t kStack = HIPEMU_ASAN ? 4096 * 1024 : 256 * 1024;  // instrumented frames are several times larger
#if HIPEMU_ASAN
static thread_local const void *g_sched_stack = nullptr;
static thread_local size_t g_sched_stack_size = 0;
static thread_local void *g_sched_fake = nullptr;
#endif

void yield_barrier(int kind) {
  ThreadCtx *me = g_cur;
  me->state = kind;
#if HIPEMU_ASAN
  __sanitizer_start_switch_fiber(&me->asan_fake, g_sched_stack, g_sched_stack_size);
#endif
  swapcontext(&me->ctx, &g_sched);
#if HIPEMU_ASAN
  __sanitizer_finish_switch_fiber(me->asan_fake, &g_sched_stack, &g_sched_stack_size);
#endif
  // resumed: restore my identity
  g_cur = me;
  g_threadIdx = me->tid;
}

static void fiber_entry() {
#if HIPEMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &g_sched_stack, &g_sched_stack_size);
#endif
  (*g_body)();
  g_cur->state = 3;
#if HIPEMU_ASAN
  __sanitizer_start_switch_fiber(nullptr, g_sched_stack, g_sched_stack_size);  // nullptr: this fiber does not come back
#endif
  swapcontext(&g_cur->ctx, &g_sched);
}

static void run_block(dim3 block, size_t smem, std::vector<ThreadCtx> &th, std::vector<char> &smem_buf) {
  const unsigned T = block.x * block.y * block.z;
  if (smem_buf.size() < smem + 64) smem_buf.resize(smem + 64);
  // 16-byte aligned dynamic LDS base, like the device
  g_dyn_smem = (char *)(((uintptr_t)smem_buf.data() + 15) & ~(uintptr_t)15);
  memset(g_dyn_smem, 0xA5, smem);  // poison: uninitialised LDS must not be relied upon
  for (unsigned t = 0; t < T; ++t) {
    ThreadCtx &c = th[t];
    if (!c.stack) c.stack = (char *)malloc(kStack);
    getcontext(&c.ctx);
    c.ctx.uc_stack.ss_sp = c.stack;
    c.ctx.uc_stack.ss_size = kStack;
    c.ctx.uc_link = nullptr;
    c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    c.state = 0;
    makecontext(&c.ctx, fiber_entry, 0);
  }
  unsigned done = 0;
  while (done < T) {
    bool progressed = false;
    for (unsigned t = 0; t < T; ++t) {
      ThreadCtx &c = th[t];
      if (c.state != 0) continue;
      g_cur = &c;
      g_threadIdx = c.tid;
#if HIPEMU_ASAN
      __sanitizer_start_switch_fiber(&g_sched_fake, c.stack, kStack);
#endif
      swapcontext(&g_sched, &c.ctx);
#if HIPEMU_ASAN
      __sanitizer_finish_switch_fiber(g_sched_fake, nullptr, nullptr);
#endif
      progressed = true;
      if (c.state == 3) ++done;
    }
    // release wave barriers
    for (unsigned w = 0; w * 64 < T; ++w) {
      unsigned lo = w * 64, hi = lo + 64 < T ? lo + 64 : T;
      bool all = true, any = false;
      for (unsigned t = lo; t < hi; ++t) {
        if (th[t].state == 2) any = true;
        else if (th[t].state != 3) all = false;
      }
      if (any && all) {
        for (unsigned t = lo; t < hi; ++t) if (th[t].state == 2) th[t].state = 0;
        progressed = true;
      }
    }
    // release the block barrier
    {
      bool all = true, any = false;
      for (unsigned t = 0; t < T; ++t) {
        if (th[t].state == 1) any = true;
        else if (th[t].state != 3) all = false;
      }
      if (any && all) {
        for (unsigned t = 0; t < T; ++t) if (th[t].state == 1) th[t].state = 0;
        progressed = true;
      }
    }
    if (!progressed) {
      fprintf(stderr, "hipemu: deadlock (divergent barrier) in block (%u,%u,%u)\n", g_blockIdx.x,
              g_blockIdx.y, g_blockIdx.z);
      abort();
    }
  }
}

void run_grid(dim3 grid, dim3 block, size_t smem, const std::function<void()> &body) {
  const long nblocks = (long)grid.x * grid.y * grid.z;
  const unsigned T = block.x * block.y * block.z;
#pragma omp parallel
  {
    std::vector<ThreadCtx> th(T);
    std::vector<char> smem_buf;
    g_body = &body;
    g_blockDim = block;
    g_gridDim = grid;
#pragma omp for schedule(dynamic)
    for (long b = 0; b < nblocks; ++b) {
      g_blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y),
                        (unsigned)(b / ((long)grid.x * grid.y)));
      run_block(block, smem, th, smem_buf);
    }
    for (auto &c : th) free(c.stack);
  }
}
}  // namespace hipemu
