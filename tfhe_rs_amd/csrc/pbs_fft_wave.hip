// pbs_fft_wave.hip — throughput PBS kernel for N = 2048, k = 1 (PARAM_MESSAGE_2_CARRY_2 and
// every other (n, l, base_log) on that ring).
//
// Replaces backends/tfhe-cuda-backend/cuda/src/pbs/programmable_bootstrap_classic.cuh:214-745
// (the "specialized 2_2" kernels).  Same results, bit for bit, as pbs_generic.hip and the
// oracle: the butterfly dataflow of DESIGN.md §4 is kept, only its grouping changes.
//
// CDNA4 mapping
//   * one WAVE per polynomial: a GLWE (k+1 = 2 polynomials) is a wave pair, a workgroup is 8
//     waves = 4 LWEs, one workgroup per CU (2 waves per SIMD, <= 256 VGPRs each).
//   * the torus accumulator (2048 u64 per polynomial) lives in VGPRs for all n iterations
//     (32 u64 per lane); the 1024-point complex transform holds 16 points per lane.
//   * every transform is 3 register passes (radix 16, 4, 16): the exchange between the first two moves lane
//     bits (5,4) with v_permlane32_swap / v_permlane16_swap, the one between the last two is a wave-private
//     LDS transposition — lanes of one wave run in lock step, so neither needs an s_barrier.  Padded slot
//     layouts make every ds_read_b128/ds_write_b128 bank-conflict free.
//   * the only cross-wave traffic is the exchange of the two forward transforms before the
//     multiply-accumulate (both rows are read from LDS; the role of a wave is a scalar pointer choice);
//     the pair synchronises through two LDS flags (no workgroup barrier, so the four LWEs of a CU drift
//     apart and overlap their LDS- and VALU-heavy phases).
//   * the issue priority of a wave rises with the phase of its iteration (s_setprio): the wave that is
//     further along runs through while its SIMD mate fills the gaps.
//   * integer side written against the measured issue costs (tools/microbench2.hip): the registers hold
//     MINUS the accumulator so that the rotate-and-subtract is xor / one 64-bit add / xor, one-level digits
//     are a two-instruction rounding with an exact per-lane fallback at the -B/2 boundary, the torus
//     conversion keeps its constants in registers.
//   * twiddles whose index depends on the lane sit in a 23 KB LDS table shared by the 8 waves (strided
//     sets stored contiguously); wave-uniform ones are broadcast reads.
//   * the bootstrap key is stored in the order this kernel consumes it (lane-contiguous
//     16-byte elements), so each of the 32 key loads per wave-iteration is one fully
//     coalesced 1 KiB request; all workgroups stream GGSW_i at about the same time, so the
//     60 MB key is served from L2 / Infinity Cache.
#include "kernels.h"
#include <type_traits>

// tuning knobs (tools/build_variants.py rebuilds this file with other values)
#ifndef WAVE_EARLY_CHUNKS
#define WAVE_EARLY_CHUNKS 1  // key chunks requested at the top of an iteration (0, 1 or 2); the rest at the MAC
#endif
// Issue priority of a wave by phase of its CMUX iteration (s_setprio, 0..3).  The two waves that share a
// SIMD belong to different LWEs; with equal priority they interleave instruction by instruction and tend
// to reach their LDS round trips and pair waits together.  Raising the priority as the iteration advances
// lets the wave that is further along run through (its partner wave is waiting for it), while the other
// one fills the gaps: measured 102.6 k -> 117 k PBS/s; the order matters (digits < forward, inverse >= MAC),
// see profiles/r01_setprio_variants.txt.
#ifndef WAVE_MAC_PREFETCH
#define WAVE_MAC_PREFETCH 0  // measured: 0 -> 117.9 k, 2 -> 117.4 k, 4 -> 117.4 k, 8 -> 117.2 k PBS/s (the partner wave already covers the LDS latency)
#endif
#ifndef WAVE_PRIO_A
#define WAVE_PRIO_A 0  // rotation + digits
#endif
#ifndef WAVE_PRIO_B
#define WAVE_PRIO_B 1  // forward transform
#endif
#ifndef WAVE_PRIO_C
#define WAVE_PRIO_C 2  // pair exchange + MAC
#endif
#ifndef WAVE_PRIO_D
#define WAVE_PRIO_D 3  // inverse transform, conversion, accumulation
#endif
#ifndef WAVE_PRIO_MB_A
#define WAVE_PRIO_MB_K 0  // multi-bit: keybundle build (key streaming)
#define WAVE_PRIO_MB_A 1
#define WAVE_PRIO_MB_B 2
#define WAVE_PRIO_MB_C 3
#define WAVE_PRIO_MB_D 3
#endif
// optional finer steps inside the transforms (-1: keep the phase's level)
#ifndef WAVE_PRIO_F3
#define WAVE_PRIO_F3 -1
#endif
#ifndef WAVE_PRIO_I3
#define WAVE_PRIO_I3 -1
#endif
#ifndef WAVE_PRIO_CONV
#define WAVE_PRIO_CONV -1
#endif
#if !defined(TFHE_HIPEMU)
#define HX_PRIO_OPT(p) do { if ((p) >= 0) __builtin_amdgcn_s_setprio((p) < 0 ? 0 : (p)); } while (0)
#else
#define HX_PRIO_OPT(p) do { } while (0)
#endif
#if !defined(TFHE_HIPEMU)
#define HX_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define HX_PRIO(p) do { } while (0)
#endif
#ifndef WAVE_FLAG_SLEEP
#define WAVE_FLAG_SLEEP 1  // s_sleep argument between two polls of a pair flag
#endif
#ifndef WAVE_MB_PTS
#define WAVE_MB_PTS 2   // multi-bit: points per lane and row of one keybundle step (16 / PTS chunks per level)
#endif
#ifndef WAVE_MB_PACE
#define WAVE_MB_PACE 1  // multi-bit: groups a wave pair may run ahead of the slowest pair of its XCD, plus 1 (0: no pacing)
#endif
#ifndef WAVE_MB_PACE_OCTET
// ... in OCTET mode (0: none).  With a wave working for all four LWEs of its workgroup and the touch of the next group's key
// lines below, the workgroups of an XCD do better unpaced: the first to arrive at a line fetches it for the others
// (g = 4 / g = 3 per 4096, same box: paced 20.8 / 36.5 ms, unpaced 20.6 / 35.9)
#define WAVE_MB_PACE_OCTET 0
#endif
#ifndef WAVE_MB_PACE_AT_KEY
// multi-bit pacing: 1 = a wave waits for its XCD right in front of the group's first key request (digits and forward transform
// do not touch the key: they run under the wait for the slower workgroups) and reports a group as soon as its last key
// request is out; 0 = wait at the top of the group, report at its end (rounds 3-5)
#define WAVE_MB_PACE_AT_KEY 0
#endif
#ifndef WAVE_MB_PACE_SLEEP
#define WAVE_MB_PACE_SLEEP 2  // s_sleep argument between two polls of the XCD's counter (64 cycles each)
#endif
#ifndef WAVE_MB_PACE_SPINS
#define WAVE_MB_PACE_SPINS 4096  // polls (with s_sleep) before a wave gives up pacing for the rest of the launch
#endif
#ifndef WAVE_MB_TURNS
#define WAVE_MB_TURNS 0  // SHARE: the two quads of a workgroup take the multiply-accumulate in turns (measured slower: 60 vs 54 ms)
#endif
#ifndef WAVE_MB_SHARE_SETS
// SHARE: register sets in rotation (a request = the 2 rows of one point and subset); 0: two for one level, three for
// several (one box, g = 3 / g = 4 per 4096: 2 -> 48.8 / 29.8 ms, 3 -> 43.8 / 30.3, 4 -> 43.0 / 30.7, 5 -> 43.9 / 31.3)
#define WAVE_MB_SHARE_SETS 0
#endif
#ifndef WAVE_MB_OCTET
#define WAVE_MB_OCTET 2  // four LWEs per workgroup: the eight waves share every key load (1: one-level sets only, 2: also the sets with a compile-time level count >= 2)
#endif
#ifndef WAVE_MB_OCTET_K_FIRST
// OCTET: 1 = the keybundle (which depends on the mask and the key only) is combined BEFORE the barrier that publishes the
// eight transforms, so the waves meet once per group (barrier, products, barrier) and run free in between
#define WAVE_MB_OCTET_K_FIRST 0
#endif
#ifndef WAVE_MB_PREFETCH
#define WAVE_MB_PREFETCH 1  // multi-bit (pair and quad modes): a load per wave and group touches the next group's key lines
#endif
#ifndef WAVE_MB_EXPERIMENT
#define WAVE_MB_EXPERIMENT 0  // timing experiments (wrong results): bit 0 = every base request reads one of 16 rows of the table, bit 1 = no scalar root loads, bit 2 = no base requests (OCTET)
#endif
#if WAVE_MB_EXPERIMENT & 1
#define MB_EXP_ROW(d) ((d) & 15u)
#else
#define MB_EXP_ROW(d) (d)
#endif
#ifndef WAVE_MB_OCTET2_K_FIRST
// OCTET, several levels: 1 = the level's barrier behind the first point's keybundle (which needs the key and the mask only)
// instead of in front of it: a wave that is early combines instead of waiting (g = 3 per 4096, same box: 36.77 -> 36.04 ms
// and 35.71 -> 35.47, profiles/r06_ab_multibit.txt section 10)
#define WAVE_MB_OCTET2_K_FIRST 1
#endif
#ifndef WAVE_MB_PF_DIST
#define WAVE_MB_PF_DIST 1  // ... of the group this many groups ahead
#endif
#ifndef WAVE_MB_PF_POS
#define WAVE_MB_PF_POS 0  // ... issued 0: in front of the inverse transform, 1: at the top of the group
#endif
#ifndef WAVE_MB_PREFETCH_OCTET
// the same touch in OCTET mode: 0 never, 1 always, 2 for the sets with several levels only (g = 3, two levels: 40.3 -> 36.5 ms
// per 4096; g = 4, one level: 20.9 -> 21.2)
#define WAVE_MB_PREFETCH_OCTET 2
#endif
#ifndef WAVE_MB_OCTET_SETS
#define WAVE_MB_OCTET_SETS 3  // OCTET: register sets in rotation (a request = the 2 rows of one point and subset)
#endif
#ifndef WAVE_MB_SETS
#define WAVE_MB_SETS 4  // multi-bit: register sets in rotation (SETS - 1 key requests in flight)
#endif
#ifndef WAVE_MB_BASES
// multi-bit monomial bases requested 0: once per group (held across the digit and transform phases: the two-level g = 3
// kernel then spills 55 registers, 8 dword stores + loads per wave and level), 1: per level ahead of the level's key
// requests (no spills), 2: per level behind the first key requests (no spills); -1: 1 for several levels, 0 for one.
// With the bases as 64 scattered table entries per request (rounds 2-3) 0 won everywhere (g = 3 / g = 4: 0 -> 46.5 /
// 33.4 ms, 1 -> 48.5 / 35.4, 2 -> 48.2 / 35.6: the gathers in front of the multiply-accumulate cost more than the
// spills); with the lane-order table (one coalesced 1 KB request, tables.h mono_lane) g = 3 / g = 4 on one box:
// 0 -> 45.1 / 30.2 ms, 1 -> 43.8 / 30.4, 2 -> 43.8 / 30.6: the two-level kernel takes 1 (and spills nothing).
#define WAVE_MB_BASES -1
#endif
#ifndef WAVE_MB_W16_SCALAR
// multi-bit: the 16th roots of unity that complete the monomial factors (wave-uniform index) come through the scalar
// data cache (1) instead of as broadcast reads of the LDS table (0): 2 (2^g - 1) reads per point fewer — at g = 4 as
// many LDS instructions as the whole rest of a group
#define WAVE_MB_W16_SCALAR 1
#endif
#ifndef WAVE_MB_ROOT_JIT
// multi-bit, scalar 16th roots: 1 = the degree is fenced (HX_OPAQUE_S) right where its root is fetched, so the address of every
// root of a level (2 (2^g - 1) x 8 or 16 pointers) is computed just in time instead of all at once at the top of the level —
// hoisted, they do not fit the scalar file and travel through vector-register lanes (g = 3 quad kernel: 337 spilled SGPRs,
// ~700 v_writelane / v_readlane per group)
#define WAVE_MB_ROOT_JIT 1
#endif
#ifndef WAVE_SPLIT_LWES
#define WAVE_SPLIT_LWES 4   // exact engine, split-key form: LWEs per workgroup (the accumulators of a CU's LWEs live in L2)
#endif
#ifndef WAVE_STAGGER
// classic one-level loop: waves 4..7 (the second wave of every SIMD) start the loop this many times 4096 cycles after
// waves 0..3, so that the two waves of a SIMD are in different phases of the CMUX (0: off).  Same box, ms per 4096,
// eight interleaved pairs of runs (profiles/r04_ab_stagger*.txt): 0 -> 33.50 / 32.40, 1 -> 33.33 / 32.20 (-0.5 %),
// 2 -> 32.50, 3 -> 32.24
#define WAVE_STAGGER 1
#endif
#ifndef WAVE_SPLIT_PROBE
#define WAVE_SPLIT_PROBE 0  // timing probes of the split-key loop (wrong results): see the uses
#endif
#ifndef WAVE_SPLIT_PACE
// exact engine, split-key form: CMUXes a wave pair may run ahead of the slowest pair of its XCD, plus 1 (0: no pacing).
// Unpaced, the 32 workgroups of an XCD drift apart over the 918 iterations and each pulls its own 256 KB key slice
// through an L2 that the CU-resident accumulators (4 MB per XCD) already fill
#define WAVE_SPLIT_PACE 0
#endif
#ifndef WAVE_FUSE_PASS1
#define WAVE_FUSE_PASS1 1    // first inverse pass interleaved with the MAC chunks
#endif
#ifndef WAVE_UNIFORM_LITERALS
// 1: the twiddles that are the same in every lane and every launch (forward d = 0..3, inverse half = 4, 8: 12 of the
// 118 16-byte LDS reads of an iteration) are literals of the instruction stream (scalar moves) instead of broadcast
// reads of the LDS table; the inverse butterflies whose twiddle is 1 or -i lose their two products (same roundings:
// fma(x, 1, y) = x + y).  The values are checked against the host tables when the tables are built (tables.hip).
#define WAVE_UNIFORM_LITERALS 1
#endif
// ... per loop: the classic one-level loop (measured: 34.6 -> 33.3 ms per 4096 with the resident twiddles below), the
// multi-bit loops, the split-key exact engine (measured slower with literals: 150.3 -> 156.0 ms per 4096 — the kernel
// already spills, the literals' scalar moves add to it)
#ifndef WAVE_LIT_MB
#define WAVE_LIT_MB 2  // 1: plain literals, 2: literals made where they are used (lit_cplx)
#endif
#ifndef WAVE_SPLIT_EARLY_RESTORE
// split-key engine: handshakes posted early / waited for late (see the limb step).  Same box, ms per 4096: 115.28 -> 114.73
// (profiles/r05_ab_split_restore.txt) — the pair waits are cheap, as in the classic loop:
#define WAVE_SPLIT_EARLY_RESTORE 1
#endif
#ifndef WAVE_DEFER_DONE
// classic one-level loop: the "done" wait in front of the inverse transposition's first store instead of behind the products:
// 32.17 -> 32.16 ms per 4096, three interleaved rounds (profiles/r05_ab_classic_defer.txt): not kept
#define WAVE_DEFER_DONE 0
#endif
#ifndef WAVE_CLASSIC_SYNC
// classic loop: the same barrier (0: none).  Same box, ms per 4096, three interleaved rounds: none 33.27, every 4 / 16 / 64:
// 34.30 / 33.42 / 33.27 (profiles/r05_ab_classic_sync.txt) — its 60 MB key stays in L2 either way
#define WAVE_CLASSIC_SYNC 0
#endif
#ifndef WAVE_SPLIT_SYNC
// split-key engine: a bare workgroup barrier every so many mask elements (0: none) keeps the four LWEs of a workgroup on the
// same key rows.  Same box, ms per 4096, two interleaved rounds: none 121.0, every 1 / 4 / 8 / 16 / 32: 119.4 / 118.6 /
// 119.1 / 118.9 / 118.7 (profiles/r05_ab_split_sync.txt); de-phasing the upper four waves behind the barrier (the classic
// loop's WAVE_STAGGER) by 1 / 2 / 4 s_sleep(64): 117.8 -> 118.7 / 118.3 / 118.8
#define WAVE_SPLIT_SYNC 16
#endif
#ifndef WAVE_SPLIT_TAIL_ASM
#define WAVE_SPLIT_TAIL_ASM 1  // split-key engine: the accumulator update as arith.h's six-instruction sequence
#endif
#ifndef WAVE_LIT_LIMBS
#define WAVE_LIT_LIMBS 0
#endif
#ifndef WAVE_ROT_PAIRS
// 1: the rotation fetches the staged words of two consecutive register rows (r, r + 1: 512 bytes apart) with ONE
// two-address LDS read off one computed address; the second address may run 512 bytes past the 16 KiB ring, where the
// staging keeps a copy of the ring's first 512 bytes (the exchange buffer has 1,008 spare bytes behind the ring).
// 16 LDS instructions and 16 address computations fewer per CMUX.  0: one read and one address per word.
#define WAVE_ROT_PAIRS 1
#endif
#ifndef WAVE_RESIDENT
#define WAVE_RESIDENT 2  // classic one-level loop: twiddles kept in registers across the iterations (ResidentTwiddles: 0..2)
#endif

#ifndef WAVE_PROBE_TS
// measurement builds only (tools/cumask_sweep.hip, variants/lib_ts.so): every workgroup records where it ran
// (HW_ID, XCC_ID) and the 100 MHz / shader-clock timestamps around its CMUX loop; hip_probe_wave_timestamps() sets the
// record buffer and can pin the LWEs per workgroup.  Never defined in the product library.
#define WAVE_PROBE_TS 0
#endif

#ifndef WAVE_MB_PROBE
// measurement builds only (tools/mb_phase_probe.py): every wave of the multi-bit loop sums the shader-clock cycles of its
// phases (s_memtime at the phase boundaries: the waits for its own LDS / scalar traffic fall into the phase that issued
// them) and writes the eight sums to the record buffer of hip_probe_wave_timestamps().  Never defined in the product library.
#define WAVE_MB_PROBE 0
#endif

namespace tfhe_hip {
#if WAVE_PROBE_TS || WAVE_MB_PROBE
__device__ uint64_t *g_wave_ts = nullptr;
static unsigned g_wave_force_per_block = 0;
extern "C" void hip_probe_wave_timestamps(uint64_t *dev_records, uint32_t lwes_per_block) {
  HX_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_wave_ts), &dev_records, sizeof(dev_records)));
  g_wave_force_per_block = lwes_per_block;
}
#endif
namespace wavek {

constexpr int N = 2048, n = 1024, LOG2N2 = 12;
constexpr int LWES_PER_BLOCK = 4, WAVES = 8, TPB = WAVES * 64;

// One wave's exchange buffer: 1024 complex points in 16-byte slots, PADDED so that every
// transpose is (one VGPR base) + (compile-time offset) and every ds_read/write_b128 is bank
// conflict free:
//   P_A(q) = q + 4*(q >> 6)   M1 <-> M2 transposes   (1084 slots)
//   P_B(q) = q + (q >> 4)     M2 <-> M3 transposes and the pair's F exchange (1087 slots)
// with M1: q = r*64 + lane ; M2: q = hi4*64 + r*4 + lo2 (lane = hi4*4 + lo2) ; M3: q = lane*16 + r
//   M1/P_A : slot = lane + 68*r                      M2/P_A : slot = (hi4*68 + lo2) + 4*r
//   M2/P_B : slot = (hi4*68 + lo2) + 4*r + (r >> 2)  M3/P_B : slot = lane*17 + r
// (slot mod 16 = (lane + const) mod 16 in all four forms, distinct inside every b128 lane group)
constexpr int BUF_SLOTS = 1087;
constexpr int BUF_BYTES = BUF_SLOTS * 16;

// compact LDS twiddle table (entries of 16 bytes), shared by the 8 waves
constexpr int T_F1 = 0;      // forward d = 0..3, even groups: fwd[1], fwd[2], fwd[4], fwd[6], fwd[8..14 step 2]
constexpr int T_F2 = 8;      // forward d = 4..7: 16 + 16 + 32 + 64
constexpr int T_F3 = 136;    // forward d = 8, 9, even groups: 128 + 256, each stored [j][lane]
constexpr int T_INV = 520;   // E[J] = inv[512 + J], J < 256; every inverse twiddle is E[j*512/half] or -i*E[.]
constexpr int T_U = 776;     // untwist, j <= 512 (mirrored above): 513 entries
constexpr int T_F6 = 1289;   // forward d = 6, all 64 groups: fwd[64 + x]
// contiguous copies of the strided E[] reads with the worst bank conflicts (one entry per distinct lane value)
constexpr int T_E8 = 1353;   // E[4 x], x < 64    (inverse half = 128)
constexpr int T_W7 = 1417;   // E[8 x], x < 32    (inverse half = 64)
constexpr int T_E32 = 1449;  // E[16 x], x < 16   (inverse half = 32)
constexpr int T_W16 = 1465;  // E[32 x], x < 8    (inverse half = 16)
constexpr int T_W16X = 1473; // 16th roots of unity e^{2 pi i t / 16} = mono[t N/8] (multi-bit monomial factors)
constexpr int T_TOTAL = 1489;
constexpr int FLAGS_BYTES = 64;
constexpr size_t SMEM_BYTES = (size_t)WAVES * BUF_BYTES + (size_t)T_TOTAL * 16 + FLAGS_BYTES;

// wave-uniform twiddles as literals (WAVE_UNIFORM_LITERALS): fwd[1], fwd[2], fwd[4], fwd[6], fwd[8..14 step 2] and
// E[64 j] = inv[512 + 64 j] of the N = 2048 tables (long-double angles rounded once, tables.hip)
constexpr double LIT_F1[8][2] = {{0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1}, {0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2},
                                 {0x1.f6297cff75cbp-1, 0x1.8f8b83c69a60bp-3},  {0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1},
                                 {0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4}, {0x1.44cf325091dd6p-1, 0x1.8bc806b151741p-1},
                                 {0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63bp-2}, {0x1.294062ed59f06p-2, 0x1.e9f4156c62ddap-1}};
constexpr double LIT_E64[4][2] = {{1.0, 0.0}, {0x1.d906bcf328d46p-1, -0x1.87de2a6aea963p-2},
                                  {0x1.6a09e667f3bcdp-1, -0x1.6a09e667f3bcdp-1}, {0x1.87de2a6aea963p-2, -0x1.d906bcf328d46p-1}};
// A literal twiddle whose scalar registers are made where it is used (LIT = 2): the two dwords are OR-ed with a scalar zero that
// the compiler cannot see through (HX_OPAQUE_S on it at the top of the transform), so the moves are not hoisted out of the
// group loop — hoisted, the twelve literals of the two transforms hold 48 scalar registers for the whole kernel (multi-bit
// OCTET kernels: 48 / 74 spilled SGPRs, a v_readlane per use).  LIT = 1: plain literals (the classic loop: its scalar file has room).
HX_DEV cplx lit_cplx(double re, double im, uint32_t z) {
#if defined(TFHE_HIPEMU)
  (void)z;
  return cplx{re, im};
#else
  const uint64_t br = __builtin_bit_cast(uint64_t, re), bi = __builtin_bit_cast(uint64_t, im);
  const uint64_t r2 = ((uint64_t)((uint32_t)(br >> 32) | z) << 32) | ((uint32_t)br | z);
  const uint64_t i2 = ((uint64_t)((uint32_t)(bi >> 32) | z) << 32) | ((uint32_t)bi | z);
  return cplx{__builtin_bit_cast(double, r2), __builtin_bit_cast(double, i2)};
#endif
}
HX_DEV cplx times_i(const cplx c) { return cplx{-c.im, c.re}; }
HX_DEV cplx times_mi(const cplx c) { return cplx{c.im, -c.re}; }
HX_DEV cplx ldg_c(const double *t, int idx) { return cplx{t[2 * idx], t[2 * idx + 1]}; }

#if defined(TFHE_HIPEMU)
HX_DEV void flag_set(volatile uint32_t *f, uint32_t v) { *f = v; }
HX_DEV void flag_wait(volatile uint32_t *f, uint32_t v) {
  while (*f < v) hipemu::yield_barrier(0);
}
#else
HX_DEV void flag_set(uint32_t *f, uint32_t v) {
  __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
HX_DEV void flag_wait(uint32_t *f, uint32_t v) {
#if defined(WAVE_SPLIT_PROBE) && WAVE_SPLIT_PROBE == 4  // timing probe (races): no waiting on the partner
  return;
#endif
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(WAVE_FLAG_SLEEP);
}
#endif

// one radix-2 stage of the §4 butterfly over register-index bit BIT; TW(r) gives the twiddle
// of the butterfly whose first element is d[r]
template <int BIT, class TW>
HX_DEV void stage(cplx (&d)[16], TW tw) {
  HX_UNROLL
  for (int r = 0; r < 16; ++r)
    if (!(r & (1 << BIT))) bfly(d[r], d[r | (1 << BIT)], tw(r));
}
// the last stage of a pass: every finished point goes straight to LDS (st(r) stores d[r]), so the 16
// stores issue under the butterflies instead of as one burst behind them
template <int BIT, class TW, class ST>
HX_DEV void stage_store(cplx (&d)[16], TW tw, ST st) {
  HX_UNROLL
  for (int r = 0; r < 16; ++r)
    if (!(r & (1 << BIT))) {
      bfly(d[r], d[r | (1 << BIT)], tw(r));
      st(r);
      st(r | (1 << BIT));
    }
}
// loads of a pass, in the order the butterflies of its first stage (register bit BIT) consume them;
// LDS returns in order, so the first butterfly starts after two loads, not sixteen
template <int BIT, class LD>
HX_DEV void load_pairs(LD ld) {
  HX_UNROLL
  for (int r = 0; r < 16; ++r)
    if (!(r & (1 << BIT))) {
      ld(r);
      ld(r | (1 << BIT));
    }
}

// the §4 butterfly with the twiddle 1 and with -i: fma(b.re, 1, a.re) = a.re + b.re and the products by 0 vanish —
// the same roundings as bfly() with those twiddles (the sign of an exact zero aside, which no later step reads)
HX_DEV void bfly_one(cplx &a, cplx &b) {
  const double o1r = a.re + b.re, o1i = a.im + b.im;
  b.re = fma(2.0, a.re, -o1r);
  b.im = fma(2.0, a.im, -o1i);
  a.re = o1r;
  a.im = o1i;
}
HX_DEV void bfly_mi(cplx &a, cplx &b) {
  const double o1r = a.re + b.im, o1i = a.im - b.re;
  b.re = fma(2.0, a.re, -o1r);
  b.im = fma(2.0, a.im, -o1i);
  a.re = o1r;
  a.im = o1i;
}

// Everything a wave needs to know about its place in the workgroup
struct WaveCtx {
  cplx *buf;         // my exchange buffer
  const cplx *obuf;  // my partner's
  const cplx *T;     // compact twiddle table
  int lane, hi4, lo2, w;
};

// slot bases of the three register<->lane mappings (see BUF_SLOTS comment)
HX_DEV int base_m1(const WaveCtx &c) { return c.lane; }
HX_DEV int base_m2(const WaveCtx &c) { return c.hi4 * 68 + c.lo2; }
HX_DEV int base_m3(const WaveCtx &c) { return c.lane * 17; }
// mapping MX (after the permlane swaps): register bits (3,2,1,0) = position bits (5,4,7,6), lane bits
// (5,4) = position bits (9,8), lane bits 3..0 = position bits 3..0.  Its slot in the M3/P_B layout is
//   ((lane>>4)*16 + (r&3)*4 + (r>>2))*17 + (lane&15) = base_mx + mx_off(r)
HX_DEV int base_mx(const WaveCtx &c) { return (c.lane >> 4) * 272 + (c.lane & 15); }
constexpr int mx_off(int r) { return ((r & 3) * 4 + (r >> 2)) * 17; }

// register bits (3,2) <-> lane bits (5,4): M1 <-> MX without touching LDS (two permlane swaps per dword)
HX_DEV void swap_regs_lane54(cplx (&d)[16]) {
  HX_UNROLL
  for (int r = 0; r < 8; ++r) {  // register bit 3 <-> lane bit 5
    uint32_t a[4], b[4];
    __builtin_memcpy(a, &d[r], 16);
    __builtin_memcpy(b, &d[r + 8], 16);
    HX_UNROLL
    for (int k = 0; k < 4; ++k) hx_permlane32_swap(a[k], b[k]);
    __builtin_memcpy(&d[r], a, 16);
    __builtin_memcpy(&d[r + 8], b, 16);
  }
  HX_UNROLL
  for (int r = 0; r < 16; ++r)
    if (!(r & 4)) {  // register bit 2 <-> lane bit 4
      uint32_t a[4], b[4];
      __builtin_memcpy(a, &d[r], 16);
      __builtin_memcpy(b, &d[r + 4], 16);
      HX_UNROLL
      for (int k = 0; k < 4; ++k) hx_permlane16_swap(a[k], b[k]);
      __builtin_memcpy(&d[r], a, 16);
      __builtin_memcpy(&d[r + 4], b, 16);
    }
}

// Twiddles kept in registers for the whole launch by the classic one-level loop (template parameter RES of the
// transforms), read from the LDS table once in front of the CMUX loop: level 1 = the eight of pass F2 (they depend on
// lane >> 4 only), level 2 = also pass I2's two.  Every LDS read that leaves the loop is worth about 0.1 % of the launch
// (one box, batch 4096: table reads 33.8 ms, literals for the wave-uniform ones 32.9, level 1 32.7, level 2 32.5; a
// third level — the first two twiddles of passes F3 and I3 — spills and runs at 33.3).
struct ResidentTwiddles {
  cplx e4[4], e5[4];  // F2
  cplx w16, e32;      // I2
};
template <int LEVEL>
HX_DEV void load_resident_twiddles(ResidentTwiddles &t, const cplx *T, int lane) {
  const int g4 = lane >> 4, l15 = lane & 15;
  if constexpr (LEVEL >= 1) {
    HX_UNROLL
    for (int j = 0; j < 4; ++j) {
      t.e4[j] = T[T_F2 + g4 * 4 + j];
      t.e5[j] = T[T_F2 + 16 + g4 * 4 + j];
    }
  }
  if constexpr (LEVEL >= 2) {
    t.w16 = T[T_W16 + (l15 & 7)];
    if (l15 & 8) t.w16 = times_mi(t.w16);
    t.e32 = T[T_E32 + l15];
  }
}

// ---- forward transform of 16 points per lane: mapping M1 in, mapping M3 out (and stored in my buffer)
//   F1 stages 0..3 (position bits 9..6, registers) -> permlane swaps -> F2 stages 4,5 (bits 5,4, registers)
//   -> LDS transposition MX -> M3 -> F3 stages 6..9 (bits 3..0, registers)
template <int RES = 0, int LIT = 0>
HX_DEV void wave_forward(cplx (&d)[16], WaveCtx c, const ResidentTwiddles *res = nullptr) {
  HX_OPAQUE(c.lane);
  uint32_t zlit = 0;
  if constexpr (LIT == 2) HX_OPAQUE_S(zlit);
  const int lane = c.lane, g4 = c.lane >> 4;
  const cplx *T = c.T;
  {  // pass F1: the same twiddles in every lane of every launch
    auto tw = [&](int x) { return LIT == 2 ? lit_cplx(LIT_F1[x][0], LIT_F1[x][1], zlit) : LIT ? cplx{LIT_F1[x][0], LIT_F1[x][1]} : T[T_F1 + x]; };
    const cplx w0 = tw(0);
    stage<3>(d, [&](int) { return w0; });
    const cplx e1 = tw(1);
    stage<2>(d, [&](int r) { return (r >> 3) ? times_i(e1) : e1; });
    HX_SCHED_FENCE();
    const cplx e2[2] = {tw(2), tw(3)};
    stage<1>(d, [&](int r) { return ((r >> 2) & 1) ? times_i(e2[r >> 3]) : e2[r >> 3]; });
    HX_SCHED_FENCE();
    const cplx e3[4] = {tw(4), tw(5), tw(6), tw(7)};
    stage<0>(d, [&](int r) { return ((r >> 1) & 1) ? times_i(e3[r >> 2]) : e3[r >> 2]; });
  }
  HX_SCHED_FENCE();
  swap_regs_lane54(d);
  HX_SCHED_FENCE();
  {  // stage 4: group (pos >> 6) = (lane>>4)*4 + (r&3); stage 5: group (pos >> 5) = that*2 + (r>>3)
    cplx e4[4], e5[4];
    HX_UNROLL
    for (int j = 0; j < 4; ++j) e4[j] = RES >= 1 ? res->e4[j] : T[T_F2 + g4 * 4 + j];
    stage<3>(d, [&](int r) { return e4[r & 3]; });
    HX_SCHED_FENCE();
    HX_UNROLL
    for (int j = 0; j < 4; ++j) e5[j] = RES >= 1 ? res->e5[j] : T[T_F2 + 16 + g4 * 4 + j];
    cplx *px = c.buf + base_mx(c);  // transposition MX -> M3, store side
    stage_store<2>(d, [&](int r) { return (r >> 3) ? times_i(e5[r & 3]) : e5[r & 3]; },
                   [&](int r) { px[mx_off(r)] = d[r]; });
  }
  HX_WAVE_SYNC();
  HX_PRIO_OPT(WAVE_PRIO_F3);
  {  // stages 6..9 over position bits 3..0 (= r bits 3..0), group index = lane . (r bits)
    const cplx w6 = T[T_F6 + lane];
    cplx *p3 = c.buf + base_m3(c);  // transposition MX -> M3, load side
    load_pairs<3>([&](int r) { d[r] = p3[r]; });
    HX_WAVE_SYNC();
    stage<3>(d, [&](int) { return w6; });
    const cplx e7 = T[T_F2 + 64 + lane];
    stage<2>(d, [&](int r) { return (r >> 3) ? times_i(e7) : e7; });
    HX_SCHED_FENCE();
    const cplx e8[2] = {T[T_F3 + lane], T[T_F3 + 64 + lane]};
    stage<1>(d, [&](int r) { return ((r >> 2) & 1) ? times_i(e8[r >> 3]) : e8[r >> 3]; });
    HX_SCHED_FENCE();
    const cplx e9[4] = {T[T_F3 + 128 + lane], T[T_F3 + 192 + lane], T[T_F3 + 256 + lane],
                        T[T_F3 + 320 + lane]};
    stage_store<0>(d, [&](int r) { return ((r >> 1) & 1) ? times_i(e9[r >> 2]) : e9[r >> 2]; },
                   [&](int r) { p3[r] = d[r]; });
  }
  HX_WAVE_SYNC();
}

// ---- inverse transform (mapping M3 in, M1 out), untwist and accumulation into the torus regs.
// Twiddle of DIT stage `half`, butterfly offset j: inv[half + j] = E[j*512/half] (nested tables),
// E[J] = T_INV[J] for J < 256 and -i*T_INV[J-256] above.
// pass I1 of the inverse on registers o[g*4 .. g*4+3]: stages half = 1, 2 (plain), mapping M3
HX_DEV void inverse_pass1_group(cplx (&o)[16], int g) {
  HX_UNROLL
  for (int r = g * 4; r < g * 4 + 4; r += 2) {
    const cplx x = o[r], y = o[r + 1];
    o[r] = cplx{x.re + y.re, x.im + y.im};
    o[r + 1] = cplx{x.re - y.re, x.im - y.im};
  }
  HX_UNROLL
  for (int r = g * 4; r < g * 4 + 2; ++r) {
    const cplx x = o[r], y = o[r | 2];
    if (r & 1) {  // j = 1: w = -i
      o[r] = cplx{x.re + y.im, x.im - y.re};
      o[r | 2] = cplx{x.re - y.im, x.im + y.re};
    } else {
      o[r] = cplx{x.re + y.re, x.im + y.im};
      o[r | 2] = cplx{x.re - y.re, x.im - y.im};
    }
  }
}

// inverse stage half = 4 (position bit 2 = r bit 2, twiddle E[128 (r & 1)], times -i for r & 2) on the points
// r0 .. r0 + 7 (r0 = 0 or 8): literal or table twiddles, same butterflies either way
template <int LIT>
HX_DEV void inverse_stage_half4(cplx (&o)[16], int r0, const cplx *T, uint32_t zlit = 0) {
  if constexpr (LIT != 0) {
    const cplx a1 = LIT == 2 ? lit_cplx(LIT_E64[2][0], LIT_E64[2][1], zlit) : cplx{LIT_E64[2][0], LIT_E64[2][1]};
    HX_UNROLL
    for (int r = r0; r < r0 + 4; ++r) {
      if ((r & 3) == 0) bfly_one(o[r], o[r | 4]);
      else if ((r & 3) == 2) bfly_mi(o[r], o[r | 4]);
      else bfly(o[r], o[r | 4], (r & 2) ? times_mi(a1) : a1);
    }
  } else {
    const cplx a0 = T[T_INV], a1 = T[T_INV + 128];
    HX_UNROLL
    for (int r = r0; r < r0 + 4; ++r) {
      const cplx e = (r & 1) ? a1 : a0;
      bfly(o[r], o[r | 4], (r & 2) ? times_mi(e) : e);
    }
  }
}

// OVERWRITE (multi-bit: dst = 0 + src (x) GGSW): the result replaces the accumulator and is not staged.
// NEG: the registers hold MINUS the accumulator (see make_digits); from_torus is odd, so the negated
// term is the conversion of the negated real, whose sign rides on the untwist multiplication for free.
// RAW (exact engine, split-key form): no torus conversion — o[r] becomes (t_re, t_im), the untwisted real values of
// coefficients r*64 + lane and 1024 + r*64 + lane; nothing is staged, the accumulator registers are not touched.
// PASS1_DONE: 0 nothing done, 1 stages half = 1, 2 done by the caller
// before_store(): called in front of the first store to the exchange buffer (the caller's deferred wait for the partner's
// "done with your buffer").  after_load(): called when the transposition has left the buffer, which nothing below touches in
// RAW mode (the split-key engine puts its digit transform back there at once, so that the partner never waits for it).
struct WaveNoHook {
  HX_DEV void operator()() const {}
};
template <int PASS1_DONE, bool OVERWRITE = false, bool NEG = false, bool RAW = false, int RES = 0, int LIT = 0,
          class BeforeStore = WaveNoHook, class AfterLoad = WaveNoHook>
HX_DEV void wave_inverse_accumulate(cplx (&o)[16], uint64_t (&acc_re)[16], uint64_t (&acc_im)[16], WaveCtx c,
                                    const ResidentTwiddles *res = nullptr, BeforeStore before_store = BeforeStore{},
                                    AfterLoad after_load = AfterLoad{}) {
  uint64_t *stg = (uint64_t *)c.buf;
  HX_OPAQUE(c.lane);
  const cplx *T = c.T;
  if constexpr (PASS1_DONE == 0) {
    HX_UNROLL
    for (int g = 0; g < 4; ++g) inverse_pass1_group(o, g);
  }
  // pass I1 (continued): stages half = 4, 8 over position bits 2, 3 (= r bits 2, 3); j = r & 3, r & 7, so the
  // twiddles are the same in every lane: E[j*128] and E[j*64]
  {
    uint32_t zlit = 0;
    if constexpr (LIT == 2) HX_OPAQUE_S(zlit);
    inverse_stage_half4<LIT>(o, 0, T, zlit);
    inverse_stage_half4<LIT>(o, 8, T, zlit);
    HX_SCHED_FENCE();
    before_store();
    cplx *p3 = c.buf + base_m3(c);  // transposition M3 -> MX, store side
    if constexpr (LIT != 0) {
      HX_UNROLL
      for (int r = 0; r < 8; ++r) {  // stage half = 8: twiddle E[64 (r & 3)], times -i for r & 4
        const cplx e = LIT == 2 ? lit_cplx(LIT_E64[r & 3][0], LIT_E64[r & 3][1], zlit) : cplx{LIT_E64[r & 3][0], LIT_E64[r & 3][1]};
        if (r == 0) bfly_one(o[r], o[r | 8]);
        else if (r == 4) bfly_mi(o[r], o[r | 8]);
        else bfly(o[r], o[r | 8], (r & 4) ? times_mi(e) : e);
        p3[r] = o[r];
        p3[r | 8] = o[r | 8];
      }
    } else {
      const cplx a0 = T[T_INV], a1 = T[T_INV + 128];
      const cplx b4[4] = {a0, T[T_INV + 64], a1, T[T_INV + 192]};
      stage_store<3>(o, [&](int r) { return (r & 4) ? times_mi(b4[r & 3]) : b4[r & 3]; },
                     [&](int r) { p3[r] = o[r]; });
    }
  }
  HX_WAVE_SYNC();
  // pass I2: stages half = 16, 32 over position bits 4, 5 (= r bits 2, 3 in mapping MX); j = (r bit 2).(lane & 15)
  {
    int ln = c.lane;
    HX_OPAQUE(ln);
    const int l15 = ln & 15;
    cplx w16, e32;
    if constexpr (RES >= 2) {
      w16 = res->w16;
      e32 = res->e32;
    } else {
      w16 = T[T_W16 + (l15 & 7)];
      e32 = T[T_E32 + l15];
    }
    const cplx *px = c.buf + base_mx(c);  // transposition M3 -> MX, load side
    load_pairs<2>([&](int r) { o[r] = px[mx_off(r)]; });
    HX_WAVE_SYNC();
    after_load();
    if constexpr (RES < 2) {
      if (l15 & 8) w16 = times_mi(w16);
    }
    stage<2>(o, [&](int) { return w16; });
    stage<3>(o, [&](int r) { return (r & 4) ? times_mi(e32) : e32; });
  }
  HX_SCHED_FENCE();
  swap_regs_lane54(o);  // MX -> M1
  HX_SCHED_FENCE();
  HX_PRIO_OPT(WAVE_PRIO_I3);
  // pass I3: stages half = 64..512 over position bits 6..9 (= r bits 0..3); j = (r bits).lane
  {
    int lane = c.lane;
    HX_OPAQUE(lane);
    cplx w7 = T[T_W7 + (lane & 31)];
    if (lane & 32) w7 = times_mi(w7);
    stage<0>(o, [&](int) { return w7; });
    const cplx e8 = T[T_E8 + lane];
    stage<1>(o, [&](int r) { return (r & 1) ? times_mi(e8) : e8; });
    HX_SCHED_FENCE();
    const cplx e9[2] = {T[T_INV + lane * 2], T[T_INV + (64 + lane) * 2]};
    stage<2>(o, [&](int r) { return (r & 2) ? times_mi(e9[r & 1]) : e9[r & 1]; });
    HX_SCHED_FENCE();
    const cplx e10[4] = {T[T_INV + lane], T[T_INV + 64 + lane], T[T_INV + 128 + lane], T[T_INV + 192 + lane]};
    stage<3>(o, [&](int r) { return (r & 4) ? times_mi(e10[r & 3]) : e10[r & 3]; });
  }
  HX_SCHED_FENCE();
  HX_PRIO_OPT(WAVE_PRIO_CONV);
  // untwist, back to the torus, accumulate (fft/mod.rs:311-330)
  int lane_u = c.lane;
  HX_OPAQUE(lane_u);
  const cplx *Tu_lo = T + T_U + lane_u;         // u[r*64 + lane]           (r < 8)
  const cplx *Tu_hi = T + T_U + 1024 - lane_u;  // u[1024 - (r*64 + lane)]  (r >= 8), mirrored
  const TorusConsts kt = torus_consts();
  HX_UNROLL
  for (int r = 0; r < 16; ++r) {
    cplx u;
    if (r < 8) {
      u = Tu_lo[r * 64];
    } else {
      const cplx e = Tu_hi[-r * 64];
      u = (r == 8 && lane_u == 0) ? e : cplx{-e.im, -e.re};  // j = 512 is stored directly
    }
    const double tr = NEG ? fma(o[r].im, u.im, -o[r].re * u.re) : fma(-o[r].im, u.im, o[r].re * u.re);
    const double ti = NEG ? fma(-o[r].im, u.re, -o[r].re * u.im) : fma(o[r].im, u.re, o[r].re * u.im);
    if constexpr (RAW) {
      o[r] = cplx{tr, ti};
    } else if constexpr (OVERWRITE) {
      acc_re[r] = from_torus(tr);
      acc_im[r] = from_torus(ti);
    } else {
      from_torus_add(acc_re[r], tr, kt);
      from_torus_add(acc_im[r], ti, kt);
      // stage the updated coefficients (c = r*64 + lane, 1024 + c) for the next iteration's rotation;
      // the buffer is free (the transposition reads above are complete) and these stores issue under the
      // conversion arithmetic instead of in front of the next rotation
      stg[lane_u + r * 64] = acc_re[r];
      stg[lane_u + 1024 + r * 64] = acc_im[r];
#if WAVE_ROT_PAIRS
      if (r == 0) stg[lane_u + 2048] = acc_re[0];  // the ring's first 64 words again behind its end (make_digits)
#endif
    }
    if ((r & 3) == 3) HX_SCHED_FENCE();
  }
  HX_WAVE_SYNC();
}

// GROUPING > 0: multi-bit PBS on the same machinery (cc/algorithms/lwe_multi_bit_programmable_bootstrapping.rs
// :116-156, :647-880): per group of g mask elements the external product  acc <- acc (x) GGSW_comb  with the
// keybundle combined in the Fourier domain on the fly (see the MULTIBIT block below); no rotation, the result
// overwrites the accumulator.
//
// SHARE (multi-bit only, an even number of LWEs per workgroup): the key of a group is what a CU's vector-L1
// delivers slowest (1 MB per LWE and group through 64 B/clk), so the two LWEs of a QUAD of waves share every key
// load: after the forward transforms the quad's four waves re-partition the multiply-accumulate by (output
// column, half of the 16 points a lane owns) instead of by (LWE, output column) — a wave combines the keybundle
// of its column at its 8 points for BOTH LWEs out of one load of each key element (their monomial factors
// differ, the key does not), reads the four digit transforms from the quad's buffers, and hands the half it
// computed for the other LWE back through LDS before the inverse transforms.  Same products in the same order
// per output point: identical bits.  Workgroup barriers replace the pair flags.
//
// LIMBS > 0: the EXACT engine (tfhe-ntt semantics, cc/algorithms/lwe_programmable_bootstrapping/ntt64_bnf_pbs.rs
// :208-280, commons/math/ntt/ntt64.rs:144-245) on this kernel's f64 machinery.  The reference's result per CMUX is
// the negacyclic product  R = sum_rows digit_poly (*) key_poly  modulo the Goldilocks prime P (key words switched to
// P, digits as signed integers), switched back to 2^64 — a property of the integers involved, not of the
// transform.  Here the centred key word kc in (-P/2, P/2] is split into LIMBS balanced 16-bit limbs,
// kc = sum_m c_m 2^(16 m), each limb polynomial is kept in the Fourier domain (integer inputs, no torus scaling),
// and per CMUX the digit transform F is multiplied with every limb in turn: S_m = sum_rows d (*) c_m is an integer of
// magnitude below (k+1) l N (B/2) 2^15 = 2^49 that the f64 transform reproduces to within about 2^-9 (RMS; the
// distance from the nearest integer is checked on every coefficient and the launch raises the scratch's round-off flag above 1/4 — the round-off
// check of every FFT-based exact multiplication), so rint() of the inverse transform IS S_m, and
//     R = sum_m S_m 2^(16 m)  mod P          (Horner, most significant limb first, 64-bit Goldilocks arithmetic)
// is the reference's value, bit for bit.  Order of the blind rotation as in the NTT path: acc starts as the LUT,
// ct1 = acc X^a_hat - acc per CMUX, the rotation by -b_hat comes last.  The accumulator lives in device memory
// between its two touches per CMUX (PbsArgs::acc_scratch, L2 / Infinity Cache resident): the registers carry the
// Horner states (64), the digit transform (64: re-published to the pair's LDS buffer after every limb's inverse
// transposition has used that buffer) and the product being transformed back (64).
template <int LEVEL_CT, int BASE_LOG_CT, int GROUPING = 0, bool SHARE = false, int LIMBS = 0, bool OCTET = false>
__global__ void __launch_bounds__(TPB) pbs_fft_wave_kernel(PbsArgs a, FftTables tb) {
  constexpr bool MULTIBIT = GROUPING > 0;
  static_assert(!SHARE || MULTIBIT, "SHARE is a mode of the multi-bit loop");
  static_assert(!OCTET || (MULTIBIT && !SHARE && LEVEL_CT >= 1), "OCTET is a mode of the multi-bit loop with the level count fixed at compile time");
  static_assert(LIMBS == 0 || (!MULTIBIT && LEVEL_CT == 1 && BASE_LOG_CT != 0 && BASE_LOG_CT <= 23),
                "split-key exact engine: one level, base_log <= 23 (the products must stay below 2^49)");
  HX_DYN_SMEM(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = HX_UNIFORM(tid >> 6);
  const int w = wave & 1;         // polynomial of the GLWE this wave owns (0 = mask, 1 = body)
  const int pair = wave >> 1;     // which of the block's 4 LWEs
  cplx *buf = (cplx *)(smem + (size_t)wave * BUF_BYTES);
  const cplx *obuf = (const cplx *)(smem + (size_t)(wave ^ 1) * BUF_BYTES);
  uint64_t *buf64 = (uint64_t *)buf;
  const cplx *T = (const cplx *)(smem + (size_t)WAVES * BUF_BYTES);
#if defined(TFHE_HIPEMU)
  volatile uint32_t *flags = (volatile uint32_t *)(smem + (size_t)WAVES * BUF_BYTES + (size_t)T_TOTAL * 16);
#else
  uint32_t *flags = (uint32_t *)(smem + (size_t)WAVES * BUF_BYTES + (size_t)T_TOTAL * 16);
#endif
  auto *f_ready_me = flags + (pair * 2 + w) * 2, *f_ready_ot = flags + (pair * 2 + (w ^ 1)) * 2;
  auto *r_done_me = f_ready_me + 1, *r_done_ot = f_ready_ot + 1;

  const uint32_t level = LEVEL_CT ? (uint32_t)LEVEL_CT : a.level;
  const uint32_t base_log = BASE_LOG_CT ? (uint32_t)BASE_LOG_CT : a.base_log;

  // ---- one-time: compact twiddle table into LDS, flags to zero
  {
    cplx *Tw = (cplx *)(smem + (size_t)WAVES * BUF_BYTES);
    for (int e = tid; e < T_TOTAL; e += (int)blockDim.x) {
      cplx v{0.0, 0.0};
      if (e < T_F2) {             // forward d = 0..3, even groups
        const int x = e - T_F1;
        v = ldg_c(tb.fwd, x == 0 ? 1 : x == 1 ? 2 : x < 4 ? 4 + 2 * (x - 2) : 8 + 2 * (x - 4));
      } else if (e < T_F3) {      // forward d = 4..7
        const int x = e - T_F2;
        if (x < 16) v = ldg_c(tb.fwd, 16 + x);
        else if (x < 32) v = ldg_c(tb.fwd, 32 + 2 * (x - 16));
        else if (x < 64) v = ldg_c(tb.fwd, 64 + 2 * (x - 32));
        else v = ldg_c(tb.fwd, 128 + 2 * (x - 64));
      } else if (e < T_INV) {     // forward d = 8, 9, even groups only
        const int x = e - T_F3;
        // stored [j][lane] so that a wave's read of one j is contiguous (the natural [lane][j] order is a
        // 2-way / 4-way bank conflict)
        if (x < 128) v = ldg_c(tb.fwd, 256 + 2 * ((x & 63) * 2 + (x >> 6)));
        else v = ldg_c(tb.fwd, 512 + 2 * (((x - 128) & 63) * 4 + ((x - 128) >> 6)));
      } else if (e < T_U) {       // E[J] = inv[512 + J], J < 256
        v = ldg_c(tb.inv, 512 + (e - T_INV));
      } else if (e < T_F6) {
        v = ldg_c(tb.untw, e - T_U);
      } else if (e < T_E8) {
        v = ldg_c(tb.fwd, 64 + (e - T_F6));
      } else if (e < T_W7) {
        v = ldg_c(tb.inv, 512 + 4 * (e - T_E8));
      } else if (e < T_E32) {
        v = ldg_c(tb.inv, 512 + 8 * (e - T_W7));
      } else if (e < T_W16) {
        v = ldg_c(tb.inv, 512 + 16 * (e - T_E32));
      } else if (e < T_W16X) {
        v = ldg_c(tb.inv, 512 + 32 * (e - T_W16));
      } else {
        v = ldg_c(tb.mono, (N / 8) * (e - T_W16X));
      }
      Tw[e] = v;
    }
    if (tid < FLAGS_BYTES / 4) flags[tid] = 0;
  }
  __syncthreads();
#if WAVE_PROBE_TS
  uint64_t *ts_rec = g_wave_ts ? g_wave_ts + (size_t)blockIdx.x * 8 : nullptr;
  if (ts_rec && tid == 0) {
    ts_rec[0] = __builtin_amdgcn_s_memrealtime();
    ts_rec[1] = __builtin_amdgcn_s_memtime();
    ts_rec[2] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
  }
#endif

  // the launch picks 1..4 LWEs per workgroup (blockDim.x = 128 per LWE): small batches spread over the CUs
  uint32_t sample = blockIdx.x * (blockDim.x >> 7) + pair;
  const bool valid = sample < a.num_samples;
  if constexpr (SHARE || OCTET) {
    // every wave of the workgroup works for its quad and meets the block barriers: the pairs past the end of a
    // ragged last workgroup redo the last ciphertext and write nothing
    if (!valid) sample = a.num_samples - 1;
  } else {
    // the whole pair leaves together.  Bare s_barriers further down (split-key loop, WAVE_CLASSIC_SYNC) stay correct: a
    // terminated wave no longer counts towards a workgroup barrier on gfx9, every surviving wave runs the same trip count in
    // front of its a_hat == 0 skip, and those loops are instantiated with SHARE = OCTET = false only (see the static_asserts)
    if (!valid) return;
  }
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * 2 * N + (size_t)w * N;
  const cplx *bsk = (const cplx *)a.bsk;
  const WaveCtx ctx0{buf, obuf, T, lane, lane >> 2, lane & 3, w};
  const WaveCtx &ctx = ctx0;

  // ---- body modulus switch (with the centered-mean correction), redundantly per wave
  uint64_t corr = 0;
  if (a.ms_type == 1) {
    uint64_t sh = 0;
    int64_t sd = 0;
    for (uint32_t i = lane; i < a.n; i += 64) {
      uint64_t h;
      int64_t dd;
      centered_ms_terms(lwe[i], LOG2N2, h, dd);
      sh += h;
      sd += dd;
    }
    buf64[lane] = sh;
    buf64[64 + lane] = (uint64_t)sd;
    HX_WAVE_SYNC();
    uint64_t th = 0, td = 0;
    for (int l = 0; l < 64; ++l) {
      th += buf64[l];
      td += buf64[64 + l];
    }
    HX_WAVE_SYNC();
    corr = centered_ms_finish(th, (int64_t)td, LOG2N2);
  }
  const uint32_t b_hat = (uint32_t)modulus_switch(lwe[a.n] + corr, LOG2N2);

  // ---- accumulator registers: coefficient (r*64 + lane) and (1024 + r*64 + lane).
  // The classic loop keeps MINUS the accumulator (NEGACC): the rotate-and-subtract of every iteration
  // then needs no 64-bit subtraction or negation (make_digits), additions being one instruction
  constexpr bool NEGACC = !MULTIBIT;
  uint64_t acc_re[16], acc_im[16];
  HX_UNROLL
  for (int r = 0; r < 16; ++r) {
    bool neg;
    if constexpr (LIMBS > 0) {  // the rotation by -b_hat comes last on this path (ntt64_bnf_pbs.rs:262-271)
      acc_re[r] = (uint64_t)0 - lut[r * 64 + lane];
      acc_im[r] = (uint64_t)0 - lut[1024 + r * 64 + lane];
      continue;
    }
    uint32_t src = monomial_div_src(r * 64 + lane, b_hat, N, neg);
    uint64_t v = lut[src];
    acc_re[r] = (neg != NEGACC) ? (uint64_t)0 - v : v;
    src = monomial_div_src(1024 + r * 64 + lane, b_hat, N, neg);
    v = lut[src];
    acc_im[r] = (neg != NEGACC) ? (uint64_t)0 - v : v;
  }

  // ct1 = acc * X^a_hat - acc for my polynomial, decomposed at level index idx, as f64 points
  // (polynomial_algorithms.rs:662-727: coefficient c takes +/- acc[(c - r) mod N], negated when
  //  c < r, all signs flipped when a_hat >= N)
  auto stage_acc = [&]() {
    int lane = ctx.lane;
    HX_OPAQUE(lane);
    uint64_t *p = buf64 + lane;
    HX_UNROLL
    for (int r = 0; r < 16; ++r) {
      p[r * 64] = acc_re[r];
      p[1024 + r * 64] = acc_im[r];
    }
#if WAVE_ROT_PAIRS
    p[2048] = acc_re[0];  // the ring's first 64 words again behind its end (make_digits)
#endif
    HX_WAVE_SYNC();
  };
  // EXACT selects the decomposer's own bit sequence; the default for one level is the two-instruction
  // rounding decomp_digit_l1_fast, which differs from it only where it returns -B/2: a lane that saw that
  // value (about one coefficient in 2^23) redoes its points with EXACT = true
  auto make_digits_impl = [&](cplx (&d)[16], uint32_t a_hat, uint32_t idx, auto exact_tag) -> int32_t {
    constexpr bool EXACT = decltype(exact_tag)::value;
    int lane = ctx.lane;
    HX_OPAQUE(lane);
    // ct1[c] = sign * acc[(c - a_hat) mod N] - acc[c] with sign = -1 for c < rr (rr = a_hat mod N), all
    // signs flipped when a_hat >= N.  The registers and the staged copy hold A = -acc, so with the staged
    // word S = A[(c - rr) mod N] the value is  A[c] - S  (sign +)  or  A[c] + S  (sign -), and both are
    //     ((A[c] ^ M) + S) ^ M ,   M = all-ones (sign +) or zero (sign -)
    // since ~(~A + S) = A - S: one 64-bit addition and xors, no subtraction, negation, compare or select.
    // u = 8 (c - rr) gives the byte offset (mod 16 KiB) and, by its sign bit, the negacyclic wrap.
    // Bit 31 of u (unused by the offset) also carries the a_hat < N flag, so the mask is one shift of u.
    const int32_t ub =
        (int32_t)((uint32_t)(((int32_t)lane - (int32_t)(a_hat & (N - 1))) * 8) + ((a_hat & N) ? 0u : 0x80000000u));
    int32_t lowest = 0;
    uint32_t vzero = 0;
#ifndef WAVE_STAGED_SGPR_BASE
    HX_LAUNDER(vzero);  // the staged copy's base in a vector register: a scalar operand doubles the cost of the add
#endif
    const char *staged = (const char *)buf64 + vzero;
    uint64_t sp0[2] = {0, 0}, sp1[2] = {0, 0};
    (void)sp0;
    (void)sp1;
    HX_UNROLL
    for (int r = 0; r < 16; ++r) {
      uint64_t x0, x1;
      if constexpr (MULTIBIT) {  // external product of the accumulator itself
        x0 = acc_re[r];
        x1 = acc_im[r];
      } else {
        const int32_t u0 = (int32_t)((uint32_t)ub + r * 512u), u1 = (int32_t)((uint32_t)u0 + 8192u);
        const uint32_t m0 = (uint32_t)(u0 >> 31), m1 = (uint32_t)(u1 >> 31);  // all-ones: sign +
        const uint64_t M0 = ((uint64_t)m0 << 32) | m0, M1 = ((uint64_t)m1 << 32) | m1;
#if WAVE_ROT_PAIRS
        if ((r & 1) == 0) {  // rows r and r + 1 of both halves: the second word sits 512 bytes behind the first
          const uint64_t *q0 = (const uint64_t *)(staged + (u0 & 0x3ff8)), *q1 = (const uint64_t *)(staged + (u1 & 0x3ff8));
          sp0[0] = q0[0];
          sp0[1] = q0[64];
          sp1[0] = q1[0];
          sp1[1] = q1[64];
        }
        const uint64_t s0 = sp0[r & 1], s1 = sp1[r & 1];
#else
        const uint64_t s0 = *(const uint64_t *)(staged + (u0 & 0x3ff8));
        const uint64_t s1 = *(const uint64_t *)(staged + (u1 & 0x3ff8));
#endif
        uint64_t a0 = acc_re[r], a1 = acc_im[r];
        if constexpr (LIMBS > 0) {  // the accumulator is not in registers here: my own coefficients from the staged copy
          a0 = *(const uint64_t *)(staged + (lane + r * 64) * 8);
          a1 = *(const uint64_t *)(staged + (1024 + lane + r * 64) * 8);
        }
        x0 = ((a0 ^ M0) + s0) ^ M0;
        x1 = ((a1 ^ M1) + s1) ^ M1;
      }
      if constexpr (LEVEL_CT == 1 && BASE_LOG_CT != 0 && BASE_LOG_CT <= 30) {
        // one level: the digit is the decomposer's initial state and depends on the high dword only
        if constexpr (EXACT) {
          d[r] = cplx{(double)decomp_digit_l1_hi((uint32_t)(x0 >> 32), BASE_LOG_CT),
                      (double)decomp_digit_l1_hi((uint32_t)(x1 >> 32), BASE_LOG_CT)};
        } else {
          const int32_t d0 = decomp_digit_l1_fast((uint32_t)(x0 >> 32), BASE_LOG_CT);
          const int32_t d1 = decomp_digit_l1_fast((uint32_t)(x1 >> 32), BASE_LOG_CT);
          lowest = d0 < lowest ? d0 : lowest;
          lowest = d1 < lowest ? d1 : lowest;
          d[r] = cplx{(double)d0, (double)d1};
        }
      } else {
        const int64_t d0 = decomp_digit(x0, base_log, level, idx);
        const int64_t d1 = decomp_digit(x1, base_log, level, idx);
        if constexpr (BASE_LOG_CT != 0 && BASE_LOG_CT <= 31)
          d[r] = cplx{(double)(int32_t)d0, (double)(int32_t)d1};  // |digit| <= 2^(base_log-1): exact
        else
          d[r] = cplx{i64_to_f64(d0), i64_to_f64(d1)};
      }
      if ((r & 3) == 3) HX_SCHED_FENCE();
    }
    return lowest;
  };
  auto make_digits = [&](cplx (&d)[16], uint32_t a_hat, uint32_t idx) {
    // the accumulator is already staged in buf64 (stage_acc at start, then by every
    // wave_inverse_accumulate); later levels of one iteration re-stage, the transposes reused the buffer
    if (!MULTIBIT && idx != 0) stage_acc();
    if constexpr (LEVEL_CT == 1 && BASE_LOG_CT != 0 && BASE_LOG_CT <= 30) {
      const int32_t lowest = make_digits_impl(d, a_hat, idx, std::false_type{});
      if (lowest == -(1 << (BASE_LOG_CT - 1))) make_digits_impl(d, a_hat, idx, std::true_type{});
    } else {
      make_digits_impl(d, a_hat, idx, std::true_type{});
    }
    HX_WAVE_SYNC();
  };

  // key rows of GGSW_i: [i][idx][row][c = w][storage s = r*64 + lane], consumed in 4 chunks of 4 points
  const uint32_t key_levels = LIMBS > 0 ? (uint32_t)LIMBS : level;  // split-key form: [i][limb][row][col][slot]
#ifndef WAVE_KEY_BUFFER
// The key rows requested through ONE buffer descriptor over the whole key — lane offsets in one vector register, the row's
// byte offset as a scalar — instead of through two per-lane 64-bit pointers (plus one more pointer pair per 4 KB window of
// immediate offsets the sweep of 16 KB crosses).  0: never, 1: always, 2: in the split-key exact engine only (same box, ms per
// 4096: split-key engine 137.9 -> 135.0; classic loop 33.4 -> 33.5: its register file is full either way and the descriptor
// costs scalar registers — profiles/r05_ab_split_buffers.txt)
#define WAVE_KEY_BUFFER 2
#endif
#ifndef WAVE_KEY_AUX
#define WAVE_KEY_AUX 0  // cache-policy bits of the buffer-addressed key requests
#endif
  constexpr bool KEYBUF = WAVE_KEY_BUFFER == 1 || (WAVE_KEY_BUFFER == 2 && LIMBS > 0);
  const HxBuffer bskb = hx_make_buffer(a.bsk, KEYBUF ? (uint32_t)((size_t)a.n * key_levels * 4 * n * sizeof(cplx)) : 0u);
  auto key_rows = [&](uint32_t i, uint32_t idx, const cplx *&b0, const cplx *&b1) {
    if constexpr (KEYBUF) {
      // the "pointers" carry the rows' byte offsets (wave-uniform): < 2^32 for every supported key
      b0 = (const cplx *)(uintptr_t)(((((size_t)i * key_levels + idx) * 2 + 0) * 2 + w) * n * sizeof(cplx));
      b1 = (const cplx *)(uintptr_t)(((((size_t)i * key_levels + idx) * 2 + 1) * 2 + w) * n * sizeof(cplx));
    } else {
      int lane = ctx.lane;
      HX_OPAQUE(lane);
      b0 = bsk + ((((size_t)i * key_levels + idx) * 2 + 0) * 2 + w) * n + lane;
      b1 = bsk + ((((size_t)i * key_levels + idx) * 2 + 1) * 2 + w) * n + lane;
    }
  };
  auto key_request = [&](cplx (&k0)[4], cplx (&k1)[4], const cplx *b0, const cplx *b1, int ch) {
    if constexpr (KEYBUF) {
      int lane = ctx.lane;
      HX_OPAQUE(lane);
      const uint32_t o0 = (uint32_t)(uintptr_t)b0, o1 = (uint32_t)(uintptr_t)b1;
      HX_UNROLL
      for (int j = 0; j < 4; ++j) {
        const hx_f64x2 v0 = hx_buffer_load_f64x2<WAVE_KEY_AUX>(bskb, (uint32_t)lane * 16u, o0 + (uint32_t)(ch * 4 + j) * 1024u);
        const hx_f64x2 v1 = hx_buffer_load_f64x2<WAVE_KEY_AUX>(bskb, (uint32_t)lane * 16u, o1 + (uint32_t)(ch * 4 + j) * 1024u);
        k0[j] = cplx{v0.x, v0.y};
        k1[j] = cplx{v1.x, v1.y};
      }
    } else {
      HX_UNROLL
      for (int j = 0; j < 4; ++j) {
        k0[j] = load_global_cplx<false>(&b0[(ch * 4 + j) * 64]);
        k1[j] = load_global_cplx<false>(&b1[(ch * 4 + j) * 64]);
      }
    }
  };

  // publish my transform, fetch the partner's, multiply-accumulate with GGSW_i rows into dst
  // (cc/fft_impl/fft64/crypto/ggsw.rs:616-697 order: row 0 then row 1 within a level).
  // Chunks 0 and 1 of the key were requested by the caller at the top of the iteration (ka*, kb*);
  // chunk c + 2 is requested into the registers chunk c has just released.  With FUSE_PASS1 the
  // first inverse pass (which only mixes the 4 points of one chunk) runs right behind each chunk.
  // publish: my transform's "ready" flag is set here (false: the caller set it earlier).  wait_done (a type): the wait for
  // the partner's "done with your buffer" closes the multiply-accumulate; std::false_type: the caller waits itself, right in
  // front of its next store to the buffer (wave_inverse_accumulate's before_store), with the butterflies in between as slack
  auto mac_impl = [&](cplx (&dst)[16], cplx (&d)[16], cplx (&ka0)[4], cplx (&ka1)[4], cplx (&kb0)[4], cplx (&kb1)[4],
                      const cplx *b0, const cplx *b1, uint32_t idx, uint32_t epoch, auto fuse_pass1, bool publish,
                      auto wait_done) {
    WaveCtx ctx = ctx0;
    HX_OPAQUE(ctx.lane);
    const int lane = ctx.lane;
    // wave_forward left my transform in my buffer (mapping M3)
    if (publish && lane == 0) flag_set(f_ready_me, epoch);
    constexpr int early = (LEVEL_CT == 1 && !MULTIBIT && LIMBS == 0) ? WAVE_EARLY_CHUNKS : 0;  // otherwise requested here
    if (early < 1) key_request(ka0, ka1, b0, b1, 0);
    if (early < 2) key_request(kb0, kb1, b0, b1, 1);
    HX_SCHED_FENCE();
    flag_wait(f_ready_ot, epoch);
    const cplx *row0 = (w == 0 ? buf : obuf) + base_m3(ctx);
    const cplx *row1 = (w == 0 ? obuf : buf) + base_m3(ctx);
    // the key pointers must not be known before the wait, or the later requests are hoisted above it
    if constexpr (KEYBUF) {
      HX_OPAQUE_S(b0);
      HX_OPAQUE_S(b1);
    } else {
      HX_OPAQUE(b0);
      HX_OPAQUE(b1);
    }
    // Both rows are read from LDS — row 0 from the buffer of the wave that holds polynomial 0, row 1 from
    // the other one — so the roles are a scalar pointer choice: no per-point test of w, no selects, and
    // the registers of my own transform are free during the products.
    // the LDS reads run WAVE_MAC_PREFETCH points ahead of the products that consume them (issued just in
    // time, each pair exposed a full LDS latency to this wave)
    constexpr int PF = WAVE_MAC_PREFETCH;
    cplx xn0[PF > 0 ? PF : 1], xn1[PF > 0 ? PF : 1];
    HX_UNROLL
    for (int q = 0; q < PF; ++q) {
      xn0[q] = row0[q];
      xn1[q] = row1[q];
    }
    HX_UNROLL
    for (int ch = 0; ch < 4; ++ch) {
      cplx(&k0)[4] = (ch & 1) ? kb0 : ka0;
      cplx(&k1)[4] = (ch & 1) ? kb1 : ka1;
      HX_UNROLL
      for (int j = 0; j < 4; ++j) {
        const int r = ch * 4 + j;
        cplx x0, x1;
        if constexpr (PF > 0) {
          x0 = xn0[r % PF];
          x1 = xn1[r % PF];
          if (r + PF < 16) {
            xn0[r % PF] = row0[r + PF];
            xn1[r % PF] = row1[r + PF];
          }
        } else {
          x0 = row0[r];
          x1 = row1[r];
        }
        const cplx t = (idx == 0) ? cmul_first(x0, k0[j]) : cmul_add(x0, k0[j], dst[r]);
        dst[r] = cmul_add(x1, k1[j], t);
        // pin the product here: otherwise the FMAs are sunk below the flag wait that follows and
        // the key loads stay live across it
        HX_OPAQUE(dst[r].re);
        HX_OPAQUE(dst[r].im);
      }
      HX_SCHED_FENCE();
      if (ch < 2) key_request(k0, k1, b0, b1, ch + 2);
      if constexpr (decltype(fuse_pass1)::value) {
        inverse_pass1_group(dst, ch);
        HX_UNROLL
        for (int j = 0; j < 4; ++j) {
          HX_OPAQUE(dst[ch * 4 + j].re);
          HX_OPAQUE(dst[ch * 4 + j].im);
        }
      }
      HX_SCHED_FENCE();
    }
    HX_WAVE_SYNC();
    if (lane == 0) flag_set(r_done_me, epoch);
    if constexpr (decltype(wait_done)::value) flag_wait(r_done_ot, epoch);  // the partner must be done with my buffer before I reuse it
  };
  auto mac = [&](cplx (&dst)[16], cplx (&d)[16], cplx (&ka0)[4], cplx (&ka1)[4], cplx (&kb0)[4], cplx (&kb1)[4],
                 const cplx *b0, const cplx *b1, uint32_t idx, uint32_t epoch, auto fuse_pass1) {
    mac_impl(dst, d, ka0, ka1, kb0, kb1, b0, b1, idx, epoch, fuse_pass1, true, std::true_type{});
  };

  if constexpr (MULTIBIT) {
    // Multi-bit PBS (cc/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:116-156, :647-880): per group of g
    // mask elements  acc <- acc (x) GGSW_comb,  GGSW_comb = GGSW_0 + sum_s GGSW_s (.) FFT(X^{deg_s})  combined
    // in the Fourier domain (pbs_common.h).  The keybundle never exists in memory: each chunk of 4 points per
    // lane and row is accumulated in registers out of the 2^g key polynomials (coalesced 1 KiB wave loads, two
    // subsets in flight) and consumed by the multiply-accumulate at once.  A lane owns positions lane*16 + r, so
    // its monomial factors are  base_s(lane) * w16[(bitrev4(r) deg_s) mod 16]  with ONE gathered table entry per
    // subset and group; the 16th roots sit in the LDS table (wave-uniform index: broadcast reads).
    constexpr uint32_t g = GROUPING, per = 1u << g;
    const uint32_t groups = a.n / g;
#if WAVE_MB_PROBE
    uint64_t mbp_t = __builtin_amdgcn_s_memtime(), mbp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MBP(k)                                               \
  do {                                                       \
    const uint64_t t_ = __builtin_amdgcn_s_memtime();        \
    mbp[k] += t_ - mbp_t;                                    \
    mbp_t = t_;                                              \
  } while (0)
#else
#define MBP(k) do { } while (0)
#endif
    const cplx *key = (const cplx *)a.bsk;  // Fourier domain: [group][subset][level][row][col][slot]
    const size_t ggsw_c = (size_t)level * 4 * n;
    // tables.h mono_lane: entry [d][lane] = mono[((1 + 4 bitrev6(lane)) d) mod 2N] — a degree's 64 bases are one
    // 1 KB run (one coalesced request with the degree as the scalar offset) instead of 64 scattered table entries
    const HxBuffer mono_lane = hx_make_buffer(tb.mono_lane, 2u * N * 64u * 16u);
    const uint32_t lane16 = (uint32_t)lane * 16u;
#if WAVE_MB_PACE && !defined(TFHE_HIPEMU)
    // Pacing.  Workgroup b runs on XCD b % 8 (round-robin dispatch; used for speed only, never for correctness),
    // 32 workgroups of an XCD are resident at a time, in index order.  Every wave pair adds 1 to its XCD's counter
    // when it finishes a group; a pair of the XCD's r-th batch may start group g once the counter shows that all
    // pairs of the earlier batches are done and all pairs of its own batch have finished group g - WAVE_MB_PACE.
    // A wave that polls WAVE_MB_PACE_SPINS times in vain (peers not resident: fewer CUs than expected) stops
    // pacing, so the scheme cannot deadlock.
    // SHARE: the two quads of a workgroup run half a group apart (see mac_turn below), so each quad index paces
    // with its likes: its own counter (another cache line), its own pair counts
    const uint32_t pclass = SHARE ? (uint32_t)(pair >> 1) : 0u;
    uint32_t *pace_ctr = a.pace + (blockIdx.x & 7u) * 32u + pclass * 16u;
    uint32_t pace_before = 0, pace_mine = 0;  // pairs of earlier batches of my XCD; pairs of my batch
    const bool pace_on = a.pace != nullptr && (!OCTET || WAVE_MB_PACE_OCTET != 0);
    bool pacing = pace_on;
    if (pace_on) {
      const uint32_t ppb = blockDim.x >> 7, xcd = blockIdx.x & 7u, my_batch = (blockIdx.x >> 3) / 32u;
      for (uint32_t j = 0; j < (my_batch + 1) * 32u; ++j) {
        const uint64_t first = (uint64_t)(xcd + 8u * j) * ppb;
        uint32_t cnt = first >= a.num_samples ? 0u : (a.num_samples - first < ppb ? (uint32_t)(a.num_samples - first) : ppb);
        if constexpr (SHARE) cnt = pclass == 0 ? (cnt < 2u ? cnt : 2u) : (cnt > 2u ? cnt - 2u : 0u);  // pairs 0, 1 / 2, 3
        if (j < my_batch * 32u) pace_before += cnt; else pace_mine += cnt;
      }
    }
#endif
    // SHARE: the quad = waves 4q .. 4q+3 = (LWE 2q, column 0), (2q, 1), (2q+1, 0), (2q+1, 1)
    const uint64_t *lwe_q0 = lwe, *lwe_q1 = lwe;
    if constexpr (SHARE) {
      const uint32_t s0 = blockIdx.x * (blockDim.x >> 7) + (uint32_t)(pair & ~1);
      const uint32_t last = a.num_samples - 1;
      lwe_q0 = a.lwe_in + (size_t)a.in_idx[s0 < last ? s0 : last] * (a.n + 1);
      lwe_q1 = a.lwe_in + (size_t)a.in_idx[s0 + 1 < last ? s0 + 1 : last] * (a.n + 1);
    }
    const uint64_t *lwe_lane = lwe;  // OCTET: lane 16 L + s reads the mask of the workgroup's LWE L
    if constexpr (OCTET) {
      const uint32_t sL = blockIdx.x * 4u + ((uint32_t)lane >> 4), last = a.num_samples - 1;
      lwe_lane = a.lwe_in + (size_t)a.in_idx[sL < last ? sL : last] * (a.n + 1);
    }
    // SHARE synchronisation, all in the LDS flag words: word v = progress of wave v (quad_sync: each of the four
    // waves of a quad posts its count and waits for the other three), word 8 = mac_turn.  The multiply-accumulate
    // is what loads the key, the transforms are what computes: the two quads of a workgroup (one wave of each per
    // SIMD) take the multiply-accumulate IN TURNS, so one quad's key loads run under the other's transforms.
    uint32_t q_epoch = 0;
    auto quad_sync = [&]() {
      ++q_epoch;
      HX_WAVE_SYNC();
      if (lane == 0) flag_set(flags + wave, q_epoch);
      const int q0 = wave & ~3;
      HX_UNROLL
      for (int v = 0; v < 4; ++v)
        if (q0 + v != wave) flag_wait(flags + q0 + v, q_epoch);
    };
    const bool two_quads = WAVE_MB_TURNS && SHARE && blockDim.x == 512;
    auto mac_enter = [&](uint32_t m) {  // m-th multiply-accumulate of the launch (group, level)
      if (two_quads) flag_wait(flags + 8, 2u * m + (uint32_t)(pair >> 1));
    };
    auto mac_leave = [&](uint32_t m) {  // after the quad_sync that ends it
      if (two_quads && (wave & 3) == 0 && lane == 0) flag_set(flags + 8, 2u * m + (uint32_t)(pair >> 1) + 1u);
    };
    // e^{2 pi i t / 16} = mono[t N / 8], t wave-uniform
    auto w16_root = [&](uint32_t t) {
#if WAVE_MB_EXPERIMENT & 2  // timing experiment only (wrong results): no scalar load
      return cplx{0.5, 0.25};
#endif
#if WAVE_MB_W16_SCALAR
      return load_uniform_cplx(tb.mono, (uint32_t)(N / 8) * t);
#else
      return T[T_W16X + t];
#endif
    };
    const uint32_t ggsw_bytes = (uint32_t)(ggsw_c * sizeof(cplx));
    auto ldc = [](HxBuffer b, uint32_t voff, uint32_t soff) {
      const hx_f64x2 v = hx_buffer_load_f64x2(b, voff, soff);
      return cplx{v.x, v.y};
    };
#ifndef WAVE_MB_KEY_AUX
#define WAVE_MB_KEY_AUX 0  // cache-policy bits of the multi-bit key requests (hx.h)
#endif
    auto ldk = [](HxBuffer b, uint32_t voff, uint32_t soff) {
      const hx_f64x2 v = hx_buffer_load_f64x2<WAVE_MB_KEY_AUX>(b, voff, soff);
      return cplx{v.x, v.y};
    };
#if WAVE_MB_PREFETCH
    // The workgroups of an XCD walk the key in step (pacing), so every line of a group's key is a first touch for all
    // of them at once.  One load per wave and group, issued before the inverse transform, touches the NEXT group's
    // lines (the resident workgroups of the XCD share the 128-byte lines among them; speed only — nothing depends on
    // which lines a wave touches); its value is consumed a group later.  g = 3 per 4096 on one box: 42.9 -> 41.7 ms
    // (42.3 with the touch at the start of the group).  Not in OCTET mode: measured slower there (25.9 -> 26.9 ms).
    uint32_t pf_val = 0;
    uint32_t pf_first, pf_stride;
    {
      const uint32_t xcd_wgs = (gridDim.x + 7u) >> 3;  // workgroups of my XCD; up to 32 of them are resident
      const uint32_t sharers = xcd_wgs < 32u ? xcd_wgs : 32u;
      pf_first = ((((blockIdx.x >> 3) % sharers) * (blockDim.x >> 6) + (uint32_t)wave) * 64u + (uint32_t)lane) * 128u;
      pf_stride = sharers * (blockDim.x >> 6) * 64u * 128u;
    }
#endif
    for (uint32_t grp = 0; grp < groups; ++grp) {
      // monomial degrees of the 2^g - 1 non-empty subsets (:30-65): subset s selects mask element m of
      // the group when bit (g-1-m) of s is set
      auto degrees = [&](const uint64_t *lw, uint32_t (&dg)[per]) {
        uint64_t m[g];
        HX_UNROLL
        for (uint32_t q = 0; q < g; ++q) m[q] = lw[(size_t)grp * g + q];
        HX_UNROLL
        for (uint32_t sidx = 1; sidx < per; ++sidx) {
          uint64_t sum = 0;
          HX_UNROLL
          for (uint32_t q = 0; q < g; ++q)
            if ((sidx >> (g - 1 - q)) & 1) sum += m[q];
          dg[sidx] = HX_UNIFORM((uint32_t)modulus_switch(sum, LOG2N2));
        }
        dg[0] = 0;
      };
      // The gathered monomial bases (one table entry per lane and subset) are requested per LEVEL, right before
      // the multiply-accumulate that uses them, not per group: held across the digit and forward-transform phases
      // they cost 4 (2^g - 1) registers per LWE (56 at g = 3 in SHARE mode) exactly where the transform needs the
      // file, and the two-level g = 3 kernel spilled 55 registers (5.4 GB of scratch writes per launch).  The
      // lane-order table (tables.h mono_lane, 4 MB per device) is L2 resident; the requests are issued ahead of the
      // level's first key requests.
      auto bases = [&](const uint32_t (&dg)[per], cplx (&bs)[per]) {
        HX_UNROLL
        for (uint32_t sidx = 1; sidx < per; ++sidx)
          bs[sidx] = ldc(mono_lane, lane16, dg[sidx] * 1024u);
        bs[0] = cplx{1.0, 0.0};
      };
      uint32_t deg[per];
      uint32_t deg_b[SHARE ? per : 1];  // SHARE: deg / base belong to the quad's first LWE, these to its second
      if constexpr (OCTET) {
        // per subset, inside the multiply-accumulate
      } else if constexpr (SHARE) {
        degrees(lwe_q0, deg);
        degrees(lwe_q1, deg_b);
      } else {
        degrees(lwe, deg);
      }
      constexpr int MB_BASES = WAVE_MB_BASES >= 0 ? WAVE_MB_BASES : (LEVEL_CT >= 2 ? 1 : 0);
      cplx base[SHARE ? 1 : per];  // pair mode; (re)written per level unless MB_BASES == 0: not live across levels then
      if constexpr (MB_BASES == 0 && !OCTET && !SHARE) bases(deg, base);
      // the group's 2^g GGSWs as one buffer: uniform base in scalar registers, lane offset in one vector register
#if WAVE_MB_PACE && !defined(TFHE_HIPEMU)
      auto pace_wait = [&]() {
        if (pacing && grp >= (uint32_t)WAVE_MB_PACE) {
          const uint32_t need = pace_before * groups + pace_mine * (grp + 1u - (uint32_t)WAVE_MB_PACE);
          uint32_t spins = 0;
          while (__hip_atomic_load(pace_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(WAVE_MB_PACE_SLEEP);
            if (++spins > (uint32_t)WAVE_MB_PACE_SPINS * 8u / (uint32_t)WAVE_MB_PACE_SLEEP) {
              pacing = false;
              break;
            }
          }
        }
      };
      auto pace_arrive = [&]() {
        if (pace_on && valid && w == 0 && lane == 0)
          __hip_atomic_fetch_add(pace_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
#if !WAVE_MB_PACE_AT_KEY
      pace_wait();
      MBP(0);
#endif
#else
      auto pace_wait = []() {};
      auto pace_arrive = []() {};
#endif
      const HxBuffer gk = hx_make_buffer(key + (size_t)grp * per * ggsw_c, per * ggsw_bytes);
#if WAVE_MB_PREFETCH
      auto touch_next_key = [&]() {
        if constexpr (!OCTET || WAVE_MB_PREFETCH_OCTET == 1 || (WAVE_MB_PREFETCH_OCTET == 2 && LEVEL_CT != 1)) {
          HX_OPAQUE(pf_val);
          if (grp + (uint32_t)WAVE_MB_PF_DIST < groups) {
            const char *nk = (const char *)(key + (size_t)(grp + (uint32_t)WAVE_MB_PF_DIST) * per * ggsw_c);
            for (uint32_t off = pf_first; off < per * ggsw_bytes; off += pf_stride) pf_val ^= *(const uint32_t *)(nk + off);
          }
        }
      };
#if WAVE_MB_PF_POS == 1
      touch_next_key();
#endif
#endif
      cplx o[16];
      cplx oq_a[(SHARE || (OCTET && LEVEL_CT != 1)) ? 8 : 1], oq_b[(SHARE || (OCTET && LEVEL_CT != 1)) ? 8 : 1];  // SHARE: [point 4 u + j][column] at index 2 j + column, first / second LWE of the quad
      if constexpr (SHARE || (OCTET && LEVEL_CT != 1)) {  // OCTET, several levels: [LWE][point][column] over the two arrays
        HX_UNROLL
        for (int j = 0; j < 8; ++j) oq_a[j] = oq_b[j] = cplx{-0.0, -0.0};
      }
      // The accumulator is dead once its digits exist (the product OVERWRITES it): with several levels on
      // a decomposition of at most 30 bits only the 32-bit decomposer states stay live across the levels (32
      // registers instead of the 64 of the accumulator), cc/commons/math/decomposition/iter.rs:122-151
      constexpr bool STATE32 = LEVEL_CT >= 2 && BASE_LOG_CT != 0 && LEVEL_CT * BASE_LOG_CT <= 30;
      HX_UNROLL
      for (int r = 0; r < 16; ++r) o[r] = cplx{-0.0, -0.0};
      int32_t st_re[STATE32 ? 16 : 1], st_im[STATE32 ? 16 : 1];
      if constexpr (STATE32) {
        HX_UNROLL
        for (int r = 0; r < 16; ++r) {
          st_re[r] = decomp_init_state32((uint32_t)(acc_re[r] >> 32), BASE_LOG_CT, LEVEL_CT);
          st_im[r] = decomp_init_state32((uint32_t)(acc_im[r] >> 32), BASE_LOG_CT, LEVEL_CT);
        }
      }
      // OCTET: lane 16 L + s holds the degree of subset s of LWE L (one vector computation per group, a v_readlane per
      // use, instead of 4 g mask words in scalar registers and a scalar adder chain per use)
      uint32_t dv = 0;
      if constexpr (OCTET) {
        int ln = ctx0.lane;
        HX_OPAQUE(ln);
        const uint64_t *lw = lwe_lane + (size_t)grp * g;
        uint64_t sum = 0;
        HX_UNROLL
        for (uint32_t qq = 0; qq < g; ++qq) {
          const uint64_t mq = lw[qq];
          if (((uint32_t)ln >> (g - 1 - qq)) & 1u) sum += mq;
        }
        dv = (uint32_t)modulus_switch(sum, LOG2N2);
      }
      (void)dv;
      for (uint32_t idx = 0; idx < level; ++idx) {
        cplx d[16];
        HX_PRIO(WAVE_PRIO_MB_A);
        if constexpr (STATE32) {
          HX_UNROLL
          for (int r = 0; r < 16; ++r) {
            d[r] = cplx{(double)decompose_one_level32(BASE_LOG_CT, st_re[r]),
                        (double)decompose_one_level32(BASE_LOG_CT, st_im[r])};
            // the new state now: left alone, `state += carry` is sunk to the end of the level and both operands of
            // every one of the 32 additions stay live (or are spilled) across the whole multiply-accumulate
            HX_OPAQUE(st_re[r]);
            HX_OPAQUE(st_im[r]);
          }
          HX_WAVE_SYNC();
        } else {
          make_digits(d, 0, idx);
        }
        MBP(1);
        HX_PRIO(WAVE_PRIO_MB_B);
        wave_forward<0, WAVE_LIT_MB>(d, ctx);
        MBP(2);
        HX_PRIO(WAVE_PRIO_MB_C);
        if constexpr (OCTET && LEVEL_CT != 1) {
          // Several levels, all eight waves sharing every key load: as the one-level form below (wave v: both columns at
          // the points 2 v, 2 v + 1 for the four LWEs), but POINT-major — the products of the levels add up in registers
          // (16 complex), so only one point's keybundle (4 LWEs x 2 columns x 2 rows) is live next to them; every factor
          // is computed (no sign trick across the points: the other point's factors would have to be kept).
          WaveCtx cx = ctx0;
          HX_OPAQUE(cx.lane);
          const int ln = cx.lane;
          const uint32_t lane_off = (uint32_t)ln * 16u;
          const uint32_t v8 = (uint32_t)wave;
          const uint32_t brv = ((v8 & 1u) << 2) | (v8 & 2u) | (v8 >> 2);  // bitrev4(2 v) = bitrev3(v)
          const uint32_t lvl_off = idx * 4u * (uint32_t)n * 16u + v8 * 2048u;
          constexpr int SETS = WAVE_MB_OCTET_SETS, STEPS = 4 * (int)per;
          cplx x0[SETS], x1[SETS];
          auto request = [&](int set, int t) {  // step t = (point, subset, column): rows 0 and 1
            const uint32_t sidx = (uint32_t)((t >> 1) % (int)per), pp = (uint32_t)((t >> 1) / (int)per), c = (uint32_t)(t & 1);
            uint32_t o0 = lvl_off;
#if WAVE_MB_ROOT_JIT
            HX_OPAQUE_S(o0);
#endif
            const uint32_t rc = c * (uint32_t)n * 16u + pp * 1024u;
            x0[set] = ldk(gk, lane_off, sidx * ggsw_bytes + o0 + rc);
            x1[set] = ldk(gk, lane_off, sidx * ggsw_bytes + o0 + rc + 2u * (uint32_t)n * 16u);
          };
          uint32_t dg[1][4];
          cplx bs[1][4];  // consumed into the factors before the next subset's are requested
          auto request_bases = [&](uint32_t sidx) {
            HX_UNROLL
            for (int L = 0; L < 4; ++L) {
              dg[0][L] = hx_readlane(dv, L * 16 + (int)sidx);
#if WAVE_MB_EXPERIMENT & 4  // timing experiment only (wrong results): no base requests
              bs[0][L] = cplx{0.5, 0.25};
              HX_OPAQUE(bs[0][L].re);
              HX_OPAQUE(bs[0][L].im);
#else
              bs[0][L] = ldc(mono_lane, lane16, MB_EXP_ROW(dg[0][L]) * 1024u);
#endif
            }
          };
#if WAVE_MB_PACE_AT_KEY
          if (idx == 0) {
            pace_wait();
            MBP(0);
          }
#endif
          HX_UNROLL
          for (int t = 0; t < SETS && t < STEPS; ++t) request(t, t);
          if (per > 1) request_bases(1);
          HX_SCHED_FENCE();
#if !WAVE_MB_OCTET2_K_FIRST
          HX_BLOCK_SYNC_LDS();  // all eight transforms of this level are in the buffers (mapping M3: slot lane*17 + r)
          MBP(3);
          HX_SCHED_FENCE();
#endif
          const int fslot = base_m3(cx) + 2 * (int)v8;
          HX_UNROLL
          for (int pp = 0; pp < 2; ++pp) {
            cplx kk[4][2][2];  // [LWE][column][row]
            HX_UNROLL
            for (int si = 0; si < (int)per; ++si) {
              cplx mf[4];
              if (si >= 1) {
                const int nx = si + 1 < (int)per ? si + 1 : 1;
                HX_UNROLL
                for (int L = 0; L < 4; ++L) {
                  uint32_t dgl = dg[0][L];
#if WAVE_MB_ROOT_JIT && WAVE_MB_W16_SCALAR
                  HX_OPAQUE_S(dgl);
#endif
                  mf[L] = cmul_first(bs[0][L], w16_root(((brv + 8u * (uint32_t)pp) * dgl) & 15u));
                }
                if (per > 2 && (si + 1 < (int)per || pp == 0)) request_bases((uint32_t)nx);
              }
              HX_SCHED_FENCE();
              HX_UNROLL
              for (int c = 0; c < 2; ++c) {
                const int t = ((pp * (int)per + si) << 1) + c, set = t % SETS;
                HX_UNROLL
                for (int L = 0; L < 4; ++L) {
                  if (si == 0) {  // subset 0 is not rotated: it initialises the accumulators of all four LWEs
                    kk[L][c][0] = x0[set];
                    kk[L][c][1] = x1[set];
                  } else {
                    kk[L][c][0] = cmul_add(x0[set], mf[L], kk[L][c][0]);
                    kk[L][c][1] = cmul_add(x1[set], mf[L], kk[L][c][1]);
                    HX_OPAQUE(kk[L][c][0].re);
                    HX_OPAQUE(kk[L][c][0].im);
                    HX_OPAQUE(kk[L][c][1].re);
                    HX_OPAQUE(kk[L][c][1].im);
                  }
                }
                HX_SCHED_FENCE();
                if (t + SETS < STEPS) request(set, t + SETS);
                HX_SCHED_FENCE();
              }
            }
#if WAVE_MB_OCTET2_K_FIRST
            // the first point's keybundle needs the key and the mask only: the workgroup meets behind it, in front of the first
            // read of a transform — a wave that is early combines instead of waiting
            if (pp == 0) {
              HX_SCHED_FENCE();
              HX_BLOCK_SYNC_LDS();  // all eight transforms of this level are in the buffers (mapping M3: slot lane*17 + r)
              MBP(3);
              HX_SCHED_FENCE();
            }
#endif
            // this point's products with the digit transforms of the four LWEs, added to the earlier levels'
            HX_UNROLL
            for (int L = 0; L < 4; ++L) {
              const cplx xa0 = ((const cplx *)(smem + (size_t)(2 * L) * BUF_BYTES) + fslot)[pp];
              const cplx xa1 = ((const cplx *)(smem + (size_t)(2 * L + 1) * BUF_BYTES) + fslot)[pp];
              cplx(&oq)[8] = L < 2 ? oq_a : oq_b;
              HX_UNROLL
              for (int c = 0; c < 2; ++c) {
                cplx &acc = oq[((L & 1) * 2 + pp) * 2 + c];
                acc = cmul_add(xa1, kk[L][c][1], cmul_add(xa0, kk[L][c][0], acc));
                HX_OPAQUE(acc.re);
                HX_OPAQUE(acc.im);
              }
            }
            HX_SCHED_FENCE();
          }
          // every wave is done with this level's transforms: the next level's may take the buffers (the last level: below)
          MBP(4);
          if (idx + 1 < level) HX_BLOCK_SYNC_LDS();
          MBP(5);
        } else
        if constexpr (OCTET) {
          // All eight waves of the workgroup share every key load: wave v combines, for ALL FOUR LWEs, the keybundle of
          // BOTH columns at the points r = 2 v, 2 v + 1 of a lane's 16.  Subset-major: the keybundle (4 LWEs x 2 points x
          // 2 columns x 2 rows) stays in registers across the 2^g subsets.  A monomial factor serves the four key elements
          // of its point (both rows, both columns), and the factor of the odd point is the even point's up to the sign
          // (-1)^deg (bitrev4(2 v + 1) = bitrev4(2 v) + 8: half a turn of the 16th root per unit of the degree) — 2^g - 1
          // factors per LWE and group are computed, not 4 (2^g - 1) as with one column at four points per wave.  The
          // products read and write the SAME slots of the eight buffers (points 2 v, 2 v + 1 belong to this wave alone), so
          // nothing separates them: two workgroup barriers per group.
          WaveCtx cx = ctx0;
          HX_OPAQUE(cx.lane);
          const int ln = cx.lane;
          const uint32_t lane_off = (uint32_t)ln * 16u;
          const uint32_t v8 = (uint32_t)wave;
          const uint32_t brv = ((v8 & 1u) << 2) | (v8 & 2u) | (v8 >> 2);  // bitrev4(2 v) = bitrev3(v)
          const uint32_t lvl_off = idx * 4u * (uint32_t)n * 16u + v8 * 2048u;
          constexpr int SETS = WAVE_MB_OCTET_SETS, RW = 4, STEPS = RW * (int)per;
          cplx x0[SETS], x1[SETS];
          auto request = [&](int set, int t) {  // step t = (subset, point, column): rows 0 and 1
            const uint32_t sidx = (uint32_t)(t / RW);
            const uint32_t pc = (uint32_t)(t % RW);  // 2 p + c
            uint32_t o0 = lvl_off;
#if WAVE_MB_ROOT_JIT
            // the scalar offset of a request is one addition: made here, not 2 x 4 x 2^g of them ahead of the level and carried
            // through vector-register lanes
            HX_OPAQUE_S(o0);
#endif
            const uint32_t rc = (pc & 1u) * (uint32_t)n * 16u + (pc >> 1) * 1024u;
            x0[set] = ldk(gk, lane_off, sidx * ggsw_bytes + o0 + rc);
            x1[set] = ldk(gk, lane_off, sidx * ggsw_bytes + o0 + rc + 2u * (uint32_t)n * 16u);
          };
          uint32_t dg[2][4];
          cplx bs[2][4];
          auto request_bases = [&](uint32_t sidx) {
            HX_UNROLL
            for (int L = 0; L < 4; ++L) {
              dg[sidx & 1][L] = hx_readlane(dv, L * 16 + (int)sidx);
#if WAVE_MB_EXPERIMENT & 4  // timing experiment only (wrong results): no base requests
              bs[sidx & 1][L] = cplx{0.5, 0.25};
              HX_OPAQUE(bs[sidx & 1][L].re);
              HX_OPAQUE(bs[sidx & 1][L].im);
#else
              bs[sidx & 1][L] = ldc(mono_lane, lane16, MB_EXP_ROW(dg[sidx & 1][L]) * 1024u);
#endif
            }
          };
#if WAVE_MB_PACE_AT_KEY
          if (idx == 0) {
            pace_wait();
            MBP(0);
          }
#endif
          HX_UNROLL
          for (int t = 0; t < SETS && t < STEPS; ++t) request(t, t);
          request_bases(1);
          HX_SCHED_FENCE();
#if !WAVE_MB_OCTET_K_FIRST
          HX_BLOCK_SYNC_LDS();  // all eight transforms are in the buffers (mapping M3: slot lane*17 + r)
          MBP(3);
          HX_SCHED_FENCE();
#endif
          cplx kq[4][RW][2];  // [LWE][2 p + c][row]
          HX_UNROLL
          for (int si = 0; si < (int)per; ++si) {
            if (si >= 1 && si + 1 < (int)per) request_bases((uint32_t)si + 1);
            HX_SCHED_FENCE();
            cplx mf[4];
            if (si >= 1) {
              HX_UNROLL
              for (int L = 0; L < 4; ++L) {
                uint32_t dgl = dg[si & 1][L];
#if WAVE_MB_ROOT_JIT && WAVE_MB_W16_SCALAR
                HX_OPAQUE_S(dgl);
#endif
                mf[L] = cmul_first(bs[si & 1][L], w16_root((brv * dgl) & 15u));
              }
            }
            HX_UNROLL
            for (int j = 0; j < RW; ++j) {
              const int t = si * RW + j, set = t % SETS;
              if (si == 0) {  // subset 0 is not rotated: it initialises the accumulators of all four LWEs
                HX_UNROLL
                for (int L = 0; L < 4; ++L) {
                  kq[L][j][0] = x0[set];
                  kq[L][j][1] = x1[set];
                }
              } else {
                if (j == 2) {  // the odd point: the factor times (-1)^deg
                  HX_UNROLL
                  for (int L = 0; L < 4; ++L) {
                    const uint32_t sgn = dg[si & 1][L] << 31;
                    mf[L].re = f64_xor_hi(mf[L].re, sgn);
                    mf[L].im = f64_xor_hi(mf[L].im, sgn);
                  }
                }
                HX_UNROLL
                for (int L = 0; L < 4; ++L) {
                  kq[L][j][0] = cmul_add(x0[set], mf[L], kq[L][j][0]);
                  kq[L][j][1] = cmul_add(x1[set], mf[L], kq[L][j][1]);
                  HX_OPAQUE(kq[L][j][0].re);
                  HX_OPAQUE(kq[L][j][0].im);
                  HX_OPAQUE(kq[L][j][1].re);
                  HX_OPAQUE(kq[L][j][1].im);
                }
              }
              HX_SCHED_FENCE();
              if (t + SETS < STEPS) request(set, t + SETS);
              HX_SCHED_FENCE();
            }
          }
#if WAVE_MB_OCTET_K_FIRST
          HX_SCHED_FENCE();
          HX_BLOCK_SYNC_LDS();  // all eight transforms are in the buffers (mapping M3: slot lane*17 + r)
          MBP(3);
          HX_SCHED_FENCE();
#endif
          // products with the digit transforms of the four LWEs (rows = the two polynomials of an LWE's pair), both
          // columns, written back over the transforms they were made of; fma(a, b, -0.0) is the rounded product a b
          // with its sign of zero
          const int fslot = base_m3(cx) + 2 * (int)v8;
          HX_UNROLL
          for (int L = 0; L < 4; ++L) {
            cplx *f0 = (cplx *)(smem + (size_t)(2 * L) * BUF_BYTES) + fslot;      // row 0 in, column 0 out
            cplx *f1 = (cplx *)(smem + (size_t)(2 * L + 1) * BUF_BYTES) + fslot;  // row 1 in, column 1 out
            HX_UNROLL
            for (int pp = 0; pp < 2; ++pp) {
              const cplx xa0 = f0[pp], xa1 = f1[pp];
              const cplx c0 = cmul_add(xa1, kq[L][2 * pp][1], cmul_add(xa0, kq[L][2 * pp][0], cplx{-0.0, -0.0}));
              const cplx c1 = cmul_add(xa1, kq[L][2 * pp + 1][1], cmul_add(xa0, kq[L][2 * pp + 1][0], cplx{-0.0, -0.0}));
              f0[pp] = c0;
              f1[pp] = c1;
            }
          }
          HX_SCHED_FENCE();
          MBP(4);
          HX_BLOCK_SYNC_LDS();  // every wave has put its two points into the buffers
          const cplx *mine = buf + base_m3(cx);
          HX_UNROLL
          for (int r = 0; r < 16; ++r) o[r] = mine[r];
          HX_WAVE_SYNC();
        } else
        if constexpr (SHARE) {
          // Wave u of the quad (u = 2 (LWE of the quad) + column of the polynomial it transforms) combines, for BOTH LWEs,
          // the keybundle of BOTH columns at the points r = 4 u .. 4 u + 3 of a lane's 16, a PAIR of points at a time
          // (bitrev4(4 u + j) = 4 bitrev2(j) + bitrev2(u): j and j + 1 are half a turn of the 16th root apart, so the odd
          // point's monomial factor is the even point's times (-1)^deg).  A factor serves the four key elements of its
          // point; (2^g - 1) x 2 factors per LWE and level are computed (8 x (2^g - 1) with one column at eight points).
          // Only the bases of the subset at hand and of the next one are live (requested as the loop goes).
          WaveCtx cx = ctx0;
          HX_OPAQUE(cx.lane);
          const int ln = cx.lane;
          const uint32_t lane_off = (uint32_t)ln * 16u;
          const uint32_t u4 = (uint32_t)(wave & 3);
          const uint32_t bru = ((u4 & 1u) << 1) | (u4 >> 1);  // bitrev2(u)
          const uint32_t lvl_off = idx * 4u * (uint32_t)n * 16u + u4 * 4096u;
          const cplx *qbuf = (const cplx *)(smem + (size_t)(wave & ~3) * BUF_BYTES);
          const int fslot = base_m3(cx) + 4 * (int)u4;
          const cplx *fa0 = qbuf + fslot, *fa1 = (const cplx *)((const char *)qbuf + BUF_BYTES) + fslot;
          const cplx *fb0 = (const cplx *)((const char *)qbuf + 2 * BUF_BYTES) + fslot;
          const cplx *fb1 = (const cplx *)((const char *)qbuf + 3 * BUF_BYTES) + fslot;
          constexpr int SETS = WAVE_MB_SHARE_SETS > 0 ? WAVE_MB_SHARE_SETS : 3, STEPS = 8 * (int)per;
          cplx x0[SETS], x1[SETS];
          auto request = [&](int set, int t) {  // step t = (pair of points, subset, point of the pair, column): rows 0 and 1
            const uint32_t sidx = (uint32_t)((t >> 2) % (int)per);
            const uint32_t j = (uint32_t)((t >> 2) / (int)per) * 2u + (uint32_t)((t >> 1) & 1), c = (uint32_t)(t & 1);
            uint32_t o0 = lvl_off;
#if WAVE_MB_ROOT_JIT
            HX_OPAQUE_S(o0);
#endif
            const uint32_t rc = c * (uint32_t)n * 16u + j * 1024u;
            x0[set] = ldk(gk, lane_off, sidx * ggsw_bytes + o0 + rc);
            x1[set] = ldk(gk, lane_off, sidx * ggsw_bytes + o0 + rc + 2u * (uint32_t)n * 16u);
          };
          cplx bsa[2], bsb[2];
          auto request_bases = [&](uint32_t sidx) {
            bsa[sidx & 1] = ldc(mono_lane, lane16, deg[sidx] * 1024u);
            bsb[sidx & 1] = ldc(mono_lane, lane16, deg_b[sidx] * 1024u);
          };
#if WAVE_MB_PACE_AT_KEY
          if (idx == 0) {
            pace_wait();
            MBP(0);
          }
#endif
          HX_UNROLL
          for (int t = 0; t < SETS && t < STEPS; ++t) request(t, t);
          if (per > 1) request_bases(1);
          HX_SCHED_FENCE();
          // ---- all four transforms of the quad are in its buffers (mapping M3: slot lane*17 + r); the first key
          // requests are already on their way
          quad_sync();
          MBP(3);
          mac_enter(grp * level + idx);
          HX_SCHED_FENCE();
          HX_UNROLL
          for (int pr = 0; pr < 2; ++pr) {
            cplx kk[2][2][2][2];  // [point of the pair][column][LWE of the quad][row]
            HX_UNROLL
            for (int si = 0; si < (int)per; ++si) {
              cplx mfa, mfb;
              if (si >= 1) {
                // the bases of the next subset that has any (the next pair starts over at subset 1)
                const int nx = si + 1 < (int)per ? si + 1 : 1;
                const cplx ba = bsa[si & 1], bb = bsb[si & 1];
                uint32_t dga = deg[si], dgb = deg_b[si];
#if WAVE_MB_ROOT_JIT && WAVE_MB_W16_SCALAR
                HX_OPAQUE_S(dga);
                HX_OPAQUE_S(dgb);
#endif
                const uint32_t br = (uint32_t)pr * 4u + bru;
                mfa = cmul_first(ba, w16_root((br * dga) & 15u));
                mfb = cmul_first(bb, w16_root((br * dgb) & 15u));
                if (per > 2 && (si + 1 < (int)per || pr == 0)) request_bases((uint32_t)nx);
              }
              HX_SCHED_FENCE();
              HX_UNROLL
              for (int pc = 0; pc < 4; ++pc) {
                const int pj = pc >> 1, c = pc & 1;
                const int t = ((pr * (int)per + si) << 2) + pc, set = t % SETS;
                if (si == 0) {  // subset 0 is not rotated: it initialises the accumulators of both LWEs
                  kk[pj][c][0][0] = kk[pj][c][1][0] = x0[set];
                  kk[pj][c][0][1] = kk[pj][c][1][1] = x1[set];
                } else {
                  if (pc == 2) {  // the odd point: the factors times (-1)^deg
                    const uint32_t sa = deg[si] << 31, sb = deg_b[si] << 31;
                    mfa.re = f64_xor_hi(mfa.re, sa);
                    mfa.im = f64_xor_hi(mfa.im, sa);
                    mfb.re = f64_xor_hi(mfb.re, sb);
                    mfb.im = f64_xor_hi(mfb.im, sb);
                  }
                  kk[pj][c][0][0] = cmul_add(x0[set], mfa, kk[pj][c][0][0]);
                  kk[pj][c][0][1] = cmul_add(x1[set], mfa, kk[pj][c][0][1]);
                  kk[pj][c][1][0] = cmul_add(x0[set], mfb, kk[pj][c][1][0]);
                  kk[pj][c][1][1] = cmul_add(x1[set], mfb, kk[pj][c][1][1]);
                  HX_UNROLL
                  for (int q = 0; q < 4; ++q) {
                    HX_OPAQUE(kk[pj][c][q >> 1][q & 1].re);
                    HX_OPAQUE(kk[pj][c][q >> 1][q & 1].im);
                  }
                }
                HX_SCHED_FENCE();
                if (t + SETS < STEPS) request(set, t + SETS);
                HX_SCHED_FENCE();
              }
            }
            HX_UNROLL
            for (int pj = 0; pj < 2; ++pj) {
              const int j = pr * 2 + pj;
              const cplx xa0 = fa0[j], xa1 = fa1[j], xb0 = fb0[j], xb1 = fb1[j];
              HX_UNROLL
              for (int c = 0; c < 2; ++c) {
                oq_a[j * 2 + c] = cmul_add(xa1, kk[pj][c][0][1], cmul_add(xa0, kk[pj][c][0][0], oq_a[j * 2 + c]));
                oq_b[j * 2 + c] = cmul_add(xb1, kk[pj][c][1][1], cmul_add(xb0, kk[pj][c][1][0], oq_b[j * 2 + c]));
                HX_OPAQUE(oq_a[j * 2 + c].re);
                HX_OPAQUE(oq_a[j * 2 + c].im);
                HX_OPAQUE(oq_b[j * 2 + c].re);
                HX_OPAQUE(oq_b[j * 2 + c].im);
              }
            }
            HX_SCHED_FENCE();
          }
          MBP(4);
          if (idx + 1 < level) {
            quad_sync();  // every wave of the quad is done with the transforms of this level (the last level: see below)
            MBP(5);
            mac_leave(grp * level + idx);
          }
        } else
        {  // publish my transform, fetch the partner's, build the keybundle chunks and multiply-accumulate
          const uint32_t epoch = grp * level + idx + 1;
          WaveCtx cx = ctx0;
          HX_OPAQUE(cx.lane);
          const int ln = cx.lane;
          if (ln == 0) flag_set(f_ready_me, epoch);
          // rows 0, 1 of level idx, column w, subset s: byte s*ggsw_bytes + (((idx*2 + row)*2 + w)*n + slot)*16
          const uint32_t lane_off = (uint32_t)ln * 16u;
          const uint32_t row0_off = (((idx * 2 + 0) * 2 + (uint32_t)w) * n) * 16u;
          const uint32_t row1_off = (((idx * 2 + 1) * 2 + (uint32_t)w) * n) * 16u;
          // A step consumes one subset of one chunk of PTS points per lane (both rows).  SETS register sets
          // rotate: the set a step frees takes the request of the step SETS ahead, so SETS - 1 requests
          // (2 PTS coalesced 1 KiB wave loads each) are in flight while one set is accumulated.
          constexpr int PTS = WAVE_MB_PTS, SETS = WAVE_MB_SETS, CHUNKS = 16 / PTS, STEPS = CHUNKS * (int)per;
          if constexpr (MB_BASES == 1) bases(deg, base);
          cplx x0[SETS][PTS], x1[SETS][PTS];
          auto request = [&](int set, int t) {
            const uint32_t sidx = (uint32_t)(t % (int)per);
            const int ch = t / (int)per;
#ifdef WAVE_MB_EXPERIMENT_SKIP_LOADS  // timing experiment only (wrong results): 1 of N key requests is issued
            if (t % WAVE_MB_EXPERIMENT_SKIP_LOADS != 0) return;
#endif
            uint32_t o0 = row0_off, o1 = row1_off;
#if WAVE_MB_ROOT_JIT
            HX_OPAQUE_S(o0);
            HX_OPAQUE_S(o1);
#endif
            HX_UNROLL
            for (int j = 0; j < PTS; ++j) {
              x0[set][j] = ldk(gk, lane_off, sidx * ggsw_bytes + o0 + (uint32_t)(ch * PTS + j) * 1024u);
              x1[set][j] = ldk(gk, lane_off, sidx * ggsw_bytes + o1 + (uint32_t)(ch * PTS + j) * 1024u);
            }
          };
#if WAVE_MB_PACE_AT_KEY
          if (idx == 0) {
            pace_wait();
            MBP(0);
          }
#endif
          HX_UNROLL
          for (int t = 0; t < SETS && t < STEPS; ++t) request(t, t);
          HX_SCHED_FENCE();
          if constexpr (MB_BASES == 2) {  // behind the first key requests
            bases(deg, base);
            HX_SCHED_FENCE();
          }
          flag_wait(f_ready_ot, epoch);
          MBP(3);
          const cplx *row0 = (w == 0 ? buf : obuf) + base_m3(cx);
          const cplx *row1 = (w == 0 ? obuf : buf) + base_m3(cx);
          cplx kb0[PTS], kb1[PTS];
          HX_UNROLL
          for (int ch = 0; ch < CHUNKS; ++ch) {
            HX_UNROLL
            for (int si = 0; si < (int)per; ++si) {
              const int t = ch * (int)per + si, set = t % SETS;
              const uint32_t sidx = (uint32_t)si;
              if (si == 0) {  // subset 0 is not rotated: it initialises the chunk's accumulators
                HX_UNROLL
                for (int j = 0; j < PTS; ++j) {
                  kb0[j] = x0[set][j];
                  kb1[j] = x1[set][j];
                }
              } else {
                HX_UNROLL
                for (int j = 0; j < PTS; ++j) {
                  constexpr uint32_t br4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
                  uint32_t dgs = deg[sidx];
#if WAVE_MB_ROOT_JIT && WAVE_MB_W16_SCALAR
                  HX_OPAQUE_S(dgs);
#endif
                  const cplx wr = w16_root((br4[ch * PTS + j] * dgs) & 15u);
                  const cplx mf = cmul_first(base[sidx], wr);
                  kb0[j] = cmul_add(x0[set][j], mf, kb0[j]);
                  kb1[j] = cmul_add(x1[set][j], mf, kb1[j]);
                  // pin the accumulation to this step: left alone, instruction selection sinks three of the four
                  // products to the end of the chunk and every loaded set stays live until then
                  HX_OPAQUE(kb0[j].re);
                  HX_OPAQUE(kb0[j].im);
                  HX_OPAQUE(kb1[j].re);
                  HX_OPAQUE(kb1[j].im);
                }
              }
              HX_SCHED_FENCE();
              if (t + SETS < STEPS) request(set, t + SETS);
              HX_SCHED_FENCE();
            }
            // chunk complete: multiply-accumulate with the two digit transforms.  o starts at -0.0: fma(a, b, -0.0)
            // is the rounded product a b with its sign of zero, so the first level needs no separate form
            HX_UNROLL
            for (int j = 0; j < PTS; ++j) {
              const int r = ch * PTS + j;
              const cplx x0r = row0[r], x1r = row1[r];
              o[r] = cmul_add(x1r, kb1[j], cmul_add(x0r, kb0[j], o[r]));
              HX_OPAQUE(o[r].re);
              HX_OPAQUE(o[r].im);
            }
            HX_SCHED_FENCE();
          }
          HX_WAVE_SYNC();
          if (ln == 0) flag_set(r_done_me, epoch);
          MBP(4);
          flag_wait(r_done_ot, epoch);  // the partner must be done with my buffer before I reuse it
          MBP(5);
        }
      }
      if constexpr (OCTET && LEVEL_CT != 1) {
        // the sums over the levels go back over the last level's transforms (slots 2 v, 2 v + 1 of the eight buffers are
        // read and written by this wave alone: nothing to wait for), the workgroup meets, every wave collects its polynomial
        WaveCtx cx = ctx0;
        HX_OPAQUE(cx.lane);
        const int fslot = base_m3(cx) + 2 * wave;
        HX_UNROLL
        for (int L = 0; L < 4; ++L) {
          cplx(&oq)[8] = L < 2 ? oq_a : oq_b;
          HX_UNROLL
          for (int c = 0; c < 2; ++c) {
            cplx *dst = (cplx *)(smem + (size_t)(2 * L + c) * BUF_BYTES) + fslot;
            HX_UNROLL
            for (int pp = 0; pp < 2; ++pp) dst[pp] = oq[((L & 1) * 2 + pp) * 2 + c];
          }
        }
        HX_SCHED_FENCE();
        HX_BLOCK_SYNC_LDS();
        const cplx *mine = buf + base_m3(cx);
        HX_UNROLL
        for (int r = 0; r < 16; ++r) o[r] = mine[r];
        HX_WAVE_SYNC();
      }
      if constexpr (SHARE) {
        // my four points of the quad's four polynomials go to the waves that own them, over the transforms they were made
        // of (slots 4 u .. 4 u + 3 of the quad's buffers are read and written by this wave alone: nothing to wait for);
        // then the quad meets and every wave collects its polynomial
        {
          WaveCtx cx = ctx0;
          HX_OPAQUE(cx.lane);
          const int fslot = base_m3(cx) + 4 * (wave & 3);
          char *qb = smem + (size_t)(wave & ~3) * BUF_BYTES;
          HX_UNROLL
          for (int c = 0; c < 2; ++c) {
            cplx *da = (cplx *)(qb + (size_t)c * BUF_BYTES) + fslot, *db = (cplx *)(qb + (size_t)(2 + c) * BUF_BYTES) + fslot;
            HX_UNROLL
            for (int j = 0; j < 4; ++j) {
              da[j] = oq_a[j * 2 + c];
              db[j] = oq_b[j * 2 + c];
            }
          }
          quad_sync();
          mac_leave(grp * level + (level - 1));
          const cplx *mine = buf + base_m3(cx);
          HX_UNROLL
          for (int r = 0; r < 16; ++r) o[r] = mine[r];
          HX_WAVE_SYNC();
        }
      }
#if WAVE_MB_PREFETCH && WAVE_MB_PF_POS == 0
      touch_next_key();
#endif
#if WAVE_MB_PACE_AT_KEY
      pace_arrive();
#endif
      MBP(6);
      HX_PRIO(WAVE_PRIO_MB_D);
      wave_inverse_accumulate<0, true, false, false, 0, WAVE_LIT_MB>(o, acc_re, acc_im, ctx);
      HX_PRIO(WAVE_PRIO_MB_K);
      MBP(7);
#if !WAVE_MB_PACE_AT_KEY
      pace_arrive();
#endif
    }
#if WAVE_MB_PROBE
    if (g_wave_ts && lane == 0) {
      for (int k = 0; k < 8; ++k) g_wave_ts[((size_t)blockIdx.x * 8 + (size_t)wave) * 8 + k] = mbp[k];
    }
#endif
  } else if constexpr (LIMBS > 0) {
    struct alignas(16) U64x2 { uint64_t x, y; };
    // my polynomial's 2 N words in the accumulator scratch as one buffer: slot r*64 + lane holds coefficients c and 1024 + c.
    // Buffer addressing: ONE vector register of lane offsets serves the 16 requests of a sweep (the scalar offset r * 1024
    // rides in the instruction); as 64-bit pointers the compiler kept one per 1 KB step beyond the immediate range and
    // spilled them (32 spilled registers, 51 scratch accesses per CMUX in round 4)
#ifndef WAVE_SPLIT_ACC_BUFFER
#define WAVE_SPLIT_ACC_BUFFER 1
#endif
#ifndef WAVE_SPLIT_ACC_AUX
// cache-policy bits of the accumulator's loads and stores (hx.h: 1 = sc0, 2 = nt, 16 = sc1).  16 (agent scope: the lines do not
// stay in the CU's vector L1, which the four LWEs' key requests share): 131.2 -> 129.6 and 132.0 -> 131.0 ms per 4096 on two
// boxes; nt, sc0 and the combinations: no gain or worse (profiles/r05_ab_split_cache_policy.txt)
#define WAVE_SPLIT_ACC_AUX 16
#endif
    const uint64_t *gacc_base = a.acc_scratch ? a.acc_scratch + ((size_t)sample * 2 + (size_t)w) * N : nullptr;
    const HxBuffer gaccb = hx_make_buffer(gacc_base, (uint32_t)(N * sizeof(uint64_t)));
    U64x2 *gacc = (U64x2 *)gacc_base + lane;
    (void)gacc;
    (void)gaccb;
    auto acc_load = [&]() {
#if WAVE_SPLIT_PROBE == 2  // timing probe (wrong results): no accumulator traffic
      return;
#endif
      int ln = ctx0.lane;
      HX_OPAQUE(ln);
      HX_UNROLL
      for (int r = 0; r < 16; ++r) {
#if WAVE_SPLIT_ACC_BUFFER
        hx_buffer_load_u64x2<WAVE_SPLIT_ACC_AUX>(gaccb, (uint32_t)ln * 16u, (uint32_t)r * 1024u, acc_re[r], acc_im[r]);
#else
        const U64x2 v = gacc[r * 64];
        acc_re[r] = v.x;
        acc_im[r] = v.y;
#endif
      }
    };
    auto acc_store = [&]() {
#if WAVE_SPLIT_PROBE == 2
      return;
#endif
      int ln = ctx0.lane;
      HX_OPAQUE(ln);
      HX_UNROLL
      for (int r = 0; r < 16; ++r) {
#if WAVE_SPLIT_ACC_BUFFER
        hx_buffer_store_u64x2<WAVE_SPLIT_ACC_AUX>(gaccb, (uint32_t)ln * 16u, (uint32_t)r * 1024u, acc_re[r], acc_im[r]);
#else
        gacc[r * 64] = U64x2{acc_re[r], acc_im[r]};
#endif
      }
    };
    acc_store();
    stage_acc();  // the rotation of the first CMUX reads the staged copy
#if WAVE_SPLIT_PACE && !defined(TFHE_HIPEMU)
    // Pacing by mask index (the multi-bit loop's scheme, see there): every pair adds 1 to its XCD's counter per mask
    // element, executed or skipped; speed only, bounded spins, never needed for correctness
    uint32_t *pace_ctr = a.pace + (blockIdx.x & 7u) * 32u;
    uint32_t pace_before = 0, pace_mine = 0;
    bool pacing = a.pace != nullptr;
    {
      const uint32_t ppb = blockDim.x >> 7, xcd = blockIdx.x & 7u, my_batch = (blockIdx.x >> 3) / 32u;
      for (uint32_t j = 0; j < (my_batch + 1) * 32u; ++j) {
        const uint64_t first = (uint64_t)(xcd + 8u * j) * ppb;
        const uint32_t cnt = first >= a.num_samples ? 0u : (a.num_samples - first < ppb ? (uint32_t)(a.num_samples - first) : ppb);
        if (j < my_batch * 32u) pace_before += cnt; else pace_mine += cnt;
      }
    }
    auto pace_wait = [&](uint32_t i) {
      if (pacing && i >= (uint32_t)WAVE_SPLIT_PACE) {
        const uint32_t need = pace_before * a.n + pace_mine * (i + 1u - (uint32_t)WAVE_SPLIT_PACE);
        uint32_t spins = 0;
        while (__hip_atomic_load(pace_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (uint32_t)WAVE_MB_PACE_SPINS) {
            pacing = false;
            break;
          }
        }
      }
    };
    auto pace_arrive = [&]() {
      if (a.pace != nullptr && w == 0 && lane == 0)
        __hip_atomic_fetch_add(pace_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
#endif
#ifndef WAVE_SPLIT_RES_FWD
#define WAVE_SPLIT_RES_FWD 0  // split-key engine: forward pass F2's eight twiddles resident (ResidentTwiddles level 1)
#endif
#ifndef WAVE_SPLIT_RES_INV
#define WAVE_SPLIT_RES_INV 0  // ... inverse pass I2's two twiddles resident (level 2; four inverse transforms per CMUX use them)
#endif
    ResidentTwiddles res_sp;
    load_resident_twiddles<(WAVE_SPLIT_RES_INV >= 2 ? 2 : WAVE_SPLIT_RES_FWD)>(res_sp, T, lane);
    uint32_t worst = 0;  // OR of the low words of every product's bit pattern: bit 0 = some product was not within 1/4 of an integer
    // t = S + error, S integer: the Horner state takes the raw bits of t + 1.5 2^51 (= GL_SPLIT_C0 + 2 S + q, arith.h: the
    // factor 2 is out of the key, the bias of the four limbs cancels against the states' start value GL_SPLIT_R0, q is
    // the round-off check), R <- R 2^16 + bits (mod P), lazy Goldilocks forms: one f64 addition, one OR, seven integer
    // instructions per product (round 4: four f64 instructions with the distance to the nearest integer, eleven integer)
    auto fold = [&](uint64_t &R, double t) {
      const uint64_t bits = f64_bits(t + GL_SPLIT_MAGIC);
      worst |= (uint32_t)bits;
      HX_LAUNDER(worst);  // one OR per product: as a tree the compiler keeps all 32 low words of a limb live and spills
      R = gl_horner16(R, bits);
    };
    uint32_t it = 0;
    uint64_t mask_next = lwe[0];
    for (uint32_t i = 0; i < a.n; ++i) {
      const uint64_t mask_cur = mask_next;
      mask_next = lwe[i + 1];
      const uint32_t a_hat = HX_UNIFORM((uint32_t)modulus_switch(mask_cur, LOG2N2));
#if WAVE_SPLIT_SYNC && !defined(TFHE_HIPEMU)
      // the workgroup's LWEs in step (speed only: no memory ordering rides on it; waves that have left do not count):
      // every WAVE_SPLIT_SYNC-th mask element a bare s_barrier, executed before the a_hat == 0 skip by every wave
      if (i % (uint32_t)WAVE_SPLIT_SYNC == 0) __builtin_amdgcn_s_barrier();
#endif
#if WAVE_SPLIT_PACE && !defined(TFHE_HIPEMU)
      if (a_hat == 0) {
        pace_arrive();
        continue;
      }
      pace_wait(i);
#else
      if (a_hat == 0) continue;
#endif
      ++it;
      cplx d[16];
      HX_PRIO(WAVE_PRIO_A);
      make_digits(d, a_hat, 0);  // both operands of the rotation from the staged copy (the registers are not the accumulator's)
      HX_PRIO(WAVE_PRIO_B);
      wave_forward<WAVE_SPLIT_RES_FWD, WAVE_LIT_LIMBS != 0>(d, ctx, &res_sp);  // d = F, my row of the digit transform; also in my buffer (mapping M3)
      uint64_t R_re[16], R_im[16];
      HX_UNROLL
      for (int r = 0; r < 16; ++r) R_re[r] = R_im[r] = GL_SPLIT_R0;  // the limbs' bias cancels (arith.h)
      // one limb: product with the limb's key rows, back to the coefficients, into the Horner states.  LAST: F is dead
      // after the product, the accumulator is requested from device memory under the inverse transform
      auto limb_step = [&](uint32_t limb, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        cplx o[16], ka0[4], ka1[4], kb0[4], kb1[4];
        const cplx *b0, *b1;
#if WAVE_SPLIT_PROBE == 1  // timing probe (wrong results): every key request hits the same 64 KB
        key_rows(0, limb, b0, b1);
#elif WAVE_SPLIT_PROBE == 3  // timing probe (wrong results): ... the same 16 KB per wave (vector L1)
        key_rows(0, 0, b0, b1);
#else
        key_rows(i, limb, b0, b1);
#endif
        HX_PRIO(WAVE_PRIO_C);
        const uint32_t epoch = (it - 1) * (uint32_t)LIMBS + limb + 1;
#if WAVE_SPLIT_EARLY_RESTORE
        // Eight handshakes per CMUX (two per limb) instead of the classic loop's two: none of them is waited for where it is
        // posted.  "Ready" of limbs 1 .. 3 is posted by the previous limb's inverse transform as soon as its transposition
        // has left the buffer (F goes back there at once, after_load), "done" is waited for in front of the inverse
        // transposition's first store (before_store), two butterfly stages after the products.
        mac_impl(o, o, ka0, ka1, kb0, kb1, b0, b1, 0, epoch, std::integral_constant<bool, WAVE_FUSE_PASS1 != 0>{}, limb == 0,
                 std::false_type{});
        if constexpr (LAST) acc_load();
        HX_PRIO(WAVE_PRIO_D);
        auto before_store = [&]() { flag_wait(r_done_ot, epoch); };
        auto after_load = [&]() {
          if constexpr (!LAST) {  // my buffer held the inverse transposition: the pair needs F again
            WaveCtx cx = ctx0;
            HX_OPAQUE(cx.lane);
            cplx *p3 = buf + base_m3(cx);
            HX_UNROLL
            for (int r = 0; r < 16; ++r) p3[r] = d[r];
            HX_WAVE_SYNC();
            if (cx.lane == 0) flag_set(f_ready_me, epoch + 1);
          }
        };
        wave_inverse_accumulate<(WAVE_FUSE_PASS1 != 0) ? 1 : 0, false, false, true, WAVE_SPLIT_RES_INV, WAVE_LIT_LIMBS != 0>(
            o, acc_re, acc_im, ctx, &res_sp, before_store, after_load);
        HX_UNROLL
        for (int r = 0; r < 16; ++r) {
          fold(R_re[r], o[r].re);
          fold(R_im[r], o[r].im);
        }
#else
        mac(o, o, ka0, ka1, kb0, kb1, b0, b1, 0, epoch, std::integral_constant<bool, WAVE_FUSE_PASS1 != 0>{});
        if constexpr (LAST) acc_load();
        HX_PRIO(WAVE_PRIO_D);
        wave_inverse_accumulate<(WAVE_FUSE_PASS1 != 0) ? 1 : 0, false, false, true, WAVE_SPLIT_RES_INV, WAVE_LIT_LIMBS != 0>(o, acc_re, acc_im, ctx, &res_sp);
        HX_UNROLL
        for (int r = 0; r < 16; ++r) {
          fold(R_re[r], o[r].re);
          fold(R_im[r], o[r].im);
        }
        if constexpr (!LAST) {  // my buffer held the inverse transposition: the pair needs F again
          WaveCtx cx = ctx0;
          HX_OPAQUE(cx.lane);
          cplx *p3 = buf + base_m3(cx);
          HX_UNROLL
          for (int r = 0; r < 16; ++r) p3[r] = d[r];
          HX_WAVE_SYNC();
        }
#endif
      };
      HX_NO_UNROLL  // one body: unrolled, the scheduler overlaps the limbs and spills hundreds of registers
      for (uint32_t limb = 0; limb + 1 < (uint32_t)LIMBS; ++limb) limb_step(limb, std::false_type{});  // limb 0 = most significant
      limb_step((uint32_t)LIMBS - 1, std::true_type{});
      // acc += modswitch_to_2^64(product mod P) (ntt64.rs:162-177).  The registers hold MINUS the accumulator and the key
      // limbs are cut from MINUS the key, R = -product: modswitch(-x mod P) = -modswitch(x) mod 2^64 exactly (P odd: the
      // rounding is symmetric), so the update is one 64-bit addition
      HX_UNROLL
      for (int r = 0; r < 16; ++r) {
#if WAVE_SPLIT_TAIL_ASM
        acc_re[r] = gl_acc_modswitch_to_pow2_lazy(acc_re[r], R_re[r]);
        acc_im[r] = gl_acc_modswitch_to_pow2_lazy(acc_im[r], R_im[r]);
#else
        acc_re[r] += gl_modswitch_to_pow2_lazy(R_re[r]);
        acc_im[r] += gl_modswitch_to_pow2_lazy(R_im[r]);
#endif
      }
      acc_store();
      stage_acc();  // for the next CMUX's rotation (my buffer is free: the last inverse transposition is over)
#if WAVE_SPLIT_PACE && !defined(TFHE_HIPEMU)
      pace_arrive();
#endif
    }
    // an f64 product was not within 1/4 of an integer: reported through the scratch's flag (no trap: a trap kills the
    // whole HIP context of the process).  The bound is statistical, not a proof — worst-case magnitudes of 2^49 leave
    // about 4 bits of f64 headroom, the typical product is 2^12 smaller; an error beyond 1/2 would alias to a small
    // distance and pass, so the check guards against drift, not against arbitrary corruption.
    if ((worst & 1u) && a.roundoff_flag != nullptr) {
#if defined(TFHE_HIPEMU)
      *a.roundoff_flag = 1u;
      if (a.bad_samples != nullptr) a.bad_samples[sample] = 1u;
#else
      __hip_atomic_store(a.roundoff_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // this ciphertext goes through the integer kernel that follows on the stream (PbsArgs::bad_samples)
      if (a.bad_samples != nullptr) __hip_atomic_store(a.bad_samples + sample, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    acc_load();
  } else {
    stage_acc();
    ResidentTwiddles res_tw;
    if constexpr (LEVEL_CT == 1) load_resident_twiddles<WAVE_RESIDENT>(res_tw, T, lane);
#if WAVE_STAGGER && !defined(TFHE_HIPEMU)
    if (wave >= 4) {
      for (int u = 0; u < WAVE_STAGGER; ++u) __builtin_amdgcn_s_sleep(64);
    }
#endif
    uint32_t it = 0;  // executed iterations (flag epoch)
    uint64_t mask_next = lwe[0];
    for (uint32_t i = 0; i < a.n; ++i) {
      // mask element i was requested one iteration ago (lwe has n + 1 words, so i + 1 is in range)
      const uint64_t mask_cur = mask_next;
      mask_next = lwe[i + 1];
      // the same word in every lane: as a scalar, the rotation's bounds and signs cost no vector work
      const uint32_t a_hat = HX_UNIFORM((uint32_t)modulus_switch(mask_cur, LOG2N2));
#if WAVE_CLASSIC_SYNC && !defined(TFHE_HIPEMU)
      // the workgroup's LWEs back in step every so many mask elements (speed only, as in the split-key engine), the
      // waves of the upper half of the workgroup de-phased again behind it (WAVE_STAGGER)
      if (i % (uint32_t)WAVE_CLASSIC_SYNC == 0 && i != 0) {
        __builtin_amdgcn_s_barrier();
        if (WAVE_STAGGER && wave >= 4)
          for (int u = 0; u < WAVE_STAGGER; ++u) __builtin_amdgcn_s_sleep(64);
      }
#endif
      if (a_hat == 0) continue;  // uniform over the pair (bootstrap.rs:334)
      ++it;
      if constexpr (LEVEL_CT == 1) {
        cplx d[16], ka0[4], ka1[4], kb0[4], kb1[4];
        const cplx *b0, *b1;
        key_rows(i, 0, b0, b1);
        if (WAVE_EARLY_CHUNKS >= 1) key_request(ka0, ka1, b0, b1, 0);
        if (WAVE_EARLY_CHUNKS >= 2) key_request(kb0, kb1, b0, b1, 1);
        HX_SCHED_FENCE();
        HX_PRIO(WAVE_PRIO_A);
        make_digits(d, a_hat, 0);
        HX_PRIO(WAVE_PRIO_B);
        wave_forward<WAVE_RESIDENT, WAVE_UNIFORM_LITERALS != 0>(d, ctx, &res_tw);
        HX_PRIO(WAVE_PRIO_C);
        // in place: d becomes the Fourier-domain output of polynomial w, first inverse pass applied
#if WAVE_DEFER_DONE
        mac_impl(d, d, ka0, ka1, kb0, kb1, b0, b1, 0, it, std::integral_constant<bool, WAVE_FUSE_PASS1 != 0>{}, true, std::false_type{});
        HX_PRIO(WAVE_PRIO_D);
        wave_inverse_accumulate<(WAVE_FUSE_PASS1 != 0) ? 1 : 0, false, NEGACC, false, WAVE_RESIDENT, WAVE_UNIFORM_LITERALS != 0>(
            d, acc_re, acc_im, ctx, &res_tw, [&]() { flag_wait(r_done_ot, it); });
#else
        mac(d, d, ka0, ka1, kb0, kb1, b0, b1, 0, it, std::integral_constant<bool, WAVE_FUSE_PASS1 != 0>{});
        HX_PRIO(WAVE_PRIO_D);
        wave_inverse_accumulate<(WAVE_FUSE_PASS1 != 0) ? 1 : 0, false, NEGACC, false, WAVE_RESIDENT, WAVE_UNIFORM_LITERALS != 0>(d, acc_re, acc_im, ctx, &res_tw);
#endif
      } else {
        cplx o[16];
        for (uint32_t idx = 0; idx < level; ++idx) {
          cplx d[16], ka0[4], ka1[4], kb0[4], kb1[4];
          const cplx *b0, *b1;
          key_rows(i, idx, b0, b1);
          HX_SCHED_FENCE();
          HX_PRIO(WAVE_PRIO_A);
          make_digits(d, a_hat, idx);
          HX_PRIO(WAVE_PRIO_B);
          wave_forward(d, ctx);
          HX_PRIO(WAVE_PRIO_C);
          mac(o, d, ka0, ka1, kb0, kb1, b0, b1, idx, (it - 1) * level + idx + 1, std::false_type{});
        }
        HX_PRIO(WAVE_PRIO_D);
        wave_inverse_accumulate<0, false, NEGACC>(o, acc_re, acc_im, ctx);
      }
    }
  }

#if WAVE_PROBE_TS
  if (ts_rec && tid == 0) {
    ts_rec[3] = __builtin_amdgcn_s_memrealtime();
    ts_rec[4] = __builtin_amdgcn_s_memtime();
  }
#endif
  // ---- sample extraction (cc/algorithms/glwe_sample_extraction.rs:119-146); many-LUT outputs
  if (!valid) return;  // SHARE: a pair past the end of the batch
  const size_t out_sz = (size_t)N + 1;
  for (uint32_t t = 0; t < a.num_many_lut; ++t) {
    const uint32_t nth = t * a.lut_stride;
    uint64_t *out = a.lwe_out + (size_t)t * a.num_samples * out_sz + (size_t)a.out_idx[sample] * out_sz;
    if constexpr (!MULTIBIT && LIMBS == 0) {
      if (t == 0 && a.emit_a != nullptr && w == 0) {
        // the next keyswitch's operands for this ciphertext (PbsArgs::emit_a): word idx of the output mask -> k =
        // idx * level_pad + lv, byte k % 16 of lane (k half, row) of step k / 32; padded levels hold the shifted zero
        const uint32_t bl = a.emit_base_log, lv_n = a.emit_level, lp = a.emit_level_pad, half_b = 1u << (bl - 1);
        int8_t *dst = a.emit_a + ((size_t)(sample >> 5) * a.emit_steps) * 1024 + (size_t)(sample & 31) * 16;
        int32_t my_sa = 0;
        HX_UNROLL
        for (int r = 0; r < 16; ++r) {
          HX_UNROLL
          for (int half = 0; half < 2; ++half) {
            const uint32_t c = (uint32_t)(half * 1024 + r * 64 + lane);
            const uint64_t reg = half ? acc_im[r] : acc_re[r];
            const uint64_t v = ((c == 0) != NEGACC) ? reg : (uint64_t)0 - reg;  // out[idx], idx = c ? N - c : 0 (nth = 0)
            const uint32_t idx = c == 0 ? 0u : (uint32_t)N - c;
            int32_t state = decomp_init_state32((uint32_t)(v >> 32), bl, lv_n);
            uint32_t pk[2] = {0u, 0u};
            for (uint32_t lv = 0; lv < lp; ++lv) {
              const int32_t d = (lv < lv_n ? decompose_one_level32(bl, state) : 0) + (int32_t)half_b;
              my_sa += d;
              pk[lv >> 2] |= (uint32_t)d << (8 * (lv & 3));
            }
            const uint32_t k = idx * lp;
            uint32_t *o = (uint32_t *)(dst + (size_t)(k >> 5) * 1024 + ((k >> 4) & 1) * 512 + (k & 15));
            o[0] = pk[0];
            if (lp == 8) o[1] = pk[1];
          }
        }
        // sum over the wave (my buffer is free: the last inverse transform is over)
        int32_t *red = (int32_t *)buf;
        HX_WAVE_SYNC();
        red[lane] = my_sa;
        HX_WAVE_SYNC();
        if (lane == 0) {
          int32_t tot = 0;
          for (int l = 0; l < 64; ++l) tot += red[l];
          a.emit_suma[sample] = tot;
        }
      }
    }
    if constexpr (LIMBS > 0) {
      // the accumulator is rotated by -b_hat first: coefficient c moves to t = (c - b_hat) mod 2N, i.e. to j = t mod N
      // with its sign flipped when t >= N; then the extraction below on index j
      HX_UNROLL
      for (int r = 0; r < 16; ++r) {
        HX_UNROLL
        for (int half = 0; half < 2; ++half) {
          const uint32_t c = (uint32_t)(half * 1024 + r * 64 + lane);
          const uint64_t A = (uint64_t)0 - (half ? acc_im[r] : acc_re[r]);  // the registers hold minus the accumulator
          const uint32_t e = (c - b_hat) & (2u * N - 1u), j = e & (N - 1u);
          const uint64_t v = (e >= (uint32_t)N) ? (uint64_t)0 - A : A;
          if (w == 0) out[j <= nth ? nth - j : N + nth - j] = (j <= nth) ? v : (uint64_t)0 - v;
          else if (j == nth) out[N] = v;
        }
      }
    } else if (w == 0) {
      // mask: out[j] = A[nth - j] (j <= nth), -A[N + nth - j] otherwise; I hold A[c]
      HX_UNROLL
      for (int r = 0; r < 16; ++r) {
        uint32_t c = r * 64 + lane;
        out[c <= nth ? nth - c : N + nth - c] = ((c <= nth) != NEGACC) ? acc_re[r] : (uint64_t)0 - acc_re[r];
        c += 1024;
        out[c <= nth ? nth - c : N + nth - c] = ((c <= nth) != NEGACC) ? acc_im[r] : (uint64_t)0 - acc_im[r];
      }
    } else {
      HX_UNROLL
      for (int r = 0; r < 16; ++r) {
        if ((uint32_t)(r * 64 + lane) == nth) out[N] = NEGACC ? (uint64_t)0 - acc_re[r] : acc_re[r];
        if ((uint32_t)(1024 + r * 64 + lane) == nth) out[N] = NEGACC ? (uint64_t)0 - acc_im[r] : acc_im[r];
      }
    }
  }
}

}  // namespace wavek

// host side of WAVE_UNIFORM_LITERALS: the literals are the table entries they stand for (checked when the tables are
// built, tables.hip)
bool wave_literal_twiddles_match(const double *fwd, const double *inv) {
  static const int fidx[8] = {1, 2, 4, 6, 8, 10, 12, 14};
  for (int x = 0; x < 8; ++x)
    if (fwd[2 * fidx[x]] != wavek::LIT_F1[x][0] || fwd[2 * fidx[x] + 1] != wavek::LIT_F1[x][1]) return false;
  for (int j = 0; j < 4; ++j)
    if (inv[2 * (512 + 64 * j)] != wavek::LIT_E64[j][0] || inv[2 * (512 + 64 * j) + 1] != wavek::LIT_E64[j][1]) return false;
  return true;
}

// LWEs per workgroup (= per CU): as few as keeps every one of the 256 CUs busy — a lone wave pair runs an
// iteration in 7.6 us, four pairs sharing a CU need 12.4 us each
static unsigned lwes_per_block(uint32_t num_samples) {
#if WAVE_PROBE_TS || WAVE_MB_PROBE
  if (g_wave_force_per_block) return g_wave_force_per_block;
#endif
  const unsigned want = (num_samples + 255) / 256;
  return want < 1 ? 1 : (want > (unsigned)wavek::LWES_PER_BLOCK ? (unsigned)wavek::LWES_PER_BLOCK : want);
}

bool pbs_fft_wave_supported(uint32_t N, uint32_t glwe_dim, uint32_t level) {
  return N == 2048 && glwe_dim == 1 && level >= 1 && level <= 4;
}

template <int L, int B>
static void launch_wave_t(hipStream_t st, const PbsArgs &a, const FftTables &tb) {
  using namespace wavek;
  hx_set_dynamic_smem_once<pbs_fft_wave_kernel<L, B>>(SMEM_BYTES);
  const unsigned per_block = lwes_per_block(a.num_samples);
  const unsigned blocks = (a.num_samples + per_block - 1) / per_block;
  HX_LAUNCH((pbs_fft_wave_kernel<L, B>), dim3(blocks), dim3(128 * per_block), SMEM_BYTES, st, a, tb);
}

// exact engine, split-key form (LIMBS = 4): N = 2048, k = 1, one level, base_log 22 or 23
bool pbs_ntt_split_supported(uint32_t N, uint32_t glwe_dim, uint32_t level, uint32_t base_log) {
  return N == 2048 && glwe_dim == 1 && level == 1 && (base_log == 23 || base_log == 22);
}
template <int B>
static void launch_split_t(hipStream_t st, const PbsArgs &a, const FftTables &tb) {
  using namespace wavek;
  hx_set_dynamic_smem_once<pbs_fft_wave_kernel<1, B, 0, false, NTT_SPLIT_LIMBS>>(SMEM_BYTES);
  unsigned per_block = lwes_per_block(a.num_samples);
  if (per_block > WAVE_SPLIT_LWES) per_block = WAVE_SPLIT_LWES;
  const unsigned blocks = (a.num_samples + per_block - 1) / per_block;
#if WAVE_SPLIT_PACE
  if (a.pace) HX_CHECK(hipMemsetAsync(a.pace, 0, 8 * 32 * sizeof(uint32_t), st));
#endif
  HX_LAUNCH((pbs_fft_wave_kernel<1, B, 0, false, NTT_SPLIT_LIMBS>), dim3(blocks), dim3(128 * per_block), SMEM_BYTES, st, a,
            tb);
}
// a.bsk = split-key Fourier form (launch_bsk_to_split), a.acc_scratch = 2 N words per sample
void launch_pbs_ntt_split_wave(hipStream_t st, const PbsArgs &a, const FftTables &tb) {
  HX_PANIC_IF_FALSE(a.acc_scratch != nullptr, "split-key exact engine: no accumulator scratch");
  if (a.base_log == 23) launch_split_t<23>(st, a, tb);
  else launch_split_t<22>(st, a, tb);
}

bool pbs_multi_bit_wave_supported(uint32_t N, uint32_t glwe_dim, uint32_t level, uint32_t base_log, uint32_t grouping) {
  return N == 2048 && glwe_dim == 1 && level >= 1 && level <= 4 && base_log <= 31 && grouping >= 1 && grouping <= 4;
}

template <int L, int B, int G>
static void launch_wave_mb_t(hipStream_t st, const PbsArgs &a, const FftTables &tb) {
  using namespace wavek;
  if (a.pace) HX_CHECK(hipMemsetAsync(a.pace, 0, 8 * 32 * sizeof(uint32_t), st));
  unsigned per_block = lwes_per_block(a.num_samples);
  const bool share = !a.mb_no_share;  // hip_backend_set_fft_kernel(7) on the multi-bit entry point: pairs only (comparison)
  if (per_block == 3 && share) per_block = 4;  // 513 .. 768 LWEs: fuller workgroups that can share (4-12 % faster)
  const unsigned blocks = (a.num_samples + per_block - 1) / per_block;
  // full workgroups of a one-level set: all eight waves share the key loads of the four LWEs (OCTET)
  if constexpr ((L == 1 || (L >= 2 && WAVE_MB_OCTET >= 2)) && WAVE_MB_OCTET != 0) {
    if (per_block == 4 && share && !a.mb_no_octet) {
      hx_set_dynamic_smem_once<pbs_fft_wave_kernel<L, B, G, false, 0, true>>(SMEM_BYTES);
      HX_LAUNCH((pbs_fft_wave_kernel<L, B, G, false, 0, true>), dim3(blocks), dim3(512), SMEM_BYTES, st, a, tb);
      return;
    }
  }
  // an even number of LWEs per workgroup: quads of waves share the key loads of their two LWEs (SHARE)
  if (per_block % 2 == 0 && share) {
    hx_set_dynamic_smem_once<pbs_fft_wave_kernel<L, B, G, true>>(SMEM_BYTES);
    HX_LAUNCH((pbs_fft_wave_kernel<L, B, G, true>), dim3(blocks), dim3(128 * per_block), SMEM_BYTES, st, a, tb);
  } else {
    hx_set_dynamic_smem_once<pbs_fft_wave_kernel<L, B, G>>(SMEM_BYTES);
    HX_LAUNCH((pbs_fft_wave_kernel<L, B, G>), dim3(blocks), dim3(128 * per_block), SMEM_BYTES, st, a, tb);
  }
}

// a.grouping set; a.bsk = Fourier-domain multi-bit key
void launch_pbs_multi_bit_wave(hipStream_t st, const PbsArgs &a, const FftTables &tb) {
  if (a.grouping == 3 && a.level == 2 && a.base_log == 15) launch_wave_mb_t<2, 15, 3>(st, a, tb);  // PARAM_MULTI_BIT_GROUP_3_MESSAGE_2_CARRY_2
  else if (a.grouping == 3 && a.level == 2 && a.base_log == 14) launch_wave_mb_t<2, 14, 3>(st, a, tb);  // PARAM_GPU_MULTI_BIT_GROUP_3_MESSAGE_2_CARRY_2 (n = 879)
  else if (a.grouping == 4 && a.level == 1 && a.base_log == 22) launch_wave_mb_t<1, 22, 4>(st, a, tb);  // the GPU group-4 sets
  else if (a.grouping == 3 && a.level == 1 && a.base_log == 22) launch_wave_mb_t<1, 22, 3>(st, a, tb);  // the GPU group-3 / group-2 sets of the
  else if (a.grouping == 2 && a.level == 1 && a.base_log == 22) launch_wave_mb_t<1, 22, 2>(st, a, tb);  // gaussian families (one level, as g = 4)
  else if (a.grouping == 1) launch_wave_mb_t<0, 0, 1>(st, a, tb);
  else if (a.grouping == 2) launch_wave_mb_t<0, 0, 2>(st, a, tb);
  else if (a.grouping == 3) launch_wave_mb_t<0, 0, 3>(st, a, tb);
  else launch_wave_mb_t<0, 0, 4>(st, a, tb);
}

void launch_pbs_fft_wave(hipStream_t st, const PbsArgs &a, const FftTables &tb) {
  if (a.level == 1 && a.base_log == 23) launch_wave_t<1, 23>(st, a, tb);       // PARAM_MESSAGE_2_CARRY_2
  else if (a.level == 1 && a.base_log == 22) launch_wave_t<1, 22>(st, a, tb);  // multi-bit g=4 GPU sets
  else if (a.level == 2 && a.base_log == 15) launch_wave_t<2, 15>(st, a, tb);
  else launch_wave_t<0, 0>(st, a, tb);                                          // any (base_log, level)
}

}  // namespace tfhe_hip
