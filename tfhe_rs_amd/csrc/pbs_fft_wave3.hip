// pbs_fft_wave3.hip — throughput PBS for polynomial size 1024 (512-point transforms), k = 1 or 2.
//
// Same algorithm and the same bits as the generic kernel (cc/fft_impl/fft64/crypto/
// bootstrap.rs:294-380, ggsw.rs:483-697), organised like the N = 2048 throughput kernel: one WAVE per GLWE
// polynomial, k+1 waves per LWE, the accumulator (16 torus words per lane) and the transform points
// (8 complex per lane) in registers for the whole blind rotation, LDS only for the exchanges.
//
// A 512-point transform is three radix-8 passes over the position bits (8..6), (5..3), (2..0), each on the
// three register-index bits, with two LDS transpositions in between:
//   LA: position p = r*64 + lane             (coefficients c = p and 512 + p: the natural order)
//   LB: p = hi3*64 + r*8 + lo3               (lane = hi3*8 + lo3)
//   LC: p = lane*8 + r                       (the order the key is stored in: slot r*64 + lane)
// Slots are padded so that every ds_read/write_b128 of a transposition is bank-conflict free:
//   LA <-> LB through P1(p) = p + 8*(p >> 6):  LA slot = lane + 72 r ;  LB slot = hi3*72 + lo3 + 8 r
//   LB <-> LC through P2(p) = p + (p >> 3):    LB slot = hi3*72 + lo3 + 9 r ;  LC slot = 9 lane + r
// The forward result stays in the wave's buffer (LC slots) where the other waves of the LWE read it for
// the GGSW products; ready/done epochs in LDS order the k+1 waves.  16-byte table entries in LDS:
// the forward twiddles re-laid per pass so that a wave's read is contiguous or a broadcast, the inverse
// table and the untwist table in their natural order.
//
// Integer side as in pbs_fft_wave.hip: the registers hold MINUS the accumulator, the rotate-and-subtract is
// xor / one 64-bit add / xor, one-level digits come from a two-instruction rounding with an exact per-lane
// fallback, the torus conversion keeps its constants in registers, and the issue priority of a wave rises
// with the phase of its iteration.
#include "kernels.h"
#include <type_traits>

namespace tfhe_hip {
namespace wave3k {

constexpr int N = 1024, n = 512, LOG2N2 = 11;
constexpr int BUF_SLOTS = 576, BUF_BYTES = BUF_SLOTS * 16;
constexpr int MAX_WAVES = 12;  // 3 waves per SIMD (up to 168 VGPRs; the kernels take 130-150).  Measured at k = 2: 6 waves 118.6 k, 9: 105.7 k, 12: 140.0 k, 15 (128 VGPRs): 108.6 k PBS/s
// forward twiddles by pass
[[maybe_unused]] constexpr int T_FA = 0;     // stages 0..2: fwd[1..7]
constexpr int T_FB3 = 7;    // stage 3: fwd[8 + hi3]
constexpr int T_FB4 = 15;   // stage 4: fwd[16 + 2 hi3 + b] stored [b][hi3]
constexpr int T_FB5 = 31;   // stage 5: fwd[32 + 4 hi3 + q] stored [q][hi3]
constexpr int T_FC6 = 63;   // stage 6: fwd[64 + lane]
constexpr int T_FC7 = 127;  // stage 7: fwd[128 + 2 lane + b] stored [b][lane]
constexpr int T_FC8 = 255;  // stage 8: fwd[256 + 4 lane + q] stored [q][lane]
constexpr int T_INV = 511;  // inv[0..511], natural order (inv[half + j])
constexpr int T_U = 1023;   // untwist u[0..511]
constexpr int T_TOTAL = 1535;
constexpr int FLAGS_BYTES = 128;
constexpr size_t smem_bytes(int waves) { return (size_t)waves * BUF_BYTES + (size_t)T_TOTAL * 16 + FLAGS_BYTES; }

HX_DEV cplx ldg_c(const double *t, int idx) { return cplx{t[2 * idx], t[2 * idx + 1]}; }

#ifndef W3_UNIFORM_LITERALS
// 1: the twiddles that are the same in every lane and every launch (forward stages 0..2: fwd[1..7]; inverse half = 4:
// inv[4..7]) are literals of the instruction stream instead of broadcast reads of the LDS table, as in the N = 2048
// kernel (pbs_fft_wave.hip); the inverse butterflies whose twiddle is 1 or -i lose their products (same roundings).
// fwd[(1 << d) + g] does not depend on N, so these are the N = 2048 kernel's values; checked against the host tables
// when they are built (tables.hip).
#define W3_UNIFORM_LITERALS 1
#endif
constexpr double W3_LIT_F[7][2] = {{0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1},   // fwd[1]
                                   {0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2},   // fwd[2]
                                   {-0x1.87de2a6aea963p-2, 0x1.d906bcf328d46p-1},  // fwd[3] = i fwd[2]
                                   {0x1.f6297cff75cbp-1, 0x1.8f8b83c69a60bp-3},    // fwd[4]
                                   {-0x1.8f8b83c69a60bp-3, 0x1.f6297cff75cbp-1},   // fwd[5] = i fwd[4]
                                   {0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1},   // fwd[6]
                                   {-0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c8p-1}}; // fwd[7] = i fwd[6]
constexpr double W3_LIT_I5[2] = {0x1.6a09e667f3bcdp-1, -0x1.6a09e667f3bcdp-1};    // inv[5] = e^{-i pi / 4}; inv[7] = -i inv[5]

#if defined(TFHE_HIPEMU)
HX_DEV void flag_set(volatile uint32_t *f, uint32_t v) { *f = v; }
HX_DEV void flag_wait(volatile uint32_t *f, uint32_t v) {
  while (*f < v) hipemu::yield_barrier(0);
}
#define W3_PRIO(p) do { } while (0)
#else
HX_DEV void flag_set(uint32_t *f, uint32_t v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
HX_DEV void flag_wait(uint32_t *f, uint32_t v) {
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1);
}
#ifndef W3_PRIO_OFF
#define W3_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define W3_PRIO(p) do { } while (0)
#endif
#endif

// one radix-2 stage over register-index bit BIT of 8 points; tw(r) is the twiddle of the butterfly (r, r | 1<<BIT)
template <int BIT, class TW>
HX_DEV void stage8(cplx (&d)[8], TW tw) {
  HX_UNROLL
  for (int r = 0; r < 8; ++r)
    if (!(r & (1 << BIT))) bfly(d[r], d[r | (1 << BIT)], tw(r));
}
// backward stages with trivial twiddles: half = 1 and (half = 2, j = 0) plain, (half = 2, j = 1) w = -i
HX_DEV void bfly_plain(cplx &x, cplx &y) {
  const cplx o1{x.re + y.re, x.im + y.im}, o2{x.re - y.re, x.im - y.im};
  x = o1;
  y = o2;
}
HX_DEV void bfly_mi(cplx &x, cplx &y) {
  const cplx o1{x.re + y.im, x.im - y.re}, o2{x.re - y.im, x.im + y.re};
  x = o1;
  y = o2;
}

// the §4 butterfly (a, b) -> (a + s b, 2 a - (a + s b)) with s = 1 and s = -i: fma(b.re, 1, a.re) = a.re + b.re, the
// products by 0 vanish (the sign of an exact zero aside, which no later step reads)
HX_DEV void bfly_one_fma(cplx &a, cplx &b) {
  const double o1r = a.re + b.re, o1i = a.im + b.im;
  b.re = fma(2.0, a.re, -o1r);
  b.im = fma(2.0, a.im, -o1i);
  a.re = o1r;
  a.im = o1i;
}
HX_DEV void bfly_mi_fma(cplx &a, cplx &b) {
  const double o1r = a.re + b.im, o1i = a.im - b.re;
  b.re = fma(2.0, a.re, -o1r);
  b.im = fma(2.0, a.im, -o1i);
  a.re = o1r;
  a.im = o1i;
}

#ifndef W3_PROBE
#define W3_PROBE 0
#endif
// timing probe: what an exchange of three register bits with lane bits would cost as 48 lane swaps
HX_DEV void w3_probe_swaps(cplx (&d)[8]) {
  HX_UNROLL
  for (int rep = 0; rep < 3; ++rep)
    HX_UNROLL
    for (int r = 0; r < 4; ++r) {
      uint32_t a[4], b[4];
      __builtin_memcpy(a, &d[r], 16);
      __builtin_memcpy(b, &d[r + 4], 16);
      HX_UNROLL
      for (int k = 0; k < 4; ++k) {
        if (rep == 1) hx_permlane16_swap(a[k], b[k]);
        else hx_permlane32_swap(a[k], b[k]);
      }
      __builtin_memcpy(&d[r], a, 16);
      __builtin_memcpy(&d[r + 4], b, 16);
    }
}

struct Ctx {
  cplx *buf;      // my exchange buffer
  const cplx *T;  // table
  int lane;
};

#ifndef W3_ROT_PAIRS
// 1: the rotation fetches the staged words of two consecutive register rows (512 bytes apart) with one two-address
// LDS read; the ring's first 512 bytes are staged a second time behind its end (the buffer has 1,024 spare bytes), as in
// the N = 2048 kernel (pbs_fft_wave.hip, WAVE_ROT_PAIRS)
#define W3_ROT_PAIRS 1
#endif
#ifndef W3_RESIDENT
// twiddles kept in registers for the whole launch (read once in front of the CMUX loop), as in the N = 2048 kernel:
// 1 = forward stages 3..5 (seven values that depend on lane >> 3), 2 = also inverse half = 8, 16, 32 (seven on lane & 7)
#define W3_RESIDENT 1
#endif
struct Resident3 {
  cplx f3, f4[2], f5[4];   // forward stages 3, 4, 5
  cplx i8, i16[2], i32[4]; // inverse half = 8, 16, 32
};
template <int LEVEL>
HX_DEV void load_resident3(Resident3 &t, const cplx *T, int lane) {
  const int hi3 = lane >> 3, lo3 = lane & 7;
  if constexpr (LEVEL >= 1) {
    t.f3 = T[T_FB3 + hi3];
    t.f4[0] = T[T_FB4 + hi3];
    t.f4[1] = T[T_FB4 + 8 + hi3];
    HX_UNROLL
    for (int q = 0; q < 4; ++q) t.f5[q] = T[T_FB5 + 8 * q + hi3];
  }
  if constexpr (LEVEL >= 2) {
    t.i8 = T[T_INV + 8 + lo3];
    t.i16[0] = T[T_INV + 16 + lo3];
    t.i16[1] = T[T_INV + 24 + lo3];
    HX_UNROLL
    for (int q = 0; q < 4; ++q) t.i32[q] = T[T_INV + 32 + 8 * q + lo3];
  }
}

// digits (layout LA) -> transform, left in registers (layout LC) and in my buffer at the LC slots
template <int RES = 0>
HX_DEV void forward(cplx (&d)[8], Ctx c, const Resident3 *res = nullptr) {
  HX_OPAQUE(c.lane);
  const int lane = c.lane, hi3 = lane >> 3, lo3 = lane & 7;
  const cplx *T = c.T;
  {  // stages 0..2: position bits 8, 7, 6 = register bits 2, 1, 0; group = the bits above
#if W3_UNIFORM_LITERALS
    auto tw = [](int x) { return cplx{W3_LIT_F[x][0], W3_LIT_F[x][1]}; };
#else
    auto tw = [&](int x) { return T[T_FA + x]; };
#endif
    const cplx w0 = tw(0);
    stage8<2>(d, [&](int) { return w0; });
    HX_SCHED_FENCE();  // twiddle loads stay next to their stage (hoisted together they cost 80 registers)
    const cplx w1[2] = {tw(1), tw(2)};
    stage8<1>(d, [&](int r) { return w1[r >> 2]; });
    HX_SCHED_FENCE();
    const cplx w2[4] = {tw(3), tw(4), tw(5), tw(6)};
    stage8<0>(d, [&](int r) { return w2[r >> 1]; });
    HX_SCHED_FENCE();
  }
#if !(W3_PROBE & 1)  // timing probe (wrong results): bit 0 = no LA <-> LB transpositions, bit 1 = their cost as lane swaps
  {
    cplx *pa = c.buf + lane;  // LA slots of P1
    HX_UNROLL
    for (int r = 0; r < 8; ++r) pa[72 * r] = d[r];
  }
  HX_WAVE_SYNC();
#endif
#if W3_PROBE & 2
  w3_probe_swaps(d);
#endif
  {  // stages 3..5: position bits 5, 4, 3; group = hi3 . (register bits above)
#if !(W3_PROBE & 1)
    const cplx *pb = c.buf + hi3 * 72 + lo3;  // LB slots of P1
    HX_UNROLL
    for (int r = 0; r < 8; ++r) d[r] = pb[8 * r];
    HX_WAVE_SYNC();
#endif
    const cplx w3 = RES >= 1 ? res->f3 : T[T_FB3 + hi3];
    stage8<2>(d, [&](int) { return w3; });
    HX_SCHED_FENCE();
    const cplx w4[2] = {RES >= 1 ? res->f4[0] : T[T_FB4 + hi3], RES >= 1 ? res->f4[1] : T[T_FB4 + 8 + hi3]};
    stage8<1>(d, [&](int r) { return w4[r >> 2]; });
    HX_SCHED_FENCE();
    cplx w5[4];
    HX_UNROLL
    for (int q = 0; q < 4; ++q) w5[q] = RES >= 1 ? res->f5[q] : T[T_FB5 + 8 * q + hi3];
    stage8<0>(d, [&](int r) { return w5[r >> 1]; });
    HX_SCHED_FENCE();
    cplx *pb2 = c.buf + hi3 * 72 + lo3;  // LB slots of P2
    HX_UNROLL
    for (int r = 0; r < 8; ++r) pb2[9 * r] = d[r];
  }
  HX_WAVE_SYNC();
  {  // stages 6..8: position bits 2, 1, 0; group = lane . (register bits above)
    cplx *pc = c.buf + lane * 9;  // LC slots of P2
    HX_UNROLL
    for (int r = 0; r < 8; ++r) d[r] = pc[r];
    HX_WAVE_SYNC();
    const cplx w6 = T[T_FC6 + lane];
    stage8<2>(d, [&](int) { return w6; });
    HX_SCHED_FENCE();
    const cplx w7[2] = {T[T_FC7 + lane], T[T_FC7 + 64 + lane]};
    stage8<1>(d, [&](int r) { return w7[r >> 2]; });
    HX_SCHED_FENCE();
    const cplx w8[4] = {T[T_FC8 + lane], T[T_FC8 + 64 + lane], T[T_FC8 + 128 + lane], T[T_FC8 + 192 + lane]};
    stage8<0>(d, [&](int r) { return w8[r >> 1]; });
    HX_SCHED_FENCE();
    HX_UNROLL
    for (int r = 0; r < 8; ++r) pc[r] = d[r];  // published for the other waves of the LWE
  }
  HX_WAVE_SYNC();
}

// o (layout LC) -> backward transform, untwist, to the torus, added to the (negated) accumulator, which is
// then staged for the next rotation
template <bool NEG, int RES = 0>
HX_DEV void inverse_accumulate(cplx (&o)[8], uint64_t (&acc_re)[8], uint64_t (&acc_im)[8], Ctx c,
                               const Resident3 *res = nullptr) {
  HX_OPAQUE(c.lane);
  const int lane = c.lane, hi3 = lane >> 3, lo3 = lane & 7;
  const cplx *T = c.T;
  {  // half = 1, 2, 4: position bits 0, 1, 2 = register bits 0, 1, 2; j = the bits below
    HX_UNROLL
    for (int r = 0; r < 8; r += 2) bfly_plain(o[r], o[r + 1]);
    bfly_plain(o[0], o[2]);
    bfly_mi(o[1], o[3]);
    bfly_plain(o[4], o[6]);
    bfly_mi(o[5], o[7]);
    HX_SCHED_FENCE();
#if W3_UNIFORM_LITERALS
    {  // half = 4: twiddles 1, e^{-i pi/4}, -i, -i e^{-i pi/4}
      const cplx a1{W3_LIT_I5[0], W3_LIT_I5[1]};
      bfly_one_fma(o[0], o[4]);
      bfly(o[1], o[5], a1);
      bfly_mi_fma(o[2], o[6]);
      bfly(o[3], o[7], cplx{a1.im, -a1.re});
    }
#else
    const cplx w4[4] = {T[T_INV + 4], T[T_INV + 5], T[T_INV + 6], T[T_INV + 7]};
    stage8<2>(o, [&](int r) { return w4[r & 3]; });
#endif
    HX_SCHED_FENCE();
    cplx *pc = c.buf + lane * 9;  // LC slots of P2
    HX_UNROLL
    for (int r = 0; r < 8; ++r) pc[r] = o[r];
  }
  HX_WAVE_SYNC();
  {  // half = 8, 16, 32: position bits 3, 4, 5 = register bits 0, 1, 2 in LB; j = (register bits below) . lo3
    const cplx *pb2 = c.buf + hi3 * 72 + lo3;  // LB slots of P2
    HX_UNROLL
    for (int r = 0; r < 8; ++r) o[r] = pb2[9 * r];
    HX_WAVE_SYNC();
    const cplx w8 = RES >= 2 ? res->i8 : T[T_INV + 8 + lo3];
    stage8<0>(o, [&](int) { return w8; });
    HX_SCHED_FENCE();
    const cplx w16[2] = {RES >= 2 ? res->i16[0] : T[T_INV + 16 + lo3], RES >= 2 ? res->i16[1] : T[T_INV + 24 + lo3]};
    stage8<1>(o, [&](int r) { return w16[r & 1]; });
    HX_SCHED_FENCE();
    cplx w32[4];
    HX_UNROLL
    for (int q = 0; q < 4; ++q) w32[q] = RES >= 2 ? res->i32[q] : T[T_INV + 32 + 8 * q + lo3];
    stage8<2>(o, [&](int r) { return w32[r & 3]; });
    HX_SCHED_FENCE();
#if !(W3_PROBE & 1)
    cplx *pb = c.buf + hi3 * 72 + lo3;  // LB slots of P1
    HX_UNROLL
    for (int r = 0; r < 8; ++r) pb[8 * r] = o[r];
#endif
  }
  HX_WAVE_SYNC();
#if W3_PROBE & 2
  w3_probe_swaps(o);
#endif
  {  // half = 64, 128, 256: position bits 6, 7, 8 = register bits 0, 1, 2 in LA; j = (register bits below) . lane
#if !(W3_PROBE & 1)
    const cplx *pa = c.buf + lane;  // LA slots of P1
    HX_UNROLL
    for (int r = 0; r < 8; ++r) o[r] = pa[72 * r];
    HX_WAVE_SYNC();
#endif
    const cplx w64 = T[T_INV + 64 + lane];
    stage8<0>(o, [&](int) { return w64; });
    HX_SCHED_FENCE();
    const cplx w128[2] = {T[T_INV + 128 + lane], T[T_INV + 192 + lane]};
    stage8<1>(o, [&](int r) { return w128[r & 1]; });
    HX_SCHED_FENCE();
    const cplx w256[4] = {T[T_INV + 256 + lane], T[T_INV + 320 + lane], T[T_INV + 384 + lane], T[T_INV + 448 + lane]};
    stage8<2>(o, [&](int r) { return w256[r & 3]; });
    HX_SCHED_FENCE();
  }
  // untwist, back to the torus, accumulate (fft/mod.rs:311-330); the buffer is free: stage the new accumulator
  const TorusConsts kt = torus_consts();
  const cplx *Tu = T + T_U + lane;
  uint64_t *stg = (uint64_t *)c.buf + lane;
  HX_UNROLL
  for (int r = 0; r < 8; ++r) {
    const cplx u = Tu[r * 64];
    const double tr = NEG ? fma(o[r].im, u.im, -o[r].re * u.re) : fma(-o[r].im, u.im, o[r].re * u.re);
    const double ti = NEG ? fma(-o[r].im, u.re, -o[r].re * u.im) : fma(o[r].im, u.re, o[r].re * u.im);
    from_torus_add(acc_re[r], tr, kt);
    from_torus_add(acc_im[r], ti, kt);
    stg[r * 64] = acc_re[r];
    stg[512 + r * 64] = acc_im[r];
#if W3_ROT_PAIRS
    if (r == 0) stg[1024] = acc_re[0];  // the ring's first 64 words again behind its end
#endif
    if (r & 1) HX_SCHED_FENCE();
  }
  HX_WAVE_SYNC();
}

// K1 = k + 1 waves per LWE; L1: one decomposition level with base_log <= 30 (two-instruction digit)
template <int K1, bool L1>
__global__ void __launch_bounds__(MAX_WAVES * 64) pbs_fft_wave3_kernel(PbsArgs a, FftTables tb, uint32_t lwes_per_block) {
  HX_DYN_SMEM(smem);
  const int tid = threadIdx.x;
  const int wave = HX_UNIFORM(tid >> 6), lane = tid & 63;
  const int waves = (int)lwes_per_block * K1;
  const int slot = wave / K1, w = wave - slot * K1;  // LWE inside the workgroup, my polynomial
  const uint32_t level = a.level, base_log = a.base_log;
  cplx *buf = (cplx *)(smem + (size_t)wave * BUF_BYTES);
  uint64_t *buf64 = (uint64_t *)buf;
  const cplx *T = (const cplx *)(smem + (size_t)waves * BUF_BYTES);
#if defined(TFHE_HIPEMU)
  volatile uint32_t *flags = (volatile uint32_t *)(smem + (size_t)waves * BUF_BYTES + (size_t)T_TOTAL * 16);
#else
  uint32_t *flags = (uint32_t *)(smem + (size_t)waves * BUF_BYTES + (size_t)T_TOTAL * 16);
#endif
  {
    cplx *Tw = (cplx *)(smem + (size_t)waves * BUF_BYTES);
    for (int e = tid; e < T_TOTAL; e += (int)blockDim.x) {
      cplx v;
      if (e < T_FB3) v = ldg_c(tb.fwd, 1 + e);
      else if (e < T_FB4) v = ldg_c(tb.fwd, 8 + (e - T_FB3));
      else if (e < T_FB5) v = ldg_c(tb.fwd, 16 + 2 * ((e - T_FB4) & 7) + ((e - T_FB4) >> 3));
      else if (e < T_FC6) v = ldg_c(tb.fwd, 32 + 4 * ((e - T_FB5) & 7) + ((e - T_FB5) >> 3));
      else if (e < T_FC7) v = ldg_c(tb.fwd, 64 + (e - T_FC6));
      else if (e < T_FC8) v = ldg_c(tb.fwd, 128 + 2 * ((e - T_FC7) & 63) + ((e - T_FC7) >> 6));
      else if (e < T_INV) v = ldg_c(tb.fwd, 256 + 4 * ((e - T_FC8) & 63) + ((e - T_FC8) >> 6));
      else if (e < T_U) v = ldg_c(tb.inv, e - T_INV);
      else v = ldg_c(tb.untw, e - T_U);
      Tw[e] = v;
    }
    if (tid < FLAGS_BYTES / 4) flags[tid] = 0;
  }
  __syncthreads();

  const uint32_t sample = blockIdx.x * lwes_per_block + (uint32_t)slot;
  if (sample >= a.num_samples) return;  // the waves of an LWE leave together; no later block barrier
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * K1 * N + (size_t)w * N;
  const cplx *bsk = (const cplx *)a.bsk;
  const Ctx ctx0{buf, T, lane};
  auto *ready = flags + slot * 2 * K1, *done = ready + K1;  // one epoch word per wave of the LWE, each

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

  // ---- accumulator registers, NEGATED: coefficients (r*64 + lane) and (512 + r*64 + lane) of polynomial w
  uint64_t acc_re[8], acc_im[8];
  HX_UNROLL
  for (int r = 0; r < 8; ++r) {
    bool neg;
    uint32_t src = monomial_div_src(r * 64 + lane, b_hat, N, neg);
    uint64_t v = lut[src];
    acc_re[r] = neg ? v : (uint64_t)0 - v;
    src = monomial_div_src(512 + r * 64 + lane, b_hat, N, neg);
    v = lut[src];
    acc_im[r] = neg ? v : (uint64_t)0 - v;
  }
  auto stage_acc = [&]() {
    int ln = ctx0.lane;
    HX_OPAQUE(ln);
    uint64_t *p = buf64 + ln;
    HX_UNROLL
    for (int r = 0; r < 8; ++r) {
      p[r * 64] = acc_re[r];
      p[512 + r * 64] = acc_im[r];
    }
#if W3_ROT_PAIRS
    p[1024] = acc_re[0];  // the ring's first 64 words again behind its end
#endif
    HX_WAVE_SYNC();
  };

  // ct1 = acc * X^a_hat - acc, digit `idx`, as f64 points in layout LA (see pbs_fft_wave.hip make_digits:
  // with A = -acc staged and in the registers, ct1[c] = ((A[c] ^ M) + S) ^ M, S = A[(c - rr) mod N])
  auto make_digits_impl = [&](cplx (&d)[8], uint32_t a_hat, uint32_t idx, auto exact_tag) -> int32_t {
    constexpr bool EXACT = decltype(exact_tag)::value;
    int ln = ctx0.lane;
    HX_OPAQUE(ln);
    // (bit 31 of the byte offset u, which the LDS address ignores, carries the a_hat < N flag: M is one shift)
    const uint32_t ub = (uint32_t)(((int32_t)ln - (int32_t)(a_hat & (N - 1))) * 8) + ((a_hat & N) ? 0u : 0x80000000u);
    int32_t lowest = 0;
    uint32_t vzero = 0;
#ifndef WAVE_STAGED_SGPR_BASE
    HX_LAUNDER(vzero);  // base of the staged copy in a vector register (a scalar operand doubles the add's cost)
#endif
    const char *staged = (const char *)buf64 + vzero;
    uint64_t sp0[2] = {0, 0}, sp1[2] = {0, 0};
    (void)sp0;
    (void)sp1;
    HX_UNROLL
    for (int r = 0; r < 8; ++r) {
      const int32_t u0 = (int32_t)(ub + r * 512u), u1 = (int32_t)(ub + r * 512u + 4096u);
      const uint32_t m0 = (uint32_t)(u0 >> 31), m1 = (uint32_t)(u1 >> 31);
      const uint64_t M0 = ((uint64_t)m0 << 32) | m0, M1 = ((uint64_t)m1 << 32) | m1;
#if W3_ROT_PAIRS
      if ((r & 1) == 0) {  // rows r and r + 1 of both halves: the second word sits 512 bytes behind the first
        const uint64_t *q0 = (const uint64_t *)(staged + (u0 & 0x1ff8)), *q1 = (const uint64_t *)(staged + (u1 & 0x1ff8));
        sp0[0] = q0[0];
        sp0[1] = q0[64];
        sp1[0] = q1[0];
        sp1[1] = q1[64];
      }
      const uint64_t s0 = sp0[r & 1], s1 = sp1[r & 1];
#else
      const uint64_t s0 = *(const uint64_t *)(staged + (u0 & 0x1ff8));
      const uint64_t s1 = *(const uint64_t *)(staged + (u1 & 0x1ff8));
#endif
      const uint64_t x0 = ((acc_re[r] ^ M0) + s0) ^ M0, x1 = ((acc_im[r] ^ M1) + s1) ^ M1;
      if constexpr (L1) {
        if constexpr (EXACT) {
          d[r] = cplx{(double)decomp_digit_l1_hi((uint32_t)(x0 >> 32), base_log),
                      (double)decomp_digit_l1_hi((uint32_t)(x1 >> 32), base_log)};
        } else {
          const int32_t d0 = decomp_digit_l1_fast((uint32_t)(x0 >> 32), base_log);
          const int32_t d1 = decomp_digit_l1_fast((uint32_t)(x1 >> 32), base_log);
          lowest = d0 < lowest ? d0 : lowest;
          lowest = d1 < lowest ? d1 : lowest;
          d[r] = cplx{(double)d0, (double)d1};
        }
      } else {
        const int64_t d0 = decomp_digit(x0, base_log, level, idx), d1 = decomp_digit(x1, base_log, level, idx);
        d[r] = base_log <= 31 ? cplx{(double)(int32_t)d0, (double)(int32_t)d1} : cplx{i64_to_f64(d0), i64_to_f64(d1)};
      }
    }
    return lowest;
  };
  auto make_digits = [&](cplx (&d)[8], uint32_t a_hat, uint32_t idx) {
    if (idx != 0) stage_acc();  // the transposes of the previous level reused the buffer
    if constexpr (L1) {
      const int32_t lowest = make_digits_impl(d, a_hat, idx, std::false_type{});
      if (lowest == -(int32_t)(1u << (base_log - 1))) make_digits_impl(d, a_hat, idx, std::true_type{});
    } else {
      make_digits_impl(d, a_hat, idx, std::true_type{});
    }
    HX_WAVE_SYNC();
  };

  stage_acc();
  Resident3 res3;
  load_resident3<W3_RESIDENT>(res3, T, lane);
  uint32_t epoch = 0;
  uint64_t mask_next = lwe[0];
  for (uint32_t i = 0; i < a.n; ++i) {
    const uint64_t mask_cur = mask_next;  // requested one iteration ago (lwe has n + 1 words)
    mask_next = lwe[i + 1];
    const uint32_t a_hat = HX_UNIFORM((uint32_t)modulus_switch(mask_cur, LOG2N2));
    if (a_hat == 0) continue;  // uniform over the LWE (bootstrap.rs:334)
    cplx o[8];
    const uint32_t levels = L1 ? 1u : level;  // one level: no loop-carried products
    for (uint32_t idx = 0; idx < levels; ++idx) {
      ++epoch;
      cplx d[8];
      W3_PRIO(0);
      make_digits(d, a_hat, idx);
      W3_PRIO(1);
      forward<W3_RESIDENT>(d, ctx0, &res3);
      W3_PRIO(2);
      // my transform is published: tell the others, wait for theirs
      if (lane == 0) flag_set(ready + w, epoch);
      int ln = ctx0.lane;
      HX_OPAQUE(ln);
      const cplx *kcol = bsk + (((size_t)i * level + idx) * K1 * K1 + (size_t)w) * n + ln;  // row 0, column w
      // one key row at a time (a second buffer costs 32 registers and, at 3 waves per SIMD, spills); row q is
      // read from the buffer of wave q; products in (level, row) order (cc/fft_impl/fft64/crypto/ggsw.rs:616-697)
      HX_UNROLL
      for (int row = 0; row < K1; ++row) {
        cplx k[8];
        // the key pointer must not be known before the previous row's wait, or this row's loads are hoisted
        // above it and live (or spill) across the spin loop
        HX_OPAQUE(kcol);
        HX_UNROLL
        for (int r = 0; r < 8; ++r) k[r] = load_global_cplx(&kcol[(size_t)row * K1 * n + r * 64]);
        if (row != w) flag_wait(ready + row, epoch);
        const cplx *f = (const cplx *)(smem + (size_t)(slot * K1 + row) * BUF_BYTES) + ln * 9;
        HX_UNROLL
        for (int r = 0; r < 8; ++r) {
          const cplx x = f[r];
          o[r] = (idx == 0 && row == 0) ? cmul_first(x, k[r]) : cmul_add(x, k[r], o[r]);
          // pin the product here: otherwise the FMAs are sunk below the next row's flag wait and all the
          // key and transform values stay live across it (the accumulator then spills)
          HX_OPAQUE(o[r].re);
          HX_OPAQUE(o[r].im);
        }
        HX_SCHED_FENCE();
      }
      HX_WAVE_SYNC();
      if (lane == 0) flag_set(done + w, epoch);
      HX_UNROLL
      for (int q = 0; q < K1; ++q)
        if (q != w) flag_wait(done + q, epoch);  // nobody reads my buffer any more: it may be reused
    }
    W3_PRIO(3);
    inverse_accumulate<true, W3_RESIDENT>(o, acc_re, acc_im, ctx0, &res3);
  }

  // ---- sample extraction (cc/algorithms/glwe_sample_extraction.rs:119-146); many-LUT outputs
  const size_t out_sz = (size_t)(K1 - 1) * N + 1;
  for (uint32_t t = 0; t < a.num_many_lut; ++t) {
    const uint32_t nth = t * a.lut_stride;
    uint64_t *out = a.lwe_out + (size_t)t * a.num_samples * out_sz + (size_t)a.out_idx[sample] * out_sz;
    if (w < K1 - 1) {  // mask polynomial w: out[w N + j] = A[nth - j] (j <= nth), -A[N + nth - j] otherwise; I hold -A
      uint64_t *om = out + (size_t)w * N;
      HX_UNROLL
      for (int r = 0; r < 8; ++r) {
        uint32_t c = r * 64 + lane;
        om[c <= nth ? nth - c : N + nth - c] = c <= nth ? (uint64_t)0 - acc_re[r] : acc_re[r];
        c += 512;
        om[c <= nth ? nth - c : N + nth - c] = c <= nth ? (uint64_t)0 - acc_im[r] : acc_im[r];
      }
    } else {  // body
      HX_UNROLL
      for (int r = 0; r < 8; ++r) {
        if ((uint32_t)(r * 64 + lane) == nth) out[(size_t)(K1 - 1) * N] = (uint64_t)0 - acc_re[r];
        if ((uint32_t)(512 + r * 64 + lane) == nth) out[(size_t)(K1 - 1) * N] = (uint64_t)0 - acc_im[r];
      }
    }
  }
}

}  // namespace wave3k

// host side of W3_UNIFORM_LITERALS: the literals are the table entries they stand for (tables.hip checks when it builds
// the N = 1024 tables)
bool wave3_literal_twiddles_match(const double *fwd, const double *inv) {
  for (int x = 0; x < 7; ++x)
    if (fwd[2 * (1 + x)] != wave3k::W3_LIT_F[x][0] || fwd[2 * (1 + x) + 1] != wave3k::W3_LIT_F[x][1]) return false;
  return inv[2 * 4] == 1.0 && inv[2 * 4 + 1] == 0.0 && inv[2 * 5] == wave3k::W3_LIT_I5[0] && inv[2 * 5 + 1] == wave3k::W3_LIT_I5[1] &&
         inv[2 * 6] == 0.0 && inv[2 * 6 + 1] == -1.0 && inv[2 * 7] == wave3k::W3_LIT_I5[1] && inv[2 * 7 + 1] == -wave3k::W3_LIT_I5[0];
}

bool pbs_fft_wave3_supported(uint32_t N, uint32_t glwe_dim, uint32_t level) {
  return N == 1024 && (glwe_dim == 1 || glwe_dim == 2) && level >= 1 && level <= 4;
}

template <int K1>
static void launch_wave3_t(hipStream_t st, const PbsArgs &a, const FftTables &tb) {
  using namespace wave3k;
  // as many LWEs per workgroup (= per CU) as the LDS holds, fewer for small batches so that every CU has work
  const unsigned max_lwes = MAX_WAVES / K1;
  unsigned per_block = (a.num_samples + 255) / 256;
  per_block = per_block < 1 ? 1 : (per_block > max_lwes ? max_lwes : per_block);
  const unsigned blocks = (a.num_samples + per_block - 1) / per_block;
  const size_t smem = smem_bytes((int)(per_block * K1));
  const bool l1 = a.level == 1 && a.base_log >= 1 && a.base_log <= 30;
  if (l1) {
    hx_set_dynamic_smem_once<pbs_fft_wave3_kernel<K1, true>>(smem_bytes(MAX_WAVES));
    HX_LAUNCH((pbs_fft_wave3_kernel<K1, true>), dim3(blocks), dim3(64 * per_block * K1), smem, st, a, tb, per_block);
  } else {
    hx_set_dynamic_smem_once<pbs_fft_wave3_kernel<K1, false>>(smem_bytes(MAX_WAVES));
    HX_LAUNCH((pbs_fft_wave3_kernel<K1, false>), dim3(blocks), dim3(64 * per_block * K1), smem, st, a, tb, per_block);
  }
}

void launch_pbs_fft_wave3(hipStream_t st, uint32_t glwe_dim, const PbsArgs &a, const FftTables &tb) {
  if (glwe_dim == 1) launch_wave3_t<2>(st, a, tb);
  else launch_wave3_t<3>(st, a, tb);
}

}  // namespace tfhe_hip
