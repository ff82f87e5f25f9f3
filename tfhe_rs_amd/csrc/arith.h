// arith.h — device-side torus / decomposer / transform arithmetic of the PBS hot path.
//
// Each function states the tfhe-rs routine whose semantics it implements ("cc/" =
// tfhe/src/core_crypto/).  All integer work is exact u64 wrapping arithmetic; the f64
// pieces follow the fixed operation order of DESIGN.md §4 (explicit fma, compiled with
// -ffp-contract=off) so results do not depend on the kernel's stage grouping.
#pragma once
#include "hx.h"

namespace tfhe_hip {

// usable from host code too (table generation)
#define HX_HD __host__ __device__ __forceinline__
HX_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// ------------------------------------------------------------------ modulus switch
// cc/fft_impl/common.rs:10-23
HX_DEV uint64_t modulus_switch(uint64_t x, uint32_t log_modulus) {
  uint64_t t = x + (1ull << (64 - log_modulus - 1));
  return t >> (64 - log_modulus);
}

// per-mask-element terms of cc/algorithms/modulus_switch.rs:57-103; the caller sums
// `half` (wrapping u64) and `halving_doubled` (i64) over the mask — integer sums, so any
// reduction order is exact.
HX_DEV void centered_ms_terms(uint64_t a, uint32_t log_modulus, uint64_t &half, int64_t &halving_doubled) {
  uint64_t rounded = modulus_switch(a, log_modulus) << (64 - log_modulus);
  int64_t err = (int64_t)(rounded - a);
  int64_t h = err / 2;  // truncation toward zero
  half = (uint64_t)h;
  halving_doubled = 2 * h - err;
}
HX_DEV uint64_t centered_ms_finish(uint64_t sum_half, int64_t sum_halving_doubled, uint32_t log_modulus) {
  uint64_t sum_halving = (uint64_t)(sum_halving_doubled / 2);
  return sum_half - sum_halving - (1ull << (64 - log_modulus - 1));
}

// ------------------------------------------------------------------ signed decomposer
// cc/commons/math/decomposition/decomposer.rs:156-185
HX_DEV uint64_t decomp_init_state(uint64_t x, uint32_t base_log, uint32_t level) {
  const uint32_t rep = base_log * level;
  uint64_t res = x >> (64 - rep - 1);
  const uint64_t rounding_bit = res & 1;
  res += 1;
  res >>= 1;
  res &= ~0ull >> (64 - rep);
  const uint64_t need_balance = (((res - 1) | (rounding_bit << (rep - 1))) & res) >> (rep - 1);
  return res - (need_balance << rep);
}
// cc/commons/math/decomposition/iter.rs:122-151
HX_DEV int64_t decompose_one_level(uint32_t base_log, uint64_t &state) {
  const uint64_t res = state & ((1ull << base_log) - 1);
  state = (uint64_t)((int64_t)state >> base_log);
  const uint64_t carry = (((res - 1) | state) & res) >> (base_log - 1);
  state += carry;
  return (int64_t)(res - (carry << base_log));
}
// The same two functions on 32-bit registers, for base_log * level <= 30: the initial state then reads bits of
// the HIGH dword of x only and fits an int32 (|state| <= 2^(rep-1)), and every later state is smaller.
// Half the vector instructions of the 64-bit forms (no carries across dwords); checked against them by the
// keyswitch parity tests (the matrix-core keyswitch uses these).
HX_DEV int32_t decomp_init_state32(uint32_t x_hi, uint32_t base_log, uint32_t level) {
  const uint32_t rep = base_log * level;
  uint32_t res = x_hi >> (32 - rep - 1);
  const uint32_t rounding_bit = res & 1;
  res = ((res + 1) >> 1) & (0xFFFFFFFFu >> (32 - rep));
  const uint32_t need_balance = (((res - 1) | (rounding_bit << (rep - 1))) & res) >> (rep - 1);
  return (int32_t)(res - (need_balance << rep));
}
HX_DEV int32_t decompose_one_level32(uint32_t base_log, int32_t &state) {
  const uint32_t res = (uint32_t)state & ((1u << base_log) - 1);
  state >>= base_log;
  const uint32_t carry = (((res - 1) | (uint32_t)state) & res) >> (base_log - 1);
  state += (int32_t)carry;
  return (int32_t)(res - (carry << base_log));
}
// digit of level-matrix index `idx` (idx 0 <-> level l): run the iterator idx+1 times
HX_DEV int64_t decomp_digit(uint64_t x, uint32_t base_log, uint32_t level, uint32_t idx) {
  uint64_t st = decomp_init_state(x, base_log, level);
  int64_t d = 0;
  for (uint32_t t = 0; t <= idx; ++t) d = decompose_one_level(base_log, st);
  return d;
}

// Single-level decomposition (level == 1, base_log <= 31) from the HIGH dword of x only.
// With one level the digit returned by decompose_one_level equals the initial state:
//   res2 = state mod B = res; state' = state >> b is 0 or -1; carry = need_balance  =>  digit = res - nb*B,
// and x >> (63 - b) reads bits of the high dword only.  Checked against the two-step form by
// the test hooks (op 11) and by every l = 1 parity test.
HX_DEV int32_t decomp_digit_l1_hi(uint32_t x_hi, uint32_t base_log) {
  const uint32_t t = x_hi >> (31 - base_log);
  const uint32_t rb = t & 1;
  const uint32_t res = ((t + 1) >> 1) & ((1u << base_log) - 1);
  const uint32_t nb = (((res - 1) | (rb << (base_log - 1))) & res) >> (base_log - 1);
  return (int32_t)(res - (nb << base_log));
}

// Same digit from two instructions: round the high dword at bit (31 - base_log), shift arithmetically.
// Equal to decomp_digit_l1_hi for every input except some of those where this returns -B/2 (the
// decomposer maps the state B/2 to +B/2 or -B/2 by its rounding bit); callers that see -B/2 fall back.
HX_DEV int32_t decomp_digit_l1_fast(uint32_t x_hi, uint32_t base_log) {
  return (int32_t)(x_hi + (1u << (31 - base_log))) >> (32 - base_log);
}

// ------------------------------------------------------------------ monomial indexing
// coefficient j of  in * X^{deg}  (negacyclic), cc/algorithms/polynomial_algorithms.rs:662-727:
// returns the source index and whether the source is negated.
HX_DEV uint32_t monomial_mul_src(uint32_t j, uint32_t deg, uint32_t N, bool &neg) {
  const uint32_t r = deg & (N - 1);
  const bool odd = (deg & N) != 0;  // deg < 2N
  if (j < r) { neg = !odd; return N - r + j; }
  neg = odd;
  return j - r;
}
// coefficient j of  in * X^{-deg}, cc/algorithms/polynomial_algorithms.rs:544-583
HX_DEV uint32_t monomial_div_src(uint32_t j, uint32_t deg, uint32_t N, bool &neg) {
  const uint32_t r = deg & (N - 1);
  const bool odd = (deg & N) != 0;
  if (j < N - r) { neg = odd; return j + r; }
  neg = !odd;
  return j - (N - r);
}

// ------------------------------------------------------------------ f64 conversions
// exact i64 -> f64 (round to nearest even): hi*2^32 is exact, one rounding in the fma
HX_DEV double i64_to_f64(int64_t v) {
  const int32_t hi = (int32_t)(v >> 32);
  const uint32_t lo = (uint32_t)v;
  return fma((double)hi, 4294967296.0, (double)lo);
}
// integer-valued double in [-2^63, 2^63] -> i64; +2^63 folds onto -2^63 (two's complement
// torus value, the reference's SIMD conversion semantics, fft/x86.rs:1030-1112)
HX_DEV int64_t f64_to_i64_sat(double x) {
  if (x >= 9223372036854775808.0) return INT64_MIN;
  if (x <= -9223372036854775808.0) return INT64_MIN;
  return (int64_t)x;
}
// cc/commons/math/torus/mod.rs:73-79 (FromTorus) with nearest-even rounding:
//   g = rint((t - rint(t)) * 2^64) as a two's-complement u64 (+2^63 folds onto -2^63).
// Evaluated without an f64 -> i64 conversion: with f = t - rint(t) in [-1/2, 1/2] and F = f * 2^64,
//   h = rint(f * 2^32)  (|h| <= 2^31)  and  l = F - h * 2^32  (|l| <= 2^31, exact)
// give g = h * 2^32 + rint(l) exactly (h * 2^32 is even, so the tie rule is unchanged).  Both
// roundings are "add 1.5 * 2^52" tricks whose low mantissa dword is the two's-complement integer; the
// high dword of l + 1.5 * 2^52 is 0x43380000 + (rint(l) < 0 ? -1 : 0), i.e. the sign extension of
// rint(l) up to the constant that the first magic number cancels (2^32 - 0x43380000 added to it).
HX_DEV uint64_t f64_bits(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return u;
}
// x with the bits of `hi_mask` flipped in its high dword (bit 31: the sign) — one 32-bit xor
HX_DEV double f64_xor_hi(double x, uint32_t hi_mask) {
  const uint64_t u = f64_bits(x) ^ ((uint64_t)hi_mask << 32);
  double r;
  __builtin_memcpy(&r, &u, 8);
  return r;
}
HX_DEV uint64_t from_torus(double t) {
  const double MAGIC = 6755399441055744.0;               // 1.5 * 2^52
  const double MAGIC_H = 6755399441055744.0 + 3167223808.0;  // + (2^32 - 0x43380000)
  const double f = t - rint(t);
  const double hm = fma(f, 4294967296.0, MAGIC_H);
  const double h = hm - MAGIC_H;
  const double fl = fma(h, -2.3283064365386963e-10, f);                  // f - h 2^-32, exact
  const uint64_t lb = f64_bits(fma(fl, 18446744073709551616.0, MAGIC));  // rounds (f 2^64 - h 2^32) + MAGIC once
  const uint32_t hi = (uint32_t)f64_bits(hm) + (uint32_t)(lb >> 32);
  return ((uint64_t)hi << 32) | (uint64_t)(uint32_t)lb;
}

// The same conversion for a run of values that are ADDED to 64-bit accumulators: the five constants
// live in vector registers (as literals each use costs a move into the fma's destination), and the high
// part h * 2^32 only touches the accumulator's high dword, so it is a 32-bit add next to one 64-bit add.
struct TorusConsts {
  double two32, magic_h, minus_two_m32, two64, magic;
};
HX_DEV TorusConsts torus_consts() {
  TorusConsts k{4294967296.0, 6755399441055744.0 + 3167223808.0, -2.3283064365386963e-10, 18446744073709551616.0,
                6755399441055744.0};
  HX_OPAQUE(k.two32);
  HX_OPAQUE(k.magic_h);
  HX_OPAQUE(k.minus_two_m32);
  HX_OPAQUE(k.two64);
  HX_OPAQUE(k.magic);
  return k;
}
HX_DEV void from_torus_add(uint64_t &acc, double t, const TorusConsts &k) {
  const double f = t - rint(t);
  const double hm = fma(f, k.two32, k.magic_h);
  const double h = hm - k.magic_h;
  const double fl = fma(h, k.minus_two_m32, f);
  const uint64_t s = acc + f64_bits(fma(fl, k.two64, k.magic));
  uint32_t s_hi = (uint32_t)(s >> 32);
  HX_LAUNDER(s_hi);  // or the two steps are merged back into a second 64-bit addition of (h << 32)
  acc = ((uint64_t)(s_hi + (uint32_t)f64_bits(hm)) << 32) | (uint32_t)s;
}

struct alignas(16) cplx {
  double re, im;
};

// 16-byte load through a pointer KNOWN to be device memory.  A pointer that went through HX_OPAQUE has lost
// its address space: the compiler would emit FLAT loads, which also count on the LDS counter (lgkmcnt) and
// return out of order with the LDS reads, so every LDS wait behind them becomes a wait for the key.
// STREAM: read-once data (nontemporal).
template <bool STREAM = false>
HX_DEV cplx load_global_cplx(const cplx *p) {
#if defined(TFHE_HIPEMU) || defined(HX_KEY_FLAT_LOADS)  // (the second: A/B knob of tools/build_variants.py)
  return *p;
#else
  typedef double v2d __attribute__((ext_vector_type(2)));
  typedef const v2d __attribute__((address_space(1))) *gptr;
  const gptr g = (gptr)(const void *)p;
  const v2d x = STREAM ? __builtin_nontemporal_load(g) : *g;
  return cplx{x.x, x.y};
#endif
}

// One table entry at a WAVE-UNIFORM index of a table nobody writes during the launch, through the scalar data cache
// (s_load: no vector-memory request, no LDS read, the value arrives in scalar registers and feeds the f64
// instructions as their one scalar operand).  The constant address space is what makes the compiler pick the scalar
// path for a uniform address.
HX_DEV cplx load_uniform_cplx(const double *table, uint32_t idx) {
#if defined(TFHE_HIPEMU)
  return cplx{table[2 * idx], table[2 * idx + 1]};
#else
  typedef const double __attribute__((address_space(4))) *cptr;
  const cptr t = (cptr)(const void *)table;
  return cplx{t[2 * idx], t[2 * idx + 1]};
#endif
}

// DESIGN.md §4 butterfly: (a, b) -> (a + s*b, 2a - (a + s*b))
HX_DEV void bfly(cplx &a, cplx &b, const cplx s) {
  const double o1r = fma(-b.im, s.im, fma(b.re, s.re, a.re));
  const double o1i = fma(b.im, s.re, fma(b.re, s.im, a.im));
  b.re = fma(2.0, a.re, -o1r);
  b.im = fma(2.0, a.im, -o1i);
  a.re = o1r;
  a.im = o1i;
}
// first MAC term / following MAC terms, cc/fft_impl/fft64/crypto/ggsw.rs:652-676
HX_DEV cplx cmul_first(const cplx x, const cplx y) {
  return cplx{fma(-x.im, y.im, x.re * y.re), fma(x.im, y.re, x.re * y.im)};
}
HX_DEV cplx cmul_add(const cplx x, const cplx y, const cplx acc) {
  return cplx{fma(-x.im, y.im, fma(x.re, y.re, acc.re)), fma(x.im, y.re, fma(x.re, y.im, acc.im))};
}

// ------------------------------------------------------------------ Goldilocks field
// p = 2^64 - 2^32 + 1, tfhe-ntt/src/prime64/generic_solinas.rs:77-129 (fully reduced)
static constexpr uint64_t GL_P = 0xFFFFFFFF00000001ull;

HX_HD uint64_t gl_add(uint64_t a, uint64_t b) {
  const uint64_t neg_b = GL_P - b;
  return a >= neg_b ? a - neg_b : a + b;
}
HX_HD uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
// Carry-based forms for the transform loops.  "Lazy" values are any u64 congruent to the field element
// (2^64 = EPS mod p lets a wrapped carry be folded back in); gl_canon brings one to [0, p).
static constexpr uint64_t GL_EPS = 0xFFFFFFFFull;  // 2^64 mod p = 2^32 - 1
HX_HD uint64_t gl_canon(uint64_t r) {  // r >= p  <=>  r + EPS wraps, and the wrapped sum is r - p
  uint64_t t;
  return __builtin_add_overflow(r, GL_EPS, &t) ? t : r;
}
// x lazy, c canonical -> lazy (c < p: the folded-in carry cannot wrap a second time)
HX_HD uint64_t gl_add_lazy(uint64_t x, uint64_t c) {
  uint64_t s;
  const bool carry = __builtin_add_overflow(x, c, &s);
  return s + (carry ? GL_EPS : 0);
}
HX_HD uint64_t gl_sub_lazy(uint64_t x, uint64_t c) {
  uint64_t d;
  const bool borrow = __builtin_sub_overflow(x, c, &d);
  return d - (borrow ? GL_EPS : 0);
}
// a, b lazy -> canonical product.  Schoolbook 32x32 partial products chained through 64-bit
// multiply-adds (one v_mad_u64_u32 each), then  lo + hl 2^64 + hh 2^96 = lo + hl EPS - hh  (mod p),
// same value as tfhe-ntt/src/prime64/generic_solinas.rs:100-129.
HX_HD uint64_t gl_mul(uint64_t a, uint64_t b) {
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  const uint64_t t0 = (uint64_t)a0 * b0;
  const uint64_t t1 = (uint64_t)a0 * b1 + (t0 >> 32);
  const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t1;
  const uint64_t hi = (uint64_t)a1 * b1 + (t1 >> 32) + (t2 >> 32);
  const uint64_t lo = (t2 << 32) | (uint32_t)t0;
  const uint32_t hl = (uint32_t)hi, hh = (uint32_t)(hi >> 32);
  uint64_t u, v;
  const bool carry = __builtin_add_overflow((uint64_t)hl * GL_EPS, lo, &u);
  const bool borrow = __builtin_sub_overflow(u, (uint64_t)hh, &v);
  uint64_t r = v + (carry ? GL_EPS : 0);  // carry: u < 2^64 - 2^33, so this cannot wrap
  r = r - (borrow ? GL_EPS : 0);          // borrow: v > 2^64 - 2^32, so this cannot wrap
  return gl_canon(r);
}
// cc/commons/math/ntt/ntt64.rs:144-160, width 64:  (x*p + 2^63) >> 64
HX_HD uint64_t gl_modswitch_from_pow2(uint64_t x) {
  const uint64_t lo = x * GL_P;
  const uint64_t hi = mulhi64(x, GL_P);
  const uint64_t s = lo + (1ull << 63);
  return hi + (s < lo ? 1 : 0);
}
// cc/commons/math/ntt/ntt64.rs:162-177, width 64:  floor((v*2^64 + (p>>1)) / p), v < p.
// Division-free: with e = 2^32-1 (p = 2^64 - e), v*2^64 = v*p + v*e, so
//   q = v + floor(w / p),  w = v*e + (p>>1) < 2^97;  w = wh*2^64 + wl = wh*p + (wh*e + wl)
//   => floor(w/p) = wh + floor(r/p), r = wh*e + wl < 3*2^64  (wh < 2^33).
HX_HD uint64_t gl_modswitch_to_pow2(uint64_t v) {
  const uint64_t e = 0xFFFFFFFFull;
  const uint64_t h = GL_P >> 1;
  // w = v*e + h  (128-bit)
  uint64_t wl = v * e;
  uint64_t wh = mulhi64(v, e);
  const uint64_t t = wl + h;
  wh += (t < wl) ? 1 : 0;
  wl = t;
  // r = wh*e + wl  (up to 66 bits): rh:rl
  const uint64_t m = wh * e;  // wh < 2^33, e < 2^32 -> < 2^65: may overflow 64 bits
  const uint64_t mh = mulhi64(wh, e);
  uint64_t rl = m + wl;
  uint64_t rh = mh + ((rl < m) ? 1 : 0);
  // q2 = floor(r / p), r < 3*2^64 -> q2 in 0..3 ; subtract p while r >= p
  uint64_t q2 = 0;
  for (int it = 0; it < 4; ++it) {
    const bool ge = (rh > 0) || (rl >= GL_P);
    if (ge) {
      const uint64_t nl = rl - GL_P;
      rh -= (rl < GL_P) ? 1 : 0;
      rl = nl;
      ++q2;
    }
  }
  return v + wh + q2;
}

// ---- split-key form of the exact engine (pbs_fft_wave.hip, LIMBS mode): lean, branch-free forms
// Horner step  R <- R 2^16 + X  (mod p) on lazy values, X >= 0 below 2^63.  The 16 bits h shifted out come back as
// h EPS (2^64 = EPS mod p) TOGETHER with X in one multiply-add that cannot wrap (h EPS < 2^48); the one addition that
// can wrap is worth EPS once more, a second multiply-add (a wrapped V is below 2^63 + 2^48: it cannot wrap again).
// Seven instructions (round 4's form added X first and carried twice: eleven).
HX_HD uint64_t gl_horner16(uint64_t R, uint64_t X) {
  const uint32_t h = (uint32_t)(R >> 48);
  const uint64_t m = (uint64_t)h * GL_EPS + X;
  uint64_t V;
  uint32_t c = __builtin_add_overflow(R << 16, m, &V) ? 1u : 0u;
#if !defined(TFHE_HIPEMU) && defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(c));  // the carry as a 0 / 1 register: the compiler's select + zero-extension + 64-bit add are three instructions
#endif
  return (uint64_t)c * GL_EPS + V;
}
// The limb products leave the inverse transform as t = S + error, S integer, |S| <= 2^49.  bits(t + 1.5 2^51) =
// GL_SPLIT_C0 + 2 S + q: the unit of that binade is 1/2, so bit 0 (q) says "t was not within 1/4 of an integer" — the
// round-off check is an OR of the raw words — and the Horner above runs on the raw bit patterns, i.e. on 2 S.  The factor
// 2 is taken out of the KEY (bsk_to_split_kernel cuts -k / 2 mod p into limbs; the sign because the engine's registers
// hold minus the accumulator), the bias of the four limbs,
// C0 (1 + 2^16 + 2^32 + 2^48) mod p, out of the states' start value: R0 2^64 = -bias (mod p), so after the four steps the
// state is  sum_m 2 S_m 2^(16 (3 - m)) = -(digits (x) key)  mod p.
static constexpr double GL_SPLIT_MAGIC = 3377699720527872.0;           // 1.5 * 2^51
static constexpr uint64_t GL_SPLIT_C0 = 0x4328000000000000ull;         // its bit pattern
static constexpr uint64_t GL_SPLIT_BIAS = 0x86504327bcd779b0ull;       // GL_SPLIT_C0 * 0x0001000100010001 mod p
static constexpr uint64_t GL_SPLIT_R0 = 0x4327bcd779afbcd8ull;         // -GL_SPLIT_BIAS / (2^64 mod p) mod p
// k / 2 mod p for k < p (p odd): k even -> k / 2, k odd -> (k + p) / 2 = (k >> 1) + (p >> 1) + 1
HX_HD uint64_t gl_half(uint64_t k) { return (k >> 1) + ((k & 1) ? (GL_P >> 1) + 1 : 0); }
// gl_modswitch_to_pow2 of a LAZY value (any 64-bit v, standing for v mod p), seven instructions:
//   floor((v 2^64 + (p >> 1)) / p) = v + floor(w / p),  w = v e + h  (e = EPS = 2^64 - p, h = p >> 1);
//   with v = vh 2^32 + vl:  w = vh p + rho,  rho = vl e + h - vh   (vh p = vh 2^32 e + vh, so vh p + rho = v e + h);
//   0 < h - vh <= rho < 2^64 + 2^63 < 2 p, hence floor(w / p) = vh + [rho >= p], and rho >= p  <=>  rho + e >= 2^64
//   <=>  the 64-bit sum  vl e + (h + e - vh)  carries (h + e - vh < 2^64, vl e < 2^64).
// v and v + p give results 2^64 apart: no canonical form is needed in front of it.  Equal to gl_modswitch_to_pow2 for
// every v < p and to gl_modswitch_to_pow2(v mod p) for every v (test_arith_hooks_match_oracle).
HX_HD uint64_t gl_modswitch_to_pow2_lazy(uint64_t v) {
  const uint32_t vh = (uint32_t)(v >> 32), vl = (uint32_t)v;
  constexpr uint64_t K = (GL_P >> 1) + GL_EPS;
  uint64_t t;
  const uint32_t more = __builtin_add_overflow((uint64_t)vl * GL_EPS, K - vh, &t) ? 1u : 0u;
  return v + vh + more;
}

// acc + gl_modswitch_to_pow2_lazy(v) in six instructions.  The carry of the lazy form above reduces to a test of the
// halves of v:  vl e + (h + e - vh) >= 2^64  <=>  (vl + 2^31 - 1) 2^32 + (2^31 + ~vh - vl) >= 2^64  <=>  vl > 2^31 + 1, or
// vl = 2^31 + 1 and vh != 2^32 - 1  <=>  vl + (2^31 - 2) + [vh != 2^32 - 1] carries out of 32 bits — a borrow feeding a carry
// (test_arith_hooks_match_oracle op 14 against the big-integer formula, every edge of both halves).  On the device the
// sequence is written out (the compiler's own lowering of the lazy form is thirteen instructions, zero-extensions included);
// two wait states between a carry's producer and its consumer, as the compiler leaves them.
HX_HD uint64_t gl_acc_modswitch_to_pow2_lazy(uint64_t acc, uint64_t v) {
#if !defined(TFHE_HIPEMU) && defined(__HIP_DEVICE_COMPILE__)
  const uint32_t vh = (uint32_t)(v >> 32), vl = (uint32_t)v, k = 0x7FFFFFFEu;
  uint32_t t;
  uint64_t x, sc;
  asm("v_subrev_co_u32_e32 %[t], vcc, -1, %[vh]\n\t"      // borrow = [vh != 2^32 - 1]
      "v_mad_u64_u32 %[x], %[sc], %[vh], 1, %[v]\n\t"     // x = v + vh
      "s_nop 0\n\t"
      "v_addc_co_u32_e32 %[t], vcc, %[k], %[vl], vcc\n\t"  // carry = the "more" of the lazy form
      "v_lshl_add_u64 %[x], %[x], 0, %[acc]\n\t"
      "s_nop 0\n\t"
      "v_addc_co_u32_e64 %[t], %[sc], 0, 0, vcc\n\t"
      "v_mad_u64_u32 %[x], %[sc], %[t], 1, %[x]"
      : [t] "=&v"(t), [x] "=&v"(x), [sc] "=&s"(sc)
      : [vh] "v"(vh), [vl] "v"(vl), [v] "v"(v), [acc] "v"(acc), [k] "v"(k)  // k in a vector register: with the carry-in a scalar would be a second constant-bus operand
      : "vcc");
  return x;
#else
  const uint32_t vh = (uint32_t)(v >> 32), vl = (uint32_t)v;
  const uint64_t more = ((uint64_t)vl + 0x7FFFFFFEu + (vh != 0xFFFFFFFFu ? 1u : 0u)) >> 32;
  return acc + v + vh + more;
#endif
}

}  // namespace tfhe_hip
