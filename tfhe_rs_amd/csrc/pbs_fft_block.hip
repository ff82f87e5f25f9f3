// pbs_fft_block.hip — LATENCY PBS kernel for N = 2048, k = 1: one workgroup (8 waves) per LWE.
//
// Replaces the role of backends/tfhe-cuda-backend/cuda/src/pbs/programmable_bootstrap_cg_classic.cuh
// (the reference's low-latency path, one PBS spread over many thread blocks with grid syncs).  Same
// results, bit for bit, as pbs_fft_wave.hip / pbs_generic.hip / the oracle: the butterfly dataflow of
// DESIGN.md §4 is kept, only its grouping changes.
//
// CDNA4 mapping (one CU runs the whole PBS, no inter-CU synchronisation):
//   * 512 threads: waves 0-3 own polynomial 0 (mask), waves 4-7 polynomial 1 (body); a thread holds 4 of
//     the 1024 complex points and 8 of the 2048 accumulator words of its polynomial, all in VGPRs.
//   * a transform is 5 radix-4 passes (two §4 stages each) with an in-place LDS exchange + one
//     s_barrier between passes; the thread -> positions map of every pass is fixed, so all 31 twiddles
//     a thread ever needs are loop invariants held in registers (no twiddle loads in the loop).
//   * forward pass P works on position bits (9-2P, 8-2P):  pos = hi<<(10-2P) | a<<(8-2P) | lo,
//     inverse pass Q on bits (2Q+1, 2Q):                    pos = hi<<(2Q+2)  | a<<(2Q)   | lo,
//     a = register index, (hi, lo) = the thread's 8 bits.  Forward pass 4 and inverse pass 0 share the
//     map pos = 4t + a (no exchange around the multiply-accumulate), inverse pass 4 and forward pass 0
//     share pos = 256a + t = the accumulator's coefficient map (no exchange around the accumulation).
//   * only the first forward and the last inverse exchange cross waves: 4 s_barrier per CMUX iteration
//     (those two, the exchange of the two transforms, the restaged accumulator); the other six exchanges
//     stay inside a wave.  The accumulator is staged in its own LDS buffer for the rotation.
#include "kernels.h"

namespace tfhe_hip {
namespace blockk {

constexpr int N = 2048, n = 1024, LOG2N2 = 12, TPB = 512;
// work / exchange buffers hold 1024 points in up to 1280 slots.  Every exchange is written completely in one
// thread->position map (runs of 4^k consecutive positions per lane group) and read completely in the next, so
// each exchange picks the slot layout that keeps BOTH of its ds_*_b128 patterns bank-conflict free:
//   runs >= 16 on both sides : natural order
//   runs 16 <-> runs 4       : pos + 4*(pos >> 4)
//   runs 4  <-> runs 1       : pos + (pos >> 2)      (also the exchange of the two transforms, runs 1 <-> 1)
constexpr int BUF_SLOTS = n + n / 4;
// A wave's 256 points (position bits 9,8 = wave for every wave-local map) live in its own 320-slot region, so
// the six wave-local exchanges never touch another wave's slots and need no workgroup barrier.
HX_DEV int lay(int runs_log4, int pos) {  // runs_log4 = log4 of the SHORTER run length of the exchange
  const int q = pos & 255, base = (pos >> 8) * 320;
  return base + (runs_log4 >= 2 ? q : runs_log4 == 1 ? q + 4 * (q >> 4) : q + (q >> 2));
}
HX_DEV int padded(int pos) { return pos + (pos >> 2); }
constexpr size_t STAGE_BYTES = (size_t)2 * N * 8;             // staged accumulator, both polynomials
constexpr size_t WORK_BYTES = (size_t)2 * BUF_SLOTS * 16;     // transform exchanges
constexpr size_t XCHG_BYTES = (size_t)2 * BUF_SLOTS * 16;     // forward results for the other polynomial's waves
constexpr size_t SMEM_BYTES = STAGE_BYTES + WORK_BYTES + XCHG_BYTES + 64;
// multi-bit products: the inverse transform's exchanges have their own buffer, so that a wave may start the next
// group's forward exchanges while another still reads the last inverse one (one workgroup barrier less per group)
constexpr size_t SMEM_MB_BYTES = SMEM_BYTES + WORK_BYTES;

HX_DEV cplx ldc(const double *t, int idx) { return cplx{t[2 * idx], t[2 * idx + 1]}; }

HX_DEV int fwd_pos(int P, int t, int a) {
  const int s = 8 - 2 * P;
  return ((t >> s) << (s + 2)) | (a << s) | (t & ((1 << s) - 1));
}
HX_DEV int inv_pos(int Q, int t, int a) {
  const int s = 2 * Q;
  return ((t >> s) << (s + 2)) | (a << s) | (t & ((1 << s) - 1));
}

// the two §4 stages of a forward pass on the thread's 4 points
HX_DEV void fwd_pass(cplx (&d)[4], const cplx w0, const cplx w1a, const cplx w1b) {
  bfly(d[0], d[2], w0);
  bfly(d[1], d[3], w0);
  bfly(d[0], d[1], w1a);
  bfly(d[2], d[3], w1b);
}
HX_DEV void inv_pass(cplx (&d)[4], const cplx w0, const cplx w1a, const cplx w1b) {
  bfly(d[0], d[1], w0);
  bfly(d[2], d[3], w0);
  bfly(d[0], d[2], w1a);
  bfly(d[1], d[3], w1b);
}
// inverse stages half = 1, 2 (plain additions; j = 1 of half = 2 multiplies by -i)
HX_DEV void inv_pass0(cplx (&o)[4]) {
  HX_UNROLL
  for (int a = 0; a < 4; a += 2) {
    const cplx x = o[a], y = o[a + 1];
    o[a] = cplx{x.re + y.re, x.im + y.im};
    o[a + 1] = cplx{x.re - y.re, x.im - y.im};
  }
  {
    const cplx x = o[0], y = o[2];
    o[0] = cplx{x.re + y.re, x.im + y.im};
    o[2] = cplx{x.re - y.re, x.im - y.im};
  }
  {
    const cplx x = o[1], y = o[3];
    o[1] = cplx{x.re + y.im, x.im - y.re};
    o[3] = cplx{x.re - y.im, x.im + y.re};
  }
}

// 4 x 4 transpose between the register index and lane bits (5, 4): new d[a'] in the lane whose bits (5,4) are x
// = old d[x] in the lane whose bits (5,4) are a'.  This IS the exchange between the passes that work on position
// bits (7,6) and (5,4) (forward 1 -> 2, inverse 2 -> 3), so it needs no LDS.
HX_DEV void swap_lane54(cplx (&d)[4]) {
  uint32_t w[4][4];
  HX_UNROLL
  for (int r = 0; r < 4; ++r) __builtin_memcpy(w[r], &d[r], 16);
  HX_UNROLL
  for (int k = 0; k < 4; ++k) {
    hx_permlane32_swap(w[0][k], w[2][k]);
    hx_permlane32_swap(w[1][k], w[3][k]);
  }
  HX_UNROLL
  for (int k = 0; k < 4; ++k) {
    hx_permlane16_swap(w[0][k], w[1][k]);
    hx_permlane16_swap(w[2][k], w[3][k]);
  }
  HX_UNROLL
  for (int r = 0; r < 4; ++r) __builtin_memcpy(&d[r], w[r], 16);
}

// MB (multi-bit latency path, multibit.hip): the "key" of group gl is the keybundle parked by mb_keybundle_kernel
// in transform-position order, the product takes the accumulator itself (no rotation) and OVERWRITES it
// (cc/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:647-880); groups come in passes of at most `gcount`,
// the accumulator crossing passes in `acc_g`.
struct MbLatArgs {
  const cplx *kb = nullptr;   // [sample][gcount][level][row][col][n]
  uint64_t *acc_g = nullptr;  // [sample][2][N]
  uint32_t gcount = 0, gpass = 0;
  int first = 1, last = 1;
  int slots = 0;  // keybundles parked in the key's slot order (multibit.hip: from 17 ciphertexts on) or in position order
};

// MB_SLOTS (multi-bit products): the parked keybundles are in the key's slot order instead of position order
template <int LEVEL_CT, int BASE_LOG_CT, bool MB = false, bool MB_SLOTS = false>
__global__ void __launch_bounds__(TPB) pbs_fft_block_kernel(PbsArgs a, FftTables tb, MbLatArgs mb) {
  HX_DYN_SMEM(smem);
  const int tid = threadIdx.x;
  const int w = tid >> 8;   // polynomial of the GLWE this thread works on
  const int t = tid & 255;
  uint64_t *stage = (uint64_t *)smem + (size_t)w * N;                       // my polynomial, staged
  cplx *work = (cplx *)(smem + STAGE_BYTES) + (size_t)w * BUF_SLOTS;
  cplx *work_inv = MB ? (cplx *)(smem + SMEM_BYTES) + (size_t)w * BUF_SLOTS : work;
  cplx *xmy = (cplx *)(smem + STAGE_BYTES + WORK_BYTES) + (size_t)w * BUF_SLOTS;
  const cplx *xot = (const cplx *)(smem + STAGE_BYTES + WORK_BYTES) + (size_t)(w ^ 1) * BUF_SLOTS;
  uint64_t *red = (uint64_t *)(smem + STAGE_BYTES);  // reduction scratch before the loop

  const uint32_t level = LEVEL_CT ? (uint32_t)LEVEL_CT : a.level;
  const uint32_t base_log = BASE_LOG_CT ? (uint32_t)BASE_LOG_CT : a.base_log;
  const uint32_t sample = blockIdx.x;
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * 2 * N + (size_t)w * N;
  const cplx *bsk = (const cplx *)a.bsk;

  // ---- loop-invariant twiddles of this thread
  cplx fw[5][3], iw[5][3], un[4];
  HX_UNROLL
  for (int P = 0; P < 5; ++P) {
    const int hi = t >> (8 - 2 * P);
    fw[P][0] = ldc(tb.fwd, (1 << (2 * P)) + hi);
    fw[P][1] = ldc(tb.fwd, (2 << (2 * P)) + 2 * hi);
    fw[P][2] = ldc(tb.fwd, (2 << (2 * P)) + 2 * hi + 1);
  }
  HX_UNROLL
  for (int Q = 1; Q < 5; ++Q) {
    const int lo = t & ((1 << (2 * Q)) - 1);
    iw[Q][0] = ldc(tb.inv, (1 << (2 * Q)) + lo);
    iw[Q][1] = ldc(tb.inv, (2 << (2 * Q)) + lo);
    iw[Q][2] = ldc(tb.inv, (2 << (2 * Q)) + (1 << (2 * Q)) + lo);
  }
  HX_UNROLL
  for (int r = 0; r < 4; ++r) un[r] = ldc(tb.untw, r * 256 + t);

  // ---- body modulus switch (with the centered-mean correction)
  uint64_t corr = 0;
  if (a.ms_type == 1) {
    uint64_t sh = 0;
    int64_t sd = 0;
    for (uint32_t i = tid; i < a.n; i += TPB) {
      uint64_t h;
      int64_t dd;
      centered_ms_terms(lwe[i], LOG2N2, h, dd);
      sh += h;
      sd += dd;
    }
    red[tid] = sh;
    red[TPB + tid] = (uint64_t)sd;
    __syncthreads();
    uint64_t th = 0, td = 0;
    for (int l = 0; l < TPB; ++l) {
      th += red[l];
      td += red[TPB + l];
    }
    __syncthreads();
    corr = centered_ms_finish(th, (int64_t)td, LOG2N2);
  }
  const uint32_t b_hat = (uint32_t)modulus_switch(lwe[a.n] + corr, LOG2N2);

  // ---- accumulator registers: coefficients (r*256 + t) and (1024 + r*256 + t), held NEGATED (as the
  // throughput kernel does): the rotate-and-subtract is then xor / one 64-bit add / xor
  constexpr bool NEGACC = !MB;  // the multi-bit product needs no rotation: the accumulator keeps its sign
  uint64_t acc_re[4], acc_im[4];
  if (MB && !mb.first) {  // a later pass over the groups: the accumulator of the previous pass
    const uint64_t *mine = mb.acc_g + (size_t)sample * 2 * N + (size_t)w * N;
    HX_UNROLL
    for (int r = 0; r < 4; ++r) {
      acc_re[r] = mine[r * 256 + t];
      acc_im[r] = mine[1024 + r * 256 + t];
    }
  } else {
    HX_UNROLL
    for (int r = 0; r < 4; ++r) {
      bool neg;
      uint32_t src = monomial_div_src(r * 256 + t, b_hat, N, neg);
      uint64_t v = lut[src];
      acc_re[r] = (neg != NEGACC) ? (uint64_t)0 - v : v;
      src = monomial_div_src(1024 + r * 256 + t, b_hat, N, neg);
      v = lut[src];
      acc_im[r] = (neg != NEGACC) ? (uint64_t)0 - v : v;
    }
  }
  const TorusConsts kt = torus_consts();
  if constexpr (!MB) {
    HX_UNROLL
    for (int r = 0; r < 4; ++r) {
      stage[r * 256 + t] = acc_re[r];
      stage[1024 + r * 256 + t] = acc_im[r];
    }
  }
  __syncthreads();

  uint64_t mask_next = lwe[0];
  const uint32_t trips = MB ? mb.gpass : a.n;
  for (uint32_t i = 0; i < trips; ++i) {
    uint32_t a_hat = 1;
    if constexpr (!MB) {
      const uint64_t mask_cur = mask_next;  // requested one iteration ago (lwe has n + 1 words)
      mask_next = lwe[i + 1];
      a_hat = (uint32_t)modulus_switch(mask_cur, LOG2N2);
      if (a_hat == 0) continue;  // uniform over the workgroup (bootstrap.rs:334)
    }
    // with A = -acc in the registers and in `stage`, S = A[(c - rr) mod N]:  ct1[c] = ((A[c] ^ M) + S) ^ M,
    // M = all-ones where the source is not negated (no wrap, a_hat < N; both flipped together), else zero
    // (bit 31 of the byte offset u, which the LDS address ignores, carries the a_hat < N flag: M is one shift)
    const uint32_t ub = (uint32_t)(((int32_t)t - (int32_t)(a_hat & (N - 1))) * 8) + ((a_hat & N) ? 0u : 0x80000000u);
    cplx o[4];
    for (uint32_t idx = 0; idx < level; ++idx) {
      // key rows [i][idx][row][c = w] at the storage slots of my 4 positions (pos = 4t + r)
      const cplx *kbase = MB ? mb.kb + ((size_t)sample * mb.gcount + i) * level * 4 * n : bsk + (size_t)i * level * 4 * n;
      const cplx *b0 = kbase + (((size_t)idx * 2 + 0) * 2 + w) * n;
      const cplx *b1 = kbase + (((size_t)idx * 2 + 1) * 2 + w) * n;
      cplx k0[4], k1[4];
      HX_UNROLL
      for (int r = 0; r < 4; ++r) {
        // parked keybundles: position order (one 64-byte run per thread and row) or the key's slot order
        const int slot = (!MB || MB_SLOTS) ? bsk_slot<N, 2>(4 * t + r) : 4 * t + r;
        k0[r] = b0[slot];
        k1[r] = b1[slot];
      }
      // ---- digits of (acc * X^a_hat - acc) at level idx, map pos = 256 r + t
      cplx d[4];
      uint64_t x0[4], x1[4];
      HX_UNROLL
      for (int r = 0; r < 4; ++r) {
        if constexpr (MB) {  // the external product of the accumulator itself
          x0[r] = acc_re[r];
          x1[r] = acc_im[r];
        } else {
          const int32_t u0 = (int32_t)(ub + r * 2048u), u1 = (int32_t)(ub + r * 2048u + 8192u);
          const uint32_t m0 = (uint32_t)(u0 >> 31), m1 = (uint32_t)(u1 >> 31);
          const uint64_t M0 = ((uint64_t)m0 << 32) | m0, M1 = ((uint64_t)m1 << 32) | m1;
          const uint64_t s0 = *(const uint64_t *)((const char *)stage + (u0 & 0x3ff8));
          const uint64_t s1 = *(const uint64_t *)((const char *)stage + (u1 & 0x3ff8));
          x0[r] = ((acc_re[r] ^ M0) + s0) ^ M0;
          x1[r] = ((acc_im[r] ^ M1) + s1) ^ M1;
        }
      }
      if constexpr (LEVEL_CT == 1 && BASE_LOG_CT != 0 && BASE_LOG_CT <= 30) {
        // two-instruction rounding; it can differ from the decomposer only where it yields -B/2, and a
        // thread that saw that value redoes its digits with the decomposer's own bit sequence
        int32_t q[8], lowest = 0;
        HX_UNROLL
        for (int r = 0; r < 4; ++r) {
          q[2 * r] = decomp_digit_l1_fast((uint32_t)(x0[r] >> 32), BASE_LOG_CT);
          q[2 * r + 1] = decomp_digit_l1_fast((uint32_t)(x1[r] >> 32), BASE_LOG_CT);
          lowest = q[2 * r] < lowest ? q[2 * r] : lowest;
          lowest = q[2 * r + 1] < lowest ? q[2 * r + 1] : lowest;
        }
        if (lowest == -(1 << (BASE_LOG_CT - 1))) {
          HX_UNROLL
          for (int r = 0; r < 4; ++r) {
            q[2 * r] = decomp_digit_l1_hi((uint32_t)(x0[r] >> 32), BASE_LOG_CT);
            q[2 * r + 1] = decomp_digit_l1_hi((uint32_t)(x1[r] >> 32), BASE_LOG_CT);
          }
        }
        HX_UNROLL
        for (int r = 0; r < 4; ++r) d[r] = cplx{(double)q[2 * r], (double)q[2 * r + 1]};
      } else {
        HX_UNROLL
        for (int r = 0; r < 4; ++r)
          d[r] = cplx{i64_to_f64(decomp_digit(x0[r], base_log, level, idx)),
                      i64_to_f64(decomp_digit(x1[r], base_log, level, idx))};
      }
      // ---- forward transform: 5 passes, exchanges in `work` (padded slots)
      HX_UNROLL
      for (int P = 0; P < 5; ++P) {
        if (P > 0 && P != 2) {
          HX_UNROLL
          for (int r = 0; r < 4; ++r) d[r] = work[lay(4 - P, fwd_pos(P, t, r))];
          HX_WAVE_SYNC();  // the next exchange reuses these slots in another layout
        }
        fwd_pass(d, fw[P][0], fw[P][1], fw[P][2]);
        if (P == 1) {
          swap_lane54(d);  // pass 2 regroups lane bits (5,4): two permlane swaps per dword, no LDS
        } else if (P < 4) {
          HX_UNROLL
          for (int r = 0; r < 4; ++r) work[lay(3 - P, fwd_pos(P, t, r))] = d[r];
          // pass P+1 regroups threads whose index differs in bits (7-2P, 6-2P): other waves only for P = 0
          if (P == 0) HX_BLOCK_SYNC_LDS();
          else HX_WAVE_SYNC();
        }
      }
      // ---- publish, fetch the other polynomial's points, multiply-accumulate (row 0 then row 1)
      HX_UNROLL
      for (int r = 0; r < 4; ++r) xmy[padded(4 * t + r)] = d[r];
      HX_BLOCK_SYNC_LDS();
      HX_UNROLL
      for (int r = 0; r < 4; ++r) {
        const cplx x = xot[padded(4 * t + r)];
        const cplx f0 = w == 0 ? d[r] : x, f1 = w == 0 ? x : d[r];
        const cplx tt = (idx == 0) ? cmul_first(f0, k0[r]) : cmul_add(f0, k0[r], o[r]);
        o[r] = cmul_add(f1, k1[r], tt);
      }
      // xmy is next written after at least four more barriers (next level's forward passes or the inverse)
    }
    // ---- inverse transform: pass 0 in place, then 4 exchanges
    inv_pass0(o);
    HX_UNROLL
    for (int Q = 1; Q < 5; ++Q) {
      if (Q == 3) {
        swap_lane54(o);  // lane bits (5,4) <-> register index, as in the forward transform
      } else {
        HX_UNROLL
        for (int r = 0; r < 4; ++r) work_inv[lay(Q - 1, inv_pos(Q - 1, t, r))] = o[r];
        // pass Q regroups threads whose index differs in bits (2Q-1, 2Q-2): other waves only for Q = 4
        if (Q == 4) HX_BLOCK_SYNC_LDS();
        else HX_WAVE_SYNC();
        HX_UNROLL
        for (int r = 0; r < 4; ++r) o[r] = work_inv[lay(Q - 1, inv_pos(Q, t, r))];
        HX_WAVE_SYNC();  // the next exchange reuses these slots in another layout
      }
      inv_pass(o, iw[Q][0], iw[Q][1], iw[Q][2]);
    }
    // ---- untwist, back to the torus, accumulate, restage (fft/mod.rs:311-330); map pos = 256 r + t
    if constexpr (MB) {
      HX_UNROLL
      for (int r = 0; r < 4; ++r) {  // dst = 0 + src (x) keybundle: the result replaces the accumulator
        acc_re[r] = from_torus(fma(-o[r].im, un[r].im, o[r].re * un[r].re));
        acc_im[r] = from_torus(fma(o[r].im, un[r].re, o[r].re * un[r].im));
      }
      // no barrier here: the next group's forward exchanges use `work` (every wave is past this group's reads of
      // it: two workgroup barriers lie in between), its publication of the transform comes behind the barrier of the
      // first forward exchange, and the inverse exchanges live in `work_inv`
    } else {
      HX_UNROLL
      for (int r = 0; r < 4; ++r) {
        // minus the untwisted value: from_torus is odd, the sign rides on the multiplication
        const double tr = fma(o[r].im, un[r].im, -o[r].re * un[r].re);
        const double ti = fma(-o[r].im, un[r].re, -o[r].re * un[r].im);
        from_torus_add(acc_re[r], tr, kt);
        from_torus_add(acc_im[r], ti, kt);
      }
      // the rotated reads of this iteration are many barriers behind: restage for the next one
      HX_UNROLL
      for (int r = 0; r < 4; ++r) {
        stage[r * 256 + t] = acc_re[r];
        stage[1024 + r * 256 + t] = acc_im[r];
      }
      HX_BLOCK_SYNC_LDS();
    }
  }
  if (MB && !mb.last) {  // more groups to come in another pass
    uint64_t *mine = mb.acc_g + (size_t)sample * 2 * N + (size_t)w * N;
    HX_UNROLL
    for (int r = 0; r < 4; ++r) {
      mine[r * 256 + t] = acc_re[r];
      mine[1024 + r * 256 + t] = acc_im[r];
    }
    return;
  }

  // ---- sample extraction (cc/algorithms/glwe_sample_extraction.rs:119-146); many-LUT outputs
  const size_t out_sz = (size_t)N + 1;
  for (uint32_t m = 0; m < a.num_many_lut; ++m) {
    const uint32_t nth = m * a.lut_stride;
    uint64_t *out = a.lwe_out + (size_t)m * a.num_samples * out_sz + (size_t)a.out_idx[sample] * out_sz;
    if (w == 0) {
      HX_UNROLL
      for (int r = 0; r < 4; ++r) {
        uint32_t c = r * 256 + t;
        out[c <= nth ? nth - c : N + nth - c] = ((c <= nth) != NEGACC) ? acc_re[r] : (uint64_t)0 - acc_re[r];
        c += 1024;
        out[c <= nth ? nth - c : N + nth - c] = ((c <= nth) != NEGACC) ? acc_im[r] : (uint64_t)0 - acc_im[r];
      }
    } else {
      HX_UNROLL
      for (int r = 0; r < 4; ++r) {
        if ((uint32_t)(r * 256 + t) == nth) out[N] = NEGACC ? (uint64_t)0 - acc_re[r] : acc_re[r];
        if ((uint32_t)(1024 + r * 256 + t) == nth) out[N] = NEGACC ? (uint64_t)0 - acc_im[r] : acc_im[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Dual-stream variant: 256 threads, every thread carries its 4 points of BOTH polynomials.  The two
// transforms are independent instruction streams inside one wave (the LDS round trip of one is covered by
// the butterflies of the other), the twiddles are shared, and the multiply-accumulate needs no exchange at
// all: a thread already holds both transforms at its positions.  3 s_barrier per CMUX iteration.
constexpr int TPB2 = 256;
constexpr size_t SMEM2_BYTES = STAGE_BYTES + WORK_BYTES + 64;

template <int LEVEL_CT, int BASE_LOG_CT>
__global__ void __launch_bounds__(TPB2) pbs_fft_block2_kernel(PbsArgs a, FftTables tb) {
  HX_DYN_SMEM(smem);
  const int t = threadIdx.x;
  uint64_t *stage = (uint64_t *)smem;                   // [polynomial][N]
  cplx *work = (cplx *)(smem + STAGE_BYTES);            // [polynomial][BUF_SLOTS]
  uint64_t *red = (uint64_t *)(smem + STAGE_BYTES);     // reduction scratch before the loop

  const uint32_t level = LEVEL_CT ? (uint32_t)LEVEL_CT : a.level;
  const uint32_t base_log = BASE_LOG_CT ? (uint32_t)BASE_LOG_CT : a.base_log;
  const uint32_t sample = blockIdx.x;
  const uint64_t *lwe = a.lwe_in + (size_t)a.in_idx[sample] * (a.n + 1);
  const uint64_t *lut = a.lut + (size_t)a.lut_idx[sample] * 2 * N;
  const cplx *bsk = (const cplx *)a.bsk;

  cplx fw[5][3], iw[5][3], un[4];
  HX_UNROLL
  for (int P = 0; P < 5; ++P) {
    const int hi = t >> (8 - 2 * P);
    fw[P][0] = ldc(tb.fwd, (1 << (2 * P)) + hi);
    fw[P][1] = ldc(tb.fwd, (2 << (2 * P)) + 2 * hi);
    fw[P][2] = ldc(tb.fwd, (2 << (2 * P)) + 2 * hi + 1);
  }
  HX_UNROLL
  for (int Q = 1; Q < 5; ++Q) {
    const int lo = t & ((1 << (2 * Q)) - 1);
    iw[Q][0] = ldc(tb.inv, (1 << (2 * Q)) + lo);
    iw[Q][1] = ldc(tb.inv, (2 << (2 * Q)) + lo);
    iw[Q][2] = ldc(tb.inv, (2 << (2 * Q)) + (1 << (2 * Q)) + lo);
  }
  HX_UNROLL
  for (int r = 0; r < 4; ++r) un[r] = ldc(tb.untw, r * 256 + t);

  uint64_t corr = 0;
  if (a.ms_type == 1) {
    uint64_t sh = 0;
    int64_t sd = 0;
    for (uint32_t i = t; i < a.n; i += TPB2) {
      uint64_t h;
      int64_t dd;
      centered_ms_terms(lwe[i], LOG2N2, h, dd);
      sh += h;
      sd += dd;
    }
    red[t] = sh;
    red[TPB2 + t] = (uint64_t)sd;
    __syncthreads();
    uint64_t th = 0, td = 0;
    for (int l = 0; l < TPB2; ++l) {
      th += red[l];
      td += red[TPB2 + l];
    }
    __syncthreads();
    corr = centered_ms_finish(th, (int64_t)td, LOG2N2);
  }
  const uint32_t b_hat = (uint32_t)modulus_switch(lwe[a.n] + corr, LOG2N2);

  uint64_t acc_re[2][4], acc_im[2][4];
  HX_UNROLL
  for (int w = 0; w < 2; ++w)
    HX_UNROLL
    for (int r = 0; r < 4; ++r) {
      bool neg;
      uint32_t src = monomial_div_src(r * 256 + t, b_hat, N, neg);
      uint64_t v = lut[w * N + src];
      acc_re[w][r] = neg ? (uint64_t)0 - v : v;
      src = monomial_div_src(1024 + r * 256 + t, b_hat, N, neg);
      v = lut[w * N + src];
      acc_im[w][r] = neg ? (uint64_t)0 - v : v;
      stage[w * N + r * 256 + t] = acc_re[w][r];
      stage[w * N + 1024 + r * 256 + t] = acc_im[w][r];
    }
  __syncthreads();

  uint64_t mask_next = lwe[0];
  for (uint32_t i = 0; i < a.n; ++i) {
    const uint64_t mask_cur = mask_next;  // requested one iteration ago (lwe has n + 1 words)
    mask_next = lwe[i + 1];
    const uint32_t a_hat = (uint32_t)modulus_switch(mask_cur, LOG2N2);
    if (a_hat == 0) continue;  // uniform over the workgroup (bootstrap.rs:334)
    const uint32_t rr = a_hat & (N - 1);
    const bool odd = (a_hat & N) != 0;
    cplx o[2][4];
    for (uint32_t idx = 0; idx < level; ++idx) {
      // key rows [i][idx][row][col] at the storage slots of my 4 positions (pos = 4t + r)
      cplx key[2][2][4];
      HX_UNROLL
      for (int row = 0; row < 2; ++row)
        HX_UNROLL
        for (int col = 0; col < 2; ++col) {
          const cplx *b = bsk + ((((size_t)i * level + idx) * 2 + row) * 2 + col) * n;
          HX_UNROLL
          for (int r = 0; r < 4; ++r) key[row][col][r] = b[bsk_slot<N, 2>(4 * t + r)];
        }
      // ---- digits of (acc * X^a_hat - acc) at level idx, map pos = 256 r + t, both polynomials
      cplx d[2][4];
      HX_UNROLL
      for (int w = 0; w < 2; ++w)
        HX_UNROLL
        for (int r = 0; r < 4; ++r) {
          const uint32_t c0 = r * 256 + t, c1 = 1024 + r * 256 + t;
          uint64_t s = stage[w * N + ((c0 - rr) & (N - 1))];
          const uint64_t x0 = (((c0 < rr) != odd) ? (uint64_t)0 - s : s) - acc_re[w][r];
          s = stage[w * N + ((c1 - rr) & (N - 1))];
          const uint64_t x1 = (((c1 < rr) != odd) ? (uint64_t)0 - s : s) - acc_im[w][r];
          if constexpr (LEVEL_CT == 1 && BASE_LOG_CT != 0 && BASE_LOG_CT <= 30) {
            d[w][r] = cplx{(double)decomp_digit_l1_hi((uint32_t)(x0 >> 32), BASE_LOG_CT),
                           (double)decomp_digit_l1_hi((uint32_t)(x1 >> 32), BASE_LOG_CT)};
          } else {
            d[w][r] = cplx{i64_to_f64(decomp_digit(x0, base_log, level, idx)),
                           i64_to_f64(decomp_digit(x1, base_log, level, idx))};
          }
        }
      // ---- forward transforms: 5 passes, the two polynomials interleaved
      HX_UNROLL
      for (int P = 0; P < 5; ++P) {
        if (P > 0) {
          HX_UNROLL
          for (int w = 0; w < 2; ++w)
            HX_UNROLL
            for (int r = 0; r < 4; ++r) d[w][r] = work[w * BUF_SLOTS + lay(4 - P, fwd_pos(P, t, r))];
          HX_WAVE_SYNC();  // the next exchange reuses these slots in another layout
        }
        HX_UNROLL
        for (int w = 0; w < 2; ++w) {
          fwd_pass(d[w], fw[P][0], fw[P][1], fw[P][2]);
          if (P < 4) {
            HX_UNROLL
            for (int r = 0; r < 4; ++r) work[w * BUF_SLOTS + lay(3 - P, fwd_pos(P, t, r))] = d[w][r];
          }
        }
        if (P < 4) {
          // pass P+1 regroups threads whose index differs in bits (7-2P, 6-2P): other waves only for P = 0
          if (P == 0) HX_BLOCK_SYNC_LDS();
          else HX_WAVE_SYNC();
        }
      }
      // ---- multiply-accumulate: both transforms are here (cc/fft_impl/fft64/crypto/ggsw.rs:616-697 order)
      HX_UNROLL
      for (int col = 0; col < 2; ++col)
        HX_UNROLL
        for (int r = 0; r < 4; ++r) {
          const cplx tt = (idx == 0) ? cmul_first(d[0][r], key[0][col][r]) : cmul_add(d[0][r], key[0][col][r], o[col][r]);
          o[col][r] = cmul_add(d[1][r], key[1][col][r], tt);
        }
    }
    // ---- inverse transforms: pass 0 in place, then 4 exchanges
    HX_UNROLL
    for (int w = 0; w < 2; ++w) inv_pass0(o[w]);
    HX_UNROLL
    for (int Q = 1; Q < 5; ++Q) {
      HX_UNROLL
      for (int w = 0; w < 2; ++w)
        HX_UNROLL
        for (int r = 0; r < 4; ++r) work[w * BUF_SLOTS + lay(Q - 1, inv_pos(Q - 1, t, r))] = o[w][r];
      // pass Q regroups threads whose index differs in bits (2Q-1, 2Q-2): other waves only for Q = 4
      if (Q == 4) HX_BLOCK_SYNC_LDS();
      else HX_WAVE_SYNC();
      HX_UNROLL
      for (int w = 0; w < 2; ++w)
        HX_UNROLL
        for (int r = 0; r < 4; ++r) o[w][r] = work[w * BUF_SLOTS + lay(Q - 1, inv_pos(Q, t, r))];
      HX_WAVE_SYNC();  // the next exchange reuses these slots in another layout
      HX_UNROLL
      for (int w = 0; w < 2; ++w) inv_pass(o[w], iw[Q][0], iw[Q][1], iw[Q][2]);
    }
    // ---- untwist, back to the torus, accumulate, restage (fft/mod.rs:311-330); map pos = 256 r + t
    HX_UNROLL
    for (int w = 0; w < 2; ++w)
      HX_UNROLL
      for (int r = 0; r < 4; ++r) {
        const double tr = fma(-o[w][r].im, un[r].im, o[w][r].re * un[r].re);
        const double ti = fma(o[w][r].im, un[r].re, o[w][r].re * un[r].im);
        acc_re[w][r] += from_torus(tr);
        acc_im[w][r] += from_torus(ti);
        stage[w * N + r * 256 + t] = acc_re[w][r];
        stage[w * N + 1024 + r * 256 + t] = acc_im[w][r];
      }
    HX_BLOCK_SYNC_LDS();
  }

  // ---- sample extraction (cc/algorithms/glwe_sample_extraction.rs:119-146); many-LUT outputs
  const size_t out_sz = (size_t)N + 1;
  for (uint32_t m = 0; m < a.num_many_lut; ++m) {
    const uint32_t nth = m * a.lut_stride;
    uint64_t *out = a.lwe_out + (size_t)m * a.num_samples * out_sz + (size_t)a.out_idx[sample] * out_sz;
    HX_UNROLL
    for (int r = 0; r < 4; ++r) {
      uint32_t c = r * 256 + t;
      out[c <= nth ? nth - c : N + nth - c] = c <= nth ? acc_re[0][r] : (uint64_t)0 - acc_re[0][r];
      if (c == nth) out[N] = acc_re[1][r];
      c += 1024;
      out[c <= nth ? nth - c : N + nth - c] = c <= nth ? acc_im[0][r] : (uint64_t)0 - acc_im[0][r];
      if (c == nth) out[N] = acc_im[1][r];
    }
  }
}

}  // namespace blockk

bool pbs_fft_block_supported(uint32_t N, uint32_t glwe_dim, uint32_t level) {
  return N == 2048 && glwe_dim == 1 && level >= 1 && level <= 8;
}

template <int L, int B>
static void launch_block_t(hipStream_t st, const PbsArgs &a, const FftTables &tb) {
  using namespace blockk;
  hx_set_dynamic_smem_once<pbs_fft_block_kernel<L, B>>(SMEM_BYTES);
  HX_LAUNCH((pbs_fft_block_kernel<L, B>), dim3(a.num_samples), dim3(TPB), SMEM_BYTES, st, a, tb, MbLatArgs{});
}

template <int L, int B>
static void launch_block_mb_t(hipStream_t st, const PbsArgs &a, const FftTables &tb, const blockk::MbLatArgs &mb) {
  using namespace blockk;
  if (mb.slots) {
    hx_set_dynamic_smem_once<pbs_fft_block_kernel<L, B, true, true>>(SMEM_MB_BYTES);
    HX_LAUNCH((pbs_fft_block_kernel<L, B, true, true>), dim3(a.num_samples), dim3(TPB), SMEM_MB_BYTES, st, a, tb, mb);
  } else {
    hx_set_dynamic_smem_once<pbs_fft_block_kernel<L, B, true>>(SMEM_MB_BYTES);
    HX_LAUNCH((pbs_fft_block_kernel<L, B, true>), dim3(a.num_samples), dim3(TPB), SMEM_MB_BYTES, st, a, tb, mb);
  }
}
// products of the multi-bit latency path (multibit.hip) on the latency kernel: N = 2048, k = 1
void launch_mb_accumulate_block(hipStream_t st, const PbsArgs &a, const FftTables &tb, const cplx *kb_lat,
                                uint64_t *acc_g, uint32_t gcount, uint32_t gpass, int first, int last, int slots) {
  blockk::MbLatArgs mb;
  mb.slots = slots;
  mb.kb = kb_lat;
  mb.acc_g = acc_g;
  mb.gcount = gcount;
  mb.gpass = gpass;
  mb.first = first;
  mb.last = last;
  if (a.level == 1 && a.base_log == 22) launch_block_mb_t<1, 22>(st, a, tb, mb);       // the GPU group-4 sets
  else if (a.level == 2 && a.base_log == 15) launch_block_mb_t<2, 15>(st, a, tb, mb);  // group-3 set
  else launch_block_mb_t<0, 0>(st, a, tb, mb);
}

template <int L, int B>
static void launch_block2_t(hipStream_t st, const PbsArgs &a, const FftTables &tb) {
  using namespace blockk;
  hx_set_dynamic_smem_once<pbs_fft_block2_kernel<L, B>>(SMEM2_BYTES);
  HX_LAUNCH((pbs_fft_block2_kernel<L, B>), dim3(a.num_samples), dim3(TPB2), SMEM2_BYTES, st, a, tb);
}

// variant 0: dual-stream (256 threads), 1: one polynomial per half workgroup (512 threads)
void launch_pbs_fft_block(hipStream_t st, const PbsArgs &a, const FftTables &tb, int variant) {
  if (variant == 0) {
    if (a.level == 1 && a.base_log == 23) launch_block2_t<1, 23>(st, a, tb);  // PARAM_MESSAGE_2_CARRY_2
    else launch_block2_t<0, 0>(st, a, tb);
  } else {
    if (a.level == 1 && a.base_log == 23) launch_block_t<1, 23>(st, a, tb);
    else launch_block_t<0, 0>(st, a, tb);
  }
}

}  // namespace tfhe_hip
