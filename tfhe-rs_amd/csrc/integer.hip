// integer.hip — radix-integer layer on top of the KS -> PBS path ("next" row N1 of SURVEY §8):
// a batched apply-LUT round driver and, built from it, carry propagation, addition and
// schoolbook multiplication of radix ciphertexts.
//
// Replaces (host orchestration + three small kernels; all heavy work is the keyswitch and PBS
// kernels of this library):
//   backends/tfhe-cuda-backend/cuda/src/integer/integer.cuh:869-990   apply_univariate_lookup_table
//   .../integer/integer.cuh:1266-1305                                 LUT generation
//   .../integer/integer.cuh (propagate_single_carry), addition.cuh, multiplication.cuh
//   semantics: tfhe/src/integer/server_key/radix_parallel/{add.rs,mul.rs}, shortint bivariate_pbs.rs
//
// MI355X-first differences from the reference's shape:
//   * a CudaRadixCiphertextFFI may hold a BATCH of independent integers ([ciphertext][block], the
//     scratch's num_blocks = blocks per integer); every round is then ONE keyswitch launch and ONE
//     PBS launch over all blocks of all integers (thousands of PBS per launch is where the PBS
//     kernel is efficient), instead of one stream per integer.
//   * rounds are described by device index arrays (gather / scatter / LUT index per block), so no
//     ciphertext is ever moved to be "aligned" for a round.
#include "kernels.h"
#include "../../include/tfhe_hip_backend.h"

#include <algorithm>
#include <functional>
#include <vector>

namespace tfhe_hip {
namespace radix {

// ------------------------------------------------------------------ device kernels
// out[oi] = a[ai] * scalar + b[bi]   (b optional), `words` u64 per LWE; null index = identity
__global__ void __launch_bounds__(256) lwe_axpy_kernel(uint64_t *out, const uint64_t *out_idx, const uint64_t *a,
                                                       const uint64_t *a_idx, uint64_t scalar, const uint64_t *b,
                                                       const uint64_t *b_idx, uint32_t words, uint32_t count) {
  const uint32_t s = blockIdx.x;
  if (s >= count) return;
  const size_t oi = out_idx ? out_idx[s] : s, ai = a_idx ? a_idx[s] : s;
  const uint64_t *pa = a + ai * words;
  uint64_t *po = out + oi * words;
  if (b) {
    const uint64_t *pb = b + (b_idx ? b_idx[s] : s) * words;
    for (uint32_t j = threadIdx.x; j < words; j += blockDim.x) po[j] = pa[j] * scalar + pb[j];
  } else {
    for (uint32_t j = threadIdx.x; j < words; j += blockDim.x) po[j] = pa[j] * scalar;
  }
}
// out[g] = sum of pool[members[m]] for m in [offsets[g], offsets[g+1])  (CSR groups)
__global__ void __launch_bounds__(256) lwe_group_sum_kernel(uint64_t *out, const uint64_t *pool,
                                                            const uint64_t *offsets, const uint64_t *members,
                                                            uint32_t words, uint32_t groups) {
  const uint32_t g = blockIdx.x;
  if (g >= groups) return;
  const uint64_t lo = offsets[g], hi = offsets[g + 1];
  for (uint32_t j = threadIdx.x; j < words; j += blockDim.x) {
    uint64_t acc = 0;
    for (uint64_t m = lo; m < hi; ++m) acc += pool[members[m] * words + j];
    out[(size_t)g * words + j] = acc;
  }
}

static void axpy(hipStream_t st, uint64_t *out, const uint64_t *out_idx, const uint64_t *a, const uint64_t *a_idx,
                 uint64_t scalar, const uint64_t *b, const uint64_t *b_idx, uint32_t words, uint32_t count) {
  if (count == 0) return;
  HX_LAUNCH(lwe_axpy_kernel, dim3(count), dim3(256), 0, st, out, out_idx, a, a_idx, scalar, b, b_idx, words, count);
}

// ------------------------------------------------------------------ host helpers
struct Params {
  uint32_t big_n, small_n, k, N, pbs_base_log, pbs_level, ks_base_log, ks_level, msg, carry, ms_type;
};

static Params make_params(CudaLweBootstrapKeyParamsFFI b, CudaLweKeyswitchKeyParamsFFI k, uint32_t msg, uint32_t carry,
                          uint32_t ms_type) {
  HX_PANIC_IF_FALSE(b.pbs_type == 1 /* CLASSICAL */, "radix layer: only the classic PBS is wired (pbs_type=%u)",
                    b.pbs_type);
  HX_PANIC_IF_FALSE(b.glwe_dimension * b.polynomial_size == k.input_lwe_dimension &&
                        k.output_lwe_dimension == b.input_lwe_dimension,
                    "radix layer: keyswitch and bootstrap key dimensions do not chain");
  HX_PANIC_IF_FALSE(msg >= 2 && carry >= msg && (b.polynomial_size % (msg * carry)) == 0,
                    "radix layer: unsupported message/carry moduli (%u, %u)", msg, carry);
  return Params{k.input_lwe_dimension, b.input_lwe_dimension, b.glwe_dimension, b.polynomial_size, b.base_log,
                b.level_count,         k.base_log,            k.level_count,    msg,               carry,
                ms_type};
}

// cuda/src/integer/integer.cuh:1266-1305 (generate_lookup_table_with_encoding, same in/out encoding)
static void generate_lut(const Params &p, uint64_t *acc, const std::function<uint64_t(uint64_t)> &f) {
  const uint32_t sup = p.msg * p.carry, box = p.N / sup;
  const uint64_t delta = ((uint64_t)1 << 63) / sup;
  std::fill(acc, acc + (size_t)p.k * p.N, 0);
  uint64_t *body = acc + (size_t)p.k * p.N;
  for (uint32_t i = 0; i < sup; ++i)
    for (uint32_t j = i * box; j < (i + 1) * box; ++j) body[j] = f(i) * delta;
  const uint32_t half = box / 2;
  for (uint32_t i = 0; i < half; ++i) body[i] = (uint64_t)0 - body[i];
  std::rotate(body, body + half, body + p.N);
}

template <class T>
static T *dev_upload(hipStream_t st, const std::vector<T> &h) {
  T *d = nullptr;
  if (h.empty()) return d;
  HX_CHECK(hipMalloc((void **)&d, h.size() * sizeof(T)));
  HX_CHECK(hipMemcpyAsync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
  HX_CHECK(hipStreamSynchronize(st));  // h may be a temporary
  return d;
}

// The KS -> PBS round driver (integer.cuh:869-990 for one GPU): `count` blocks gathered from `in`
// through in_idx are keyswitched, bootstrapped with LUT lut_idx[s] and scattered to out[out_idx[s]].
struct LutDriver {
  static constexpr uint32_t kMagic = 0x52445231;  // "RDR1"
  uint32_t magic = kMagic;
  Params p{};
  uint32_t gpu = 0, cap = 0, num_luts = 0;
  uint64_t *d_ks = nullptr, *d_luts = nullptr, *d_trivial = nullptr;  // d_trivial = 0, 1, ..., cap - 1
  int8_t *pbs_buf = nullptr;

  void init(hipStream_t st, uint32_t gpu_index, const Params &params, uint32_t capacity,
            const std::vector<std::vector<uint64_t>> &luts) {
    p = params;
    gpu = gpu_index;
    cap = capacity;
    num_luts = (uint32_t)luts.size();
    const size_t lw = (size_t)(p.k + 1) * p.N;
    HX_CHECK(hipMalloc((void **)&d_ks, (size_t)cap * (p.small_n + 1) * sizeof(uint64_t)));
    HX_CHECK(hipMalloc((void **)&d_luts, std::max<size_t>(1, num_luts) * lw * sizeof(uint64_t)));
    for (uint32_t i = 0; i < num_luts; ++i)
      HX_CHECK(hipMemcpyAsync(d_luts + i * lw, luts[i].data(), lw * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    std::vector<uint64_t> triv(cap);
    for (uint32_t i = 0; i < cap; ++i) triv[i] = i;
    d_trivial = dev_upload(st, triv);
    HX_CHECK(hipStreamSynchronize(st));
    scratch_cuda_programmable_bootstrap_64_async(st, gpu, &pbs_buf, p.small_n, p.k, p.N, p.pbs_level, cap, true,
                                                 (enum PBS_MS_REDUCTION_T)p.ms_type);
  }
  // one round, split into launches of at most `cap` blocks; a null in_idx / out_idx means "block s"
  void round(hipStream_t st, uint64_t *out, const uint64_t *out_idx, const uint64_t *in, const uint64_t *in_idx,
             const uint64_t *lut_idx, uint32_t count, const void *ksk, const void *bsk) const {
    const size_t w = (size_t)p.big_n + 1;
    for (uint32_t off = 0; off < count; off += cap) {
      const uint32_t c = std::min(cap, count - off);
      cuda_keyswitch_lwe_ciphertext_vector_64_64_async(st, gpu, d_ks, d_trivial, in_idx ? in : in + off * w,
                                                       in_idx ? in_idx + off : d_trivial, ksk, p.big_n, p.small_n,
                                                       p.ks_base_log, p.ks_level, c);
      cuda_programmable_bootstrap_64_async(st, gpu, out_idx ? out : out + off * w,
                                           out_idx ? out_idx + off : d_trivial, d_luts, lut_idx + off, d_ks,
                                           d_trivial, bsk, pbs_buf, p.small_n, p.k, p.N, p.pbs_base_log, p.pbs_level,
                                           c, 1, 0);
    }
  }
  void release(hipStream_t st) {
    HX_CHECK(hipStreamSynchronize(st));
    if (pbs_buf) cleanup_cuda_programmable_bootstrap_64(st, gpu, &pbs_buf);
    if (d_ks) HX_CHECK(hipFree(d_ks));
    if (d_luts) HX_CHECK(hipFree(d_luts));
    if (d_trivial) HX_CHECK(hipFree(d_trivial));
    d_ks = d_luts = d_trivial = nullptr;
    magic = 0;
  }
};

static hipStream_t S0(const CudaStreamsFFI &s) {
  HX_PANIC_IF_FALSE(s.gpu_count >= 1 && s.streams != nullptr, "radix layer: empty stream set");
  return (hipStream_t)s.streams[0];
}
static uint32_t G0(const CudaStreamsFFI &s) { return s.gpu_indexes ? s.gpu_indexes[0] : 0; }

// ------------------------------------------------------------------ apply a univariate LUT
struct ApplyLutMem {
  static constexpr uint32_t kMagic = 0x4C555431;  // "LUT1"
  uint32_t magic = kMagic;
  LutDriver drv;
  uint64_t *d_lut_idx = nullptr;  // all zero
  uint64_t degree = 0;
};

// ------------------------------------------------------------------ carry propagation
// States as in radix_parallel/add.rs (OutputCarry): 0 none, 1 generated, 2 propagated.
enum : uint64_t { LUT_STATE_FIRST = 0, LUT_STATE = 1, LUT_MSG = 2, LUT_SCAN = 3, LUT_FINAL = 4, LUT_CARRY = 5 };

struct PropagateMem {
  static constexpr uint32_t kMagic = 0x50524F50;  // "PROP"
  uint32_t magic = kMagic;
  LutDriver drv;
  uint32_t blocks = 0;      // blocks per integer
  uint32_t max_cts = 0;     // integers the scratch was sized for
  // scratch ciphertexts: X = [states | messages] (2T), P = packed inputs (T)
  uint64_t *d_x = nullptr, *d_p = nullptr;
  // cached index arrays for a batch size
  uint32_t cached_cts = 0;
  std::vector<uint64_t *> dev_arrays;
  struct Round {
    uint64_t *a_idx, *b_idx, *o_idx, *lut_idx;
    uint32_t count;
  };
  uint64_t *r1_in = nullptr, *r1_lut = nullptr;
  std::vector<Round> scan;
  Round fin{};
  uint64_t *first_src = nullptr, *first_dst = nullptr;
  uint32_t first_count = 0;

  void build_indexes(hipStream_t st, uint32_t cts) {
    for (auto *d : dev_arrays) HX_CHECK(hipFree(d));
    dev_arrays.clear();
    scan.clear();
    const uint32_t L = blocks, T = cts * L;
    auto up = [&](const std::vector<uint64_t> &h) {
      uint64_t *d = dev_upload(st, h);
      if (d) dev_arrays.push_back(d);
      return d;
    };
    std::vector<uint64_t> in(2 * T), lut(2 * T);
    for (uint32_t t = 0; t < T; ++t) {
      in[t] = in[T + t] = t;
      lut[t] = (t % L == 0) ? LUT_STATE_FIRST : LUT_STATE;
      lut[T + t] = LUT_MSG;
    }
    r1_in = up(in);
    r1_lut = up(lut);
    for (uint32_t d = 1; d < L; d <<= 1) {  // Hillis-Steele inclusive scan of the carry states
      std::vector<uint64_t> a, b, o, l;
      for (uint32_t t = 0; t < T; ++t)
        if (t % L >= d) {
          a.push_back(t - d);
          b.push_back(t);
          o.push_back(t);
          l.push_back(LUT_SCAN);
        }
      scan.push_back(Round{up(a), up(b), up(o), up(l), (uint32_t)a.size()});
    }
    std::vector<uint64_t> a, b, o, l, fs, fd;
    for (uint32_t t = 0; t < T; ++t) {
      if (t % L == 0) {
        fs.push_back(T + t);  // message of block 0 is final
        fd.push_back(t);
      } else {
        a.push_back(t - 1);   // state entering block t
        b.push_back(T + t);   // its message
        o.push_back(t);
        l.push_back(LUT_FINAL);
      }
    }
    fin = Round{up(a), up(b), up(o), up(l), (uint32_t)a.size()};
    first_src = up(fs);
    first_dst = up(fd);
    first_count = (uint32_t)fs.size();
    cached_cts = cts;
  }

  void init(hipStream_t st, uint32_t gpu, const Params &p, uint32_t num_blocks, uint32_t cts) {
    blocks = num_blocks;
    max_cts = cts;
    const uint64_t m = p.msg;
    std::vector<std::function<uint64_t(uint64_t)>> fs = {
        [m](uint64_t x) -> uint64_t { return x >= m ? 1 : 0; },                         // first block: no propagate
        [m](uint64_t x) -> uint64_t { return x >= m ? 1 : (x == m - 1 ? 2 : 0); },      // state
        [m](uint64_t x) -> uint64_t { return x % m; },                                  // message
        [m](uint64_t x) -> uint64_t { return (x % m) == 2 ? (x / m) : (x % m); },       // scan: cur==prop ? prev : cur
        [m](uint64_t x) -> uint64_t { return ((x % m) + ((x / m) == 1 ? 1 : 0)) % m; }, // message + incoming carry
        [m](uint64_t x) -> uint64_t { return x / m; },                                  // carry of a block
    };
    std::vector<std::vector<uint64_t>> luts;
    for (auto &f : fs) {
      luts.emplace_back((size_t)(p.k + 1) * p.N);
      generate_lut(p, luts.back().data(), f);
    }
    const uint32_t T = cts * num_blocks;
    drv.init(st, gpu, p, std::min<uint32_t>(2 * T, 1u << 16), luts);
    const size_t w = p.big_n + 1;
    HX_CHECK(hipMalloc((void **)&d_x, (size_t)2 * T * w * sizeof(uint64_t)));
    HX_CHECK(hipMalloc((void **)&d_p, (size_t)T * w * sizeof(uint64_t)));
  }

  // in place on `blocks_ptr` (cts integers of `blocks` blocks, every block value < 2*msg)
  void run(hipStream_t st, uint64_t *blocks_ptr, uint32_t cts, const void *ksk, const void *bsk) {
    HX_PANIC_IF_FALSE(cts >= 1 && cts <= max_cts, "carry propagation: %u integers exceed the scratch capacity %u", cts,
                      max_cts);
    if (cached_cts != cts) build_indexes(st, cts);
    const Params &p = drv.p;
    const uint32_t w = p.big_n + 1, T = cts * blocks;
    // round 1: state and message of every block (two LUTs on the same inputs, one launch)
    drv.round(st, d_x, nullptr, blocks_ptr, r1_in, r1_lut, 2 * T, ksk, bsk);
    // prefix scan of the states
    for (const Round &r : scan) {
      axpy(st, d_p, nullptr, d_x, r.a_idx, p.msg, d_x, r.b_idx, w, r.count);
      drv.round(st, d_x, r.o_idx, d_p, nullptr, r.lut_idx, r.count, ksk, bsk);
    }
    // message + carry entering each block
    axpy(st, d_p, nullptr, d_x, fin.a_idx, p.msg, d_x, fin.b_idx, w, fin.count);
    drv.round(st, blocks_ptr, fin.o_idx, d_p, nullptr, fin.lut_idx, fin.count, ksk, bsk);
    axpy(st, blocks_ptr, first_dst, d_x, first_src, 1, nullptr, nullptr, w, first_count);
  }

  void release(hipStream_t st) {
    drv.release(st);
    for (auto *d : dev_arrays) HX_CHECK(hipFree(d));
    dev_arrays.clear();
    if (d_x) HX_CHECK(hipFree(d_x));
    if (d_p) HX_CHECK(hipFree(d_p));
    magic = 0;
  }
};

// ------------------------------------------------------------------ multiplication
// radix_parallel/mul.rs: block products (low / high halves through bivariate LUTs), column sums in
// groups that fit the carry space, final carry propagation.
struct MulMem {
  static constexpr uint32_t kMagic = 0x4D554C31;  // "MUL1"
  uint32_t magic = kMagic;
  LutDriver drv;       // LUTs: 0 product low, 1 product high, 2 message, 3 carry
  PropagateMem prop;
  uint32_t blocks = 0, max_cts = 0, sub = 0;  // sub = integers per pass
  uint32_t slots = 0;                          // pool slots per integer
  uint64_t *d_pool = nullptr, *d_pack = nullptr, *d_sum = nullptr;

  struct Step {  // one reduction step, ciphertext-relative
    std::vector<uint64_t> offsets, members;      // CSR of groups (pool slots)
    std::vector<uint64_t> msg_slot, carry_slot;  // output slots per group (carry_slot = ~0 when dropped)
  };
  std::vector<uint64_t> prod_lhs, prod_rhs, prod_slot, prod_lut;  // products
  std::vector<Step> steps;
  std::vector<std::vector<uint64_t>> final_cols;  // <= 2 slots per column

  void plan() {
    const uint32_t L = blocks, chunk = (drv.p.msg * drv.p.carry - 1) / (drv.p.msg - 1);
    std::vector<std::vector<uint64_t>> cols(L);
    uint64_t next = 0;
    for (uint32_t i = 0; i < L; ++i)
      for (uint32_t j = 0; i + j < L; ++j) {
        prod_lhs.push_back(j);
        prod_rhs.push_back(i);
        prod_lut.push_back(0);
        prod_slot.push_back(next);
        cols[i + j].push_back(next++);
        if (i + j + 1 < L) {
          prod_lhs.push_back(j);
          prod_rhs.push_back(i);
          prod_lut.push_back(1);
          prod_slot.push_back(next);
          cols[i + j + 1].push_back(next++);
        }
      }
    auto max_len = [&]() {
      size_t m = 0;
      for (auto &c : cols) m = std::max(m, c.size());
      return m;
    };
    while (max_len() > 2) {
      Step s;
      s.offsets.push_back(0);
      std::vector<std::vector<uint64_t>> nc(L);
      for (uint32_t c = 0; c < L; ++c) {
        size_t pos = 0;
        const size_t n = cols[c].size();
        while (pos < n) {
          const size_t len = std::min<size_t>(chunk, n - pos);
          if (len == 1) {  // nothing to add: the term stays as it is
            nc[c].push_back(cols[c][pos]);
          } else {
            for (size_t m = 0; m < len; ++m) s.members.push_back(cols[c][pos + m]);
            s.offsets.push_back(s.members.size());
            s.msg_slot.push_back(next);
            nc[c].push_back(next++);
            if (c + 1 < L) {
              s.carry_slot.push_back(next);
              nc[c + 1].push_back(next++);
            } else {
              s.carry_slot.push_back(~(uint64_t)0);
            }
          }
          pos += len;
        }
      }
      cols.swap(nc);
      steps.push_back(std::move(s));
    }
    final_cols = cols;
    slots = (uint32_t)next;
  }

  void init(hipStream_t st, uint32_t gpu, const Params &p, uint32_t num_blocks, uint32_t cts) {
    blocks = num_blocks;
    max_cts = cts;
    const uint64_t m = p.msg;
    std::vector<std::function<uint64_t(uint64_t)>> fs = {
        [m](uint64_t x) -> uint64_t { return ((x / m) * (x % m)) % m; },
        [m](uint64_t x) -> uint64_t { return ((x / m) * (x % m)) / m; },
        [m](uint64_t x) -> uint64_t { return x % m; },
        [m](uint64_t x) -> uint64_t { return x / m; },
    };
    std::vector<std::vector<uint64_t>> luts;
    for (auto &f : fs) {
      luts.emplace_back((size_t)(p.k + 1) * p.N);
      generate_lut(p, luts.back().data(), f);
    }
    drv.p = p;
    plan();
    // integers per pass: the term pool of one pass may take 12 GiB of the 288 GB (32-block integers: 37 MB
    // each, ~340 per pass) — the later reduction rounds shrink fast, so passes should be as wide as possible
    const size_t w = p.big_n + 1;
    const size_t per_ct = (size_t)slots * w * sizeof(uint64_t);
    sub = (uint32_t)std::max<size_t>(1, std::min<size_t>(cts, ((size_t)12 << 30) / per_ct));
    drv.init(st, gpu, p, 1u << 16, luts);
    HX_CHECK(hipMalloc((void **)&d_pool, (size_t)sub * per_ct));
    const size_t n_prod = prod_slot.size();
    size_t max_groups = 0;
    for (auto &s : steps) max_groups = std::max(max_groups, s.msg_slot.size());
    HX_CHECK(hipMalloc((void **)&d_pack, (size_t)sub * n_prod * w * sizeof(uint64_t)));
    HX_CHECK(hipMalloc((void **)&d_sum, std::max<size_t>(1, (size_t)sub * max_groups) * w * sizeof(uint64_t)));
    prop.init(st, gpu, p, num_blocks, sub);
  }

  // lhs <- lhs * rhs for `cts` integers
  void run(hipStream_t st, uint64_t *lhs, const uint64_t *rhs, uint32_t cts, const void *ksk, const void *bsk) {
    HX_PANIC_IF_FALSE(cts >= 1 && cts <= max_cts, "multiplication: %u integers exceed the scratch capacity %u", cts,
                      max_cts);
    const Params &p = drv.p;
    const uint32_t L = blocks, w = p.big_n + 1;
    const size_t n_prod = prod_slot.size();
    for (uint32_t c0 = 0; c0 < cts; c0 += sub) {
      const uint32_t nb = std::min(sub, cts - c0);
      uint64_t *l0 = lhs + (size_t)c0 * L * w;
      const uint64_t *r0 = rhs + (size_t)c0 * L * w;
      std::vector<uint64_t *> tmp;
      auto up = [&](const std::vector<uint64_t> &h) {
        uint64_t *d = dev_upload(st, h);
        if (d) tmp.push_back(d);
        return d;
      };
      {  // block products
        std::vector<uint64_t> a(nb * n_prod), b(nb * n_prod), o(nb * n_prod), l(nb * n_prod);
        for (uint32_t c = 0; c < nb; ++c)
          for (size_t q = 0; q < n_prod; ++q) {
            a[c * n_prod + q] = (uint64_t)c * L + prod_lhs[q];
            b[c * n_prod + q] = (uint64_t)c * L + prod_rhs[q];
            o[c * n_prod + q] = (uint64_t)c * slots + prod_slot[q];
            l[c * n_prod + q] = prod_lut[q];
          }
        uint64_t *da = up(a), *db = up(b), *dout = up(o), *dl = up(l);
        axpy(st, d_pack, nullptr, l0, da, p.msg, r0, db, w, (uint32_t)(nb * n_prod));
        drv.round(st, d_pool, dout, d_pack, nullptr, dl, (uint32_t)(nb * n_prod), ksk, bsk);
      }
      for (const Step &s : steps) {  // column sums
        const size_t G = s.msg_slot.size();
        if (G == 0) continue;
        std::vector<uint64_t> off(nb * G + 1), mem(nb * s.members.size());
        std::vector<uint64_t> in, out, lut;
        for (uint32_t c = 0; c < nb; ++c) {
          for (size_t g = 0; g < G; ++g) {
            off[c * G + g] = c * s.members.size() + s.offsets[g];
            in.push_back(c * G + g);
            out.push_back((uint64_t)c * slots + s.msg_slot[g]);
            lut.push_back(2);
            if (s.carry_slot[g] != ~(uint64_t)0) {
              in.push_back(c * G + g);
              out.push_back((uint64_t)c * slots + s.carry_slot[g]);
              lut.push_back(3);
            }
          }
          for (size_t m = 0; m < s.members.size(); ++m) mem[c * s.members.size() + m] = (uint64_t)c * slots + s.members[m];
        }
        off[nb * G] = nb * s.members.size();
        uint64_t *doff = up(off), *dmem = up(mem);
        HX_LAUNCH(lwe_group_sum_kernel, dim3((unsigned)(nb * G)), dim3(256), 0, st, d_sum, d_pool, doff, dmem, w,
                  (uint32_t)(nb * G));
        drv.round(st, d_pool, up(out), d_sum, up(in), up(lut), (uint32_t)in.size(), ksk, bsk);
      }
      {  // at most two terms per column: add them into lhs, then propagate the carries
        std::vector<uint64_t> a, b, o, a1, o1;
        for (uint32_t c = 0; c < nb; ++c)
          for (uint32_t col = 0; col < L; ++col) {
            const auto &v = final_cols[col];
            if (v.size() == 2) {
              a.push_back((uint64_t)c * slots + v[0]);
              b.push_back((uint64_t)c * slots + v[1]);
              o.push_back((uint64_t)c * L + col);
            } else {
              a1.push_back((uint64_t)c * slots + v[0]);
              o1.push_back((uint64_t)c * L + col);
            }
          }
        axpy(st, l0, up(o), d_pool, up(a), 1, d_pool, up(b), w, (uint32_t)a.size());
        axpy(st, l0, up(o1), d_pool, up(a1), 1, nullptr, nullptr, w, (uint32_t)a1.size());
        prop.run(st, l0, nb, ksk, bsk);
      }
      HX_CHECK(hipStreamSynchronize(st));
      for (auto *d : tmp) HX_CHECK(hipFree(d));
    }
  }

  void release(hipStream_t st) {
    drv.release(st);
    prop.release(st);
    if (d_pool) HX_CHECK(hipFree(d_pool));
    if (d_pack) HX_CHECK(hipFree(d_pack));
    if (d_sum) HX_CHECK(hipFree(d_sum));
    magic = 0;
  }
};

static uint32_t batch_of(const CudaRadixCiphertextFFI *ct, uint32_t blocks, const char *what) {
  HX_PANIC_IF_FALSE(ct != nullptr && ct->ptr != nullptr, "%s: null radix ciphertext", what);
  HX_PANIC_IF_FALSE(blocks != 0 && ct->num_radix_blocks % blocks == 0,
                    "%s: %u blocks is not a whole number of %u-block integers", what, ct->num_radix_blocks, blocks);
  return ct->num_radix_blocks / blocks;
}

}  // namespace radix
}  // namespace tfhe_hip

using namespace tfhe_hip;
using namespace tfhe_hip::radix;

extern "C" {

// ---- cuda/include/integer/integer.h:127-148 ------------------------------------------------
uint64_t scratch_cuda_apply_univariate_lut_64_async(CudaStreamsFFI streams, int8_t **mem_ptr, void const *input_lut,
                                                    CudaLweBootstrapKeyParamsFFI bsk_params,
                                                    CudaLweKeyswitchKeyParamsFFI ksk_params,
                                                    uint32_t input_lwe_ciphertext_count, uint32_t message_modulus,
                                                    uint32_t carry_modulus, uint64_t lut_degree,
                                                    bool allocate_gpu_memory,
                                                    enum PBS_MS_REDUCTION_T noise_reduction_type) {
  HX_PANIC_IF_FALSE(allocate_gpu_memory, "apply_univariate_lut: size-only scratch is not supported");
  HX_PANIC_IF_FALSE(input_lut != nullptr && mem_ptr != nullptr, "apply_univariate_lut: null pointer");
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, (uint32_t)noise_reduction_type);
  auto *m = new ApplyLutMem();
  const size_t lw = (size_t)(p.k + 1) * p.N;
  std::vector<std::vector<uint64_t>> luts(1);
  luts[0].assign((const uint64_t *)input_lut, (const uint64_t *)input_lut + lw);
  m->drv.init(S0(streams), G0(streams), p, std::max<uint32_t>(1, input_lwe_ciphertext_count), luts);
  m->degree = lut_degree;
  HX_CHECK(hipMalloc((void **)&m->d_lut_idx, std::max<uint32_t>(1, input_lwe_ciphertext_count) * sizeof(uint64_t)));
  HX_CHECK(hipMemsetAsync(m->d_lut_idx, 0, std::max<uint32_t>(1, input_lwe_ciphertext_count) * sizeof(uint64_t),
                          S0(streams)));
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return 0;
}

void cuda_apply_univariate_lut_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *output_radix_lwe,
                                        CudaRadixCiphertextFFI const *input_radix_lwe, int8_t *mem_ptr,
                                        void *const *ksks, void *const *bsks) {
  auto *m = reinterpret_cast<ApplyLutMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == ApplyLutMem::kMagic, "apply_univariate_lut: foreign scratch pointer");
  HX_PANIC_IF_FALSE(output_radix_lwe && input_radix_lwe && ksks && bsks, "apply_univariate_lut: null pointer");
  HX_PANIC_IF_FALSE(output_radix_lwe->lwe_dimension == input_radix_lwe->lwe_dimension,
                    "input and output radix ciphertexts should have the same lwe dimension");
  const uint32_t n = input_radix_lwe->num_radix_blocks;
  HX_PANIC_IF_FALSE(n <= m->drv.cap && n <= output_radix_lwe->num_radix_blocks,
                    "num radix blocks on which lut is applied should be smaller or equal to the number of lut radix "
                    "blocks");
  m->drv.round(S0(streams), (uint64_t *)output_radix_lwe->ptr, nullptr, (const uint64_t *)input_radix_lwe->ptr,
               nullptr, m->d_lut_idx, n, ksks[0], bsks[0]);
  if (output_radix_lwe->degrees)
    for (uint32_t i = 0; i < n; ++i) output_radix_lwe->degrees[i] = m->degree;
  if (output_radix_lwe->noise_levels)
    for (uint32_t i = 0; i < n; ++i) output_radix_lwe->noise_levels[i] = 1;
}

void cleanup_cuda_apply_univariate_lut_64(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  auto *m = reinterpret_cast<ApplyLutMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == ApplyLutMem::kMagic, "cleanup apply_univariate_lut: foreign scratch pointer");
  m->drv.release(S0(streams));
  HX_CHECK(hipFree(m->d_lut_idx));
  m->magic = 0;
  delete m;
  *mem_ptr_void = nullptr;
}

// ---- cuda/include/linear_algebra.h:26-28 -----------------------------------------------------
void cuda_add_lwe_ciphertext_vector_inplace_64(void *stream, uint32_t gpu_index,
                                               CudaRadixCiphertextFFI *lwe_array_inout,
                                               CudaRadixCiphertextFFI const *input_2) {
  (void)gpu_index;
  HX_PANIC_IF_FALSE(lwe_array_inout && input_2 && lwe_array_inout->num_radix_blocks == input_2->num_radix_blocks &&
                        lwe_array_inout->lwe_dimension == input_2->lwe_dimension,
                    "add: operands must have the same shape");
  axpy((hipStream_t)stream, (uint64_t *)lwe_array_inout->ptr, nullptr, (const uint64_t *)lwe_array_inout->ptr, nullptr,
       1, (const uint64_t *)input_2->ptr, nullptr, lwe_array_inout->lwe_dimension + 1,
       lwe_array_inout->num_radix_blocks);
  for (uint32_t i = 0; i < lwe_array_inout->num_radix_blocks; ++i) {
    if (lwe_array_inout->degrees && input_2->degrees) lwe_array_inout->degrees[i] += input_2->degrees[i];
    if (lwe_array_inout->noise_levels && input_2->noise_levels)
      lwe_array_inout->noise_levels[i] += input_2->noise_levels[i];
  }
}

// ---- cuda/include/integer/integer.h:383-413 --------------------------------------------------
// num_blocks = blocks per integer; the ciphertexts handed to the launch may hold any whole number
// of integers up to the capacity given through hip_integer_scratch_batch (default 1).
static uint32_t g_scratch_batch = 1;
void hip_integer_scratch_batch(uint32_t num_integers) { g_scratch_batch = num_integers ? num_integers : 1; }

uint64_t scratch_cuda_propagate_single_carry_64_inplace_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                              CudaLweBootstrapKeyParamsFFI bsk_params,
                                                              CudaLweKeyswitchKeyParamsFFI ksk_params,
                                                              uint32_t num_blocks, uint32_t message_modulus,
                                                              uint32_t carry_modulus, uint32_t requested_flag,
                                                              bool allocate_gpu_memory,
                                                              enum PBS_MS_REDUCTION_T noise_reduction_type) {
  HX_PANIC_IF_FALSE(allocate_gpu_memory, "propagate_single_carry: size-only scratch is not supported");
  HX_PANIC_IF_FALSE(requested_flag == 0, "propagate_single_carry: overflow / carry flags are not wired");
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, (uint32_t)noise_reduction_type);
  auto *m = new PropagateMem();
  m->init(S0(streams), G0(streams), p, num_blocks, g_scratch_batch);
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return 0;
}
uint64_t scratch_cuda_add_and_propagate_single_carry_64_inplace_async(
    CudaStreamsFFI streams, int8_t **mem_ptr, CudaLweBootstrapKeyParamsFFI bsk_params,
    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks, uint32_t message_modulus, uint32_t carry_modulus,
    uint32_t requested_flag, bool allocate_gpu_memory, enum PBS_MS_REDUCTION_T noise_reduction_type) {
  return scratch_cuda_propagate_single_carry_64_inplace_async(streams, mem_ptr, bsk_params, ksk_params, num_blocks,
                                                              message_modulus, carry_modulus, requested_flag,
                                                              allocate_gpu_memory, noise_reduction_type);
}

void cuda_propagate_single_carry_64_inplace_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lwe_array,
                                                  CudaRadixCiphertextFFI *carry_out,
                                                  const CudaRadixCiphertextFFI *carry_in, int8_t *mem_ptr,
                                                  void *const *bsks, void *const *ksks, uint32_t requested_flag,
                                                  uint32_t uses_carry) {
  (void)carry_out;
  auto *m = reinterpret_cast<PropagateMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == PropagateMem::kMagic, "propagate_single_carry: foreign scratch pointer");
  HX_PANIC_IF_FALSE(requested_flag == 0 && uses_carry == 0 && carry_in == nullptr,
                    "propagate_single_carry: input carry / flags are not wired");
  const uint32_t cts = batch_of(lwe_array, m->blocks, "propagate_single_carry");
  m->run(S0(streams), (uint64_t *)lwe_array->ptr, cts, ksks[0], bsks[0]);
  for (uint32_t i = 0; i < lwe_array->num_radix_blocks; ++i) {
    if (lwe_array->degrees) lwe_array->degrees[i] = m->drv.p.msg - 1;
    if (lwe_array->noise_levels) lwe_array->noise_levels[i] = 1;
  }
}

void cuda_add_and_propagate_single_carry_64_inplace_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *lhs_array,
                                                          const CudaRadixCiphertextFFI *rhs_array,
                                                          CudaRadixCiphertextFFI *carry_out,
                                                          const CudaRadixCiphertextFFI *carry_in, int8_t *mem_ptr,
                                                          void *const *bsks, void *const *ksks,
                                                          uint32_t requested_flag, uint32_t uses_carry) {
  cuda_add_lwe_ciphertext_vector_inplace_64(streams.streams[0], G0(streams), lhs_array, rhs_array);
  cuda_propagate_single_carry_64_inplace_async(streams, lhs_array, carry_out, carry_in, mem_ptr, bsks, ksks,
                                               requested_flag, uses_carry);
}

void cleanup_cuda_propagate_single_carry_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  auto *m = reinterpret_cast<PropagateMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == PropagateMem::kMagic, "cleanup propagate_single_carry: foreign scratch pointer");
  m->release(S0(streams));
  delete m;
  *mem_ptr_void = nullptr;
}
void cleanup_cuda_add_and_propagate_single_carry_64_inplace(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  cleanup_cuda_propagate_single_carry_64_inplace(streams, mem_ptr_void);
}

// ---- cuda/include/integer/integer.h:173-187 --------------------------------------------------
uint64_t scratch_cuda_integer_mult_inplace_64_async(CudaStreamsFFI streams, int8_t **mem_ptr,
                                                    bool const is_boolean_left, bool const is_boolean_right,
                                                    uint32_t message_modulus, uint32_t carry_modulus,
                                                    CudaLweBootstrapKeyParamsFFI bsk_params,
                                                    CudaLweKeyswitchKeyParamsFFI ksk_params, uint32_t num_blocks,
                                                    bool allocate_gpu_memory,
                                                    enum PBS_MS_REDUCTION_T noise_reduction_type) {
  HX_PANIC_IF_FALSE(allocate_gpu_memory, "integer_mult: size-only scratch is not supported");
  HX_PANIC_IF_FALSE(!is_boolean_left && !is_boolean_right, "integer_mult: boolean operands are not wired");
  const Params p = make_params(bsk_params, ksk_params, message_modulus, carry_modulus, (uint32_t)noise_reduction_type);
  auto *m = new MulMem();
  m->init(S0(streams), G0(streams), p, num_blocks, g_scratch_batch);
  *mem_ptr = reinterpret_cast<int8_t *>(m);
  return 0;
}

void cuda_integer_mult_inplace_64_async(CudaStreamsFFI streams, CudaRadixCiphertextFFI *radix_lwe_inout,
                                        bool const is_bool_left, CudaRadixCiphertextFFI const *radix_lwe_right,
                                        bool const is_bool_right, void *const *bsks, void *const *ksks,
                                        int8_t *mem_ptr, uint32_t polynomial_size, uint32_t num_blocks) {
  auto *m = reinterpret_cast<MulMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == MulMem::kMagic, "integer_mult: foreign scratch pointer");
  HX_PANIC_IF_FALSE(!is_bool_left && !is_bool_right && polynomial_size == m->drv.p.N && num_blocks == m->blocks,
                    "integer_mult: call does not match the scratch");
  const uint32_t cts = batch_of(radix_lwe_inout, m->blocks, "integer_mult");
  HX_PANIC_IF_FALSE(radix_lwe_right && radix_lwe_right->num_radix_blocks == radix_lwe_inout->num_radix_blocks,
                    "integer_mult: operands must have the same shape");
  m->run(S0(streams), (uint64_t *)radix_lwe_inout->ptr, (const uint64_t *)radix_lwe_right->ptr, cts, ksks[0], bsks[0]);
  for (uint32_t i = 0; i < radix_lwe_inout->num_radix_blocks; ++i) {
    if (radix_lwe_inout->degrees) radix_lwe_inout->degrees[i] = m->drv.p.msg - 1;
    if (radix_lwe_inout->noise_levels) radix_lwe_inout->noise_levels[i] = 1;
  }
}

void cleanup_cuda_integer_mult_inplace_64(CudaStreamsFFI streams, int8_t **mem_ptr_void) {
  auto *m = reinterpret_cast<MulMem *>(*mem_ptr_void);
  HX_PANIC_IF_FALSE(m && m->magic == MulMem::kMagic, "cleanup integer_mult: foreign scratch pointer");
  m->release(S0(streams));
  delete m;
  *mem_ptr_void = nullptr;
}

// number of PBS one multiplication / one carry propagation of `num_blocks` blocks issues (for benches)
uint64_t hip_integer_mult_pbs_count(int8_t *mem_ptr) {
  auto *m = reinterpret_cast<MulMem *>(mem_ptr);
  HX_PANIC_IF_FALSE(m && m->magic == MulMem::kMagic, "hip_integer_mult_pbs_count: foreign scratch pointer");
  uint64_t n = m->prod_slot.size();
  for (auto &s : m->steps)
    for (size_t g = 0; g < s.msg_slot.size(); ++g) n += 1 + (s.carry_slot[g] != ~(uint64_t)0);
  const uint32_t L = m->blocks;
  n += 2 * (uint64_t)L;
  for (uint32_t d = 1; d < L; d <<= 1) n += L - d;
  n += L - 1;
  return n;
}

}  // extern "C"
