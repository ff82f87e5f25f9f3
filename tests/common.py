"""Shared fixtures for the parity tests: parameter sets, seeded key material.

Parameter constants restate tfhe-rs' own:
  C1      = PARAM_MESSAGE_2_CARRY_2 (= V1_4_PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128,
            tfhe/src/shortint/parameters/v1_4/classic/tuniform/p_fail_2_minus_128/ks_pbs.rs:28-47)
  C1P     = V1_4_PARAM_MESSAGE_1_CARRY_2_KS_PBS_GAUSSIAN_2M128 shape (N=1024, k=2)
            (.../v1_4/classic/gaussian/p_fail_2_minus_128/ks_pbs.rs:57-80); noise restated as
            TUniform bounds of similar magnitude (our PRNG is not the reference's anyway)
  C4      = V1_1_PARAM_MULTI_BIT_GROUP_3_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128
            (.../v1_1/multi_bit/tuniform/p_fail_2_minus_128/ks_pbs.rs:118-137)
  TOY_*   = small sets in the spirit of core_crypto/algorithms/test/mod.rs:56-200
"""
import dataclasses
import functools

import numpy as np

from . import oracle as orc


@dataclasses.dataclass(frozen=True)
class Params:
    name: str
    n: int              # small LWE dimension
    k: int              # GLWE dimension
    N: int              # polynomial size
    pbs_base_log: int
    pbs_level: int
    ks_base_log: int
    ks_level: int
    lwe_noise: int      # TUniform bound_log2 for the small key
    glwe_noise: int     # TUniform bound_log2 for the GLWE key
    plaintext_modulus: int  # message_modulus * carry_modulus (padding bit on top)
    ms_type: int = 0    # 0 standard, 1 centered mean (PBS_MS_REDUCTION_T)
    grouping: int = 0   # multi-bit grouping factor (0 = classic)

    @property
    def big_n(self):
        return self.k * self.N

    @property
    def delta(self):
        return (1 << 63) // self.plaintext_modulus

    @property
    def log2N2(self):
        return (2 * self.N).bit_length() - 1


C1 = Params("PARAM_MESSAGE_2_CARRY_2", 918, 1, 2048, 23, 1, 4, 4, 45, 17, 16, ms_type=1)
C1P = Params("PARAM_MESSAGE_1_CARRY_2_N1024", 885, 2, 1024, 23, 1, 3, 5, 46, 24, 8, ms_type=0)
C33 = Params("PARAM_MESSAGE_3_CARRY_3_N8192", 1006, 1, 8192, 15, 2, 3, 7, 45, 17, 64, ms_type=0)   # timing only
C4 = Params("PARAM_MULTI_BIT_GROUP_3_MESSAGE_2_CARRY_2", 918, 1, 2048, 15, 2, 3, 6, 45, 17, 16,
            grouping=3)

# the reference's GPU default (v1_1/multi_bit/tuniform/p_fail_2_minus_128/ks_pbs_gpu.rs:205-228); timing only
C4G4 = Params("PARAM_GPU_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2", 920, 1, 2048, 22, 1, 3, 5, 45, 17, 16, grouping=4)
# the reference's GPU g = 3 set of the same precision (same file, :118-137): base_log 14 where the CPU set above has 15
# ... and the gaussian 2^-64 GPU sets with one level (v1_1/multi_bit/gaussian/p_fail_2_minus_64/ks_pbs_gpu.rs); timing only
C4G3_L1 = Params("PARAM_GPU_MULTI_BIT_GROUP_3_MESSAGE_2_CARRY_2_GAUSSIAN_2M64", 813, 1, 2048, 22, 1, 3, 5, 45, 17, 16, grouping=3)
C4G2_L1 = Params("PARAM_GPU_MULTI_BIT_GROUP_2_MESSAGE_2_CARRY_2_GAUSSIAN_2M64", 820, 1, 2048, 22, 1, 3, 5, 45, 17, 16, grouping=2)
C4G3 = Params("PARAM_GPU_MULTI_BIT_GROUP_3_MESSAGE_2_CARRY_2", 879, 1, 2048, 14, 2, 2, 8, 46, 17, 16, grouping=3)

TOY_K1 = Params("toy_k1_N256", 24, 1, 256, 15, 2, 4, 5, 40, 20, 4, ms_type=1)
TOY_K1_L1 = Params("toy_k1_N512_l1", 32, 1, 512, 23, 1, 4, 5, 40, 12, 4, ms_type=0)
TOY_K2 = Params("toy_k2_N256", 20, 2, 256, 12, 3, 3, 6, 40, 20, 4, ms_type=1)
TOY_K3 = Params("toy_k3_N512", 16, 3, 512, 18, 2, 4, 5, 40, 18, 8, ms_type=0)
TOY_2048 = Params("toy_k1_N2048_l1", 12, 1, 2048, 23, 1, 4, 4, 45, 17, 16, ms_type=1)
TOY_2048_P64 = Params("toy_k1_N2048_l1_p64", 12, 1, 2048, 23, 1, 4, 4, 45, 17, 64, ms_type=1)   # 3 message + 3 carry bits
TOY_2048_L2 = Params("toy_k1_N2048_l2", 9, 1, 2048, 15, 2, 3, 6, 45, 17, 16, ms_type=0)
TOY_1024_K2 = Params("toy_k2_N1024_l1", 10, 2, 1024, 23, 1, 3, 5, 46, 24, 8, ms_type=0)
TOY_1024_K1_L2 = Params("toy_k1_N1024_l2", 11, 1, 1024, 15, 2, 3, 5, 46, 20, 8, ms_type=1)
TOY_8192 = Params("toy_k1_N8192_l1", 5, 1, 8192, 23, 1, 4, 5, 45, 17, 16, ms_type=1)    # accumulator in device memory
TOY_16384 = Params("toy_k1_N16384_l2", 4, 1, 16384, 15, 2, 4, 5, 45, 17, 16, ms_type=0)
TOY_MB = Params("toy_multibit_g3", 18, 1, 256, 15, 2, 4, 5, 40, 20, 4, grouping=3)
TOY_MB2 = Params("toy_multibit_g2", 16, 1, 512, 15, 2, 4, 5, 40, 20, 4, grouping=2)
TOY_MB_2048 = Params("toy_multibit_g3_N2048", 9, 1, 2048, 15, 2, 3, 6, 45, 17, 16, grouping=3)   # throughput kernel
TOY_MB_K3 = Params("toy_multibit_g3_k3_N512", 12, 3, 512, 18, 2, 4, 5, 40, 18, 8, grouping=3)
TOY_MB_8192 = Params("toy_multibit_g2_N8192", 4, 1, 8192, 15, 2, 4, 5, 45, 17, 16, grouping=2)   # accumulator in device memory
TOY_MB4_2048 = Params("toy_multibit_g4_N2048_l1", 8, 1, 2048, 22, 1, 3, 6, 45, 17, 16, grouping=4)
TOY_MB3_L1_2048 = Params("toy_multibit_g3_N2048_l1", 9, 1, 2048, 22, 1, 3, 6, 45, 17, 16, grouping=3)   # the gaussian GPU g = 3 / g = 2 sets
TOY_MB2_L1_2048 = Params("toy_multibit_g2_N2048_l1", 8, 1, 2048, 22, 1, 3, 6, 45, 17, 16, grouping=2)   # decomposition (one level)
TOY_MB3G_2048 = Params("toy_multibit_g3_N2048_b14", 9, 1, 2048, 14, 2, 3, 6, 45, 17, 16, grouping=3)   # the GPU g = 3 set's decomposition


@dataclasses.dataclass
class Keys:
    p: Params
    lwe_sk: np.ndarray      # n bits (small key)
    glwe_sk: np.ndarray     # k*N bits (== big LWE key)
    bsk: np.ndarray         # standard-domain BSK (classic or multi-bit layout)
    ksk: np.ndarray         # big -> small


@functools.lru_cache(maxsize=8)
def make_keys(p: Params, seed: int = 0x74666865, with_ksk: bool = True) -> Keys:
    rng = orc.Rng(seed)
    lwe_sk = rng.binary_key(p.n)
    glwe_sk = rng.binary_key(p.k * p.N)
    if p.grouping:
        bsk = orc.gen_multi_bit_bsk(seed + 1, lwe_sk, glwe_sk, p.k, p.N, p.pbs_base_log,
                                    p.pbs_level, p.grouping, p.glwe_noise)
    else:
        bsk = orc.gen_bsk(seed + 1, lwe_sk, glwe_sk, p.k, p.N, p.pbs_base_log, p.pbs_level,
                          p.glwe_noise)
    ksk = (orc.gen_ksk(seed + 2, glwe_sk, lwe_sk, p.ks_base_log, p.ks_level, p.lwe_noise)
           if with_ksk else np.zeros(0, dtype=np.uint64))
    return Keys(p, lwe_sk, glwe_sk, bsk, ksk)


def encrypt_small(p: Params, keys: Keys, msgs, seed=1):
    """Fresh encryptions under the SMALL key (what the PBS consumes)."""
    rng = orc.Rng(seed)
    return np.stack([orc.lwe_encrypt(rng, keys.lwe_sk, (int(m) * p.delta) % (1 << 64), p.lwe_noise)
                     for m in msgs])


def encrypt_big(p: Params, keys: Keys, msgs, seed=2):
    """Fresh encryptions under the BIG key (what KS->PBS consumes)."""
    rng = orc.Rng(seed)
    return np.stack([orc.lwe_encrypt(rng, keys.glwe_sk, (int(m) * p.delta) % (1 << 64), p.glwe_noise)
                     for m in msgs])


def decode(p: Params, raw):
    """round_decode: nearest multiple of delta, modulo 2*plaintext_modulus (padding bit kept)."""
    raw = int(raw)
    return ((raw + p.delta // 2) // p.delta) % (2 * p.plaintext_modulus)


def decrypt_big(p: Params, keys: Keys, ct):
    return decode(p, orc.lwe_decrypt(ct, keys.glwe_sk))


def decrypt_small(p: Params, keys: Keys, ct):
    return decode(p, orc.lwe_decrypt(ct, keys.lwe_sk))


def torus_distance(a, b):
    """max |a-b| on the 2^64 torus, element-wise wrapped to signed."""
    d = (np.asarray(a, dtype=np.uint64) - np.asarray(b, dtype=np.uint64)).astype(np.int64)
    return np.abs(d.astype(np.float64)).max() if d.size else 0.0


# ---------------------------------------------------------------- centered-mean modulus switch, pure Python
def centered_ms_reference(lwe, log_mod):
    """Exact-integer restatement of tfhe/src/core_crypto/algorithms/modulus_switch.rs:57-103 (+ the plain switch of
    fft_impl/common.rs:10-23 for the mask), independent of the C oracle: returns the switched ciphertext."""
    M = (1 << 64) - 1

    def ms(x):
        return ((x + (1 << (63 - log_mod))) & M) >> (64 - log_mod)

    def trunc_half(v):   # Rust's signed `/ 2`: toward zero
        return -((-v) // 2) if v < 0 else v // 2

    H, D = 0, 0
    for a in lwe[:-1]:
        a = int(a)
        e = ((ms(a) << (64 - log_mod)) - a) & M
        e = e - (1 << 64) if e >= (1 << 63) else e
        h = trunc_half(e)
        H = (H + h) & M
        D += 2 * h - e
    corr = (H - trunc_half(D) - (1 << (63 - log_mod))) & M
    return np.array([ms(int(a)) for a in lwe[:-1]] + [ms((int(lwe[-1]) + corr) & M)], dtype=np.uint64), corr


def centered_ms_edge_vectors(n, log_mod, seed=5):
    """Mask patterns on the rounding boundaries of the switch (s = 64 - log_mod): exact ties r 2^s + 2^(s-1) (round
    up, even error), tie +- 1 (odd errors of both signs: every halving truncates, and with n odd the sum of the
    doubled halving errors is odd, so ITS halving truncates too — toward zero for either sign), exact multiples
    (zero error), all-ones / all-zero words, and a mix; bodies on and next to a boundary."""
    rng = np.random.default_rng(seed)
    s = 64 - log_mod
    M = (1 << 64) - 1
    r = [int(v) for v in rng.integers(0, 1 << log_mod, size=n)]
    tie = [((x << s) + (1 << (s - 1))) & M for x in r]
    pats = {
        "ties": tie,
        "tie_minus_1": [(t - 1) & M for t in tie],
        "tie_plus_1": [(t + 1) & M for t in tie],
        "multiples": [(x << s) & M for x in r],
        "all_ones": [M] * n,
        "zeros": [0] * n,
        "mixed": [[tie[i], (tie[i] - 1) & M, (tie[i] + 1) & M, (r[i] << s) & M, M][i % 5] for i in range(n)],
        # errors of alternating sign around the tie, largest magnitudes: the halving errors cancel pairwise (n even) or
        # leave one (n odd)
        "alternating": [((tie[i] - 1) if i % 2 else (tie[i] + 1)) & M for i in range(n)],
        "alternating_multiples": [(((r[i] << s) - 1) if i % 2 else ((r[i] << s) + 1)) & M for i in range(n)],
    }
    bodies = [0, (1 << (s - 1)) - 1, 1 << (s - 1), (1 << 63) + (1 << (s - 1)), M]
    out = {}
    for name, mask in pats.items():
        for j, b in enumerate(bodies):
            out[f"{name}/body{j}"] = np.array(mask + [b], dtype=np.uint64)
    return out


def generate_many_lut(p: Params, functions):
    """shortint's many-LUT accumulator (tfhe/src/shortint/engine/mod.rs:169-254 fill_many_lut_accumulator): the
    plaintext space of p.plaintext_modulus values is shared by len(functions) functions; inputs must stay below
    max_degree + 1 = plaintext_modulus // len(functions); function t occupies the boxes of sub-table t, of
    sample_extraction_stride = (max_degree + 1) * box coefficients.  Returns (accumulator, max_degree, stride)."""
    fn = len(functions)
    sup = p.plaintext_modulus
    assert 1 <= fn <= sup // 2, "Cannot generate many lut accumulator for that many functions"
    box = p.N // sup
    max_degree = sup // fn - 1
    sub = (max_degree + 1) * box
    body = np.zeros(p.N, dtype=np.uint64)
    for t, f in enumerate(functions):
        for m in range(max_degree + 1):
            body[t * sub + m * box:t * sub + (m + 1) * box] = np.uint64((int(f(m)) * p.delta) % (1 << 64))
    half = box // 2
    body[:half] = (np.uint64(0) - body[:half])
    body = np.roll(body, -half)
    return np.concatenate([np.zeros(p.k * p.N, dtype=np.uint64), body]), max_degree, sub
