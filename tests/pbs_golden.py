"""Regenerates the key material and inputs of the reference's GPU golden-value test of the 64-bit bootstrap
(tfhe/src/core_crypto/gpu/algorithms/test/pbs_golden/mod.rs) from its fixed seed, so that the committed golden
ciphertexts (tests/golden/pbs_golden_v1.json, captured by the reference on an H100) can be DECRYPTED and compared
in phase with what this backend and the oracle compute on the very same inputs under the very same keys.

Restated (test infrastructure, CPU only; the AES-CTR generator is tests/kat_vectors.py's):
  * deterministic_generators (pbs_golden/mod.rs:129-143): DeterministicSeeder(Seed(GOLDEN_SEED)); its first uniform
    u128 keys the encryption generator's MASK stream, its second the NOISE stream (EncryptionRandomGenerator::new,
    commons/generators/encryption/mod.rs:110-115), its third the secret generator.
  * secrets: input LWE key first, then the GLWE key (one byte & 1 per bit, uniform_binary.rs:9-21).
  * par_generate_lwe_bootstrap_key (lwe_bootstrap_key_generation.rs:250-314): forks hand out consecutive exact-size byte
    ranges, i.e. GGSW i, level l first, row r in stream order; TUniform noise needs ceil((b+2)/8) bytes per sample with
    success probability 1 (t_uniform.rs:103-155), so forked and unforked draws consume the same bytes.
  * par_generate_lwe_multi_bit_bootstrap_key (lwe_multi_bit_bootstrap_key_generation.rs:21-78,150-260): group by group,
    the 2^g GGSWs of a group in the order of the subset index, each encrypting the product of the selected key bits.
  * encrypt_golden_inputs (mod.rs:187-208): the three messages in order from the same generator, after the key.
"""
import json
import os

import numpy as np

from . import oracle as orc
from .kat_vectors import FastStream, M64

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_JSON = os.path.join(HERE, "golden", "pbs_golden_v1.json")

# shortint/parameters/v1_*/classic/tuniform/p_fail_2_minus_128/ks_pbs.rs (PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128) and
# core_crypto/algorithms/test/mod.rs:196-210 (PARAM_GPU_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128)
CLASSICAL = dict(n=918, k=1, N=2048, pbs_base_log=23, pbs_level=1, lwe_noise=45, glwe_noise=17, grouping=0)
MULTI_BIT_GROUP_4 = dict(n=920, k=1, N=2048, pbs_base_log=22, pbs_level=1, lwe_noise=45, glwe_noise=17, grouping=4)
MSG_MODULUS = 16            # message_modulus * carry_modulus
DELTA = (1 << 63) // MSG_MODULUS


def f(x):
    return (2 * x - 1) % MSG_MODULUS   # wrapping_mul(2).wrapping_sub(1).wrapping_rem(msg_modulus), x >= 1 here


def load_golden():
    with open(GOLDEN_JSON) as fh:
        g = json.load(fh)
    dec = lambda s: np.array([int(s[i:i + 16], 16) for i in range(0, len(s), 16)], dtype=np.uint64)
    return dict(seed=int(g["golden_seed"], 16), messages=g["golden_messages"], batch_size=g["batch_size"],
                classical=[dec(s) for s in g["classical"]], multi_bit_group_4=[dec(s) for s in g["multi_bit_group_4"]])


def tuniform(stream, bound_log2, count):
    """t_uniform.rs:103-131: ceil((b+2)/8) little-endian bytes, masked to b+2 bits; (c >> 1) + (c & 1) - 2^b."""
    nbytes = (bound_log2 + 2 + 7) // 8
    raw = stream.take(count * nbytes).reshape(count, nbytes).astype(np.uint64)
    c = np.zeros(count, dtype=np.uint64)
    for j in range(nbytes):
        c |= raw[:, j] << np.uint64(8 * j)
    c &= np.uint64((1 << (bound_log2 + 2)) - 1)
    return (c >> np.uint64(1)) + (c & np.uint64(1)) - np.uint64(1 << bound_log2)   # wrapping u64


def _glwe_encrypt(mask, noise, glwe_sk, k, N, body_pt, bound_log2):
    a = mask.uniform_u64(k * N)
    body = np.array(body_pt, dtype=np.uint64) + tuniform(noise, bound_log2, N)
    for j in range(k):
        orc.negacyclic_mul_add(body, glwe_sk[j * N:(j + 1) * N].astype(np.int64), a[j * N:(j + 1) * N])
    return np.concatenate([a, body])


def _ggsw(mask, noise, glwe_sk, k, N, cleartext, base_log, level, bound_log2):
    """encrypt_constant_ggsw_ciphertext (ggsw_encryption.rs:20-44,141-147): level l first; row r < k encrypts
    -S_r * m * q/B^lvl, row k encrypts m * q/B^lvl."""
    rows = []
    for lvl in range(level, 0, -1):
        factor = ((-int(cleartext)) << (64 - base_log * lvl)) & M64
        for row in range(k + 1):
            body = np.zeros(N, dtype=np.uint64)
            if row < k:
                body = (glwe_sk[row * N:(row + 1) * N] * np.uint64(factor)).astype(np.uint64)
            else:
                body[0] = (-factor) & M64
            rows.append(_glwe_encrypt(mask, noise, glwe_sk, k, N, body, bound_log2))
    return rows


def material(P, seed, messages):
    """-> dict(small_sk, glwe_sk, bsk (standard domain, the reference's container order), inputs [len(messages)][n+1])"""
    n, k, N, g = P["n"], P["k"], P["N"], P["grouping"]
    seeder = FastStream(seed)
    s_mask, s_noise, s_secret = (int.from_bytes(seeder.take(16).tobytes(), "little") for _ in range(3))
    mask, noise, secret = FastStream(s_mask), FastStream(s_noise), FastStream(s_secret)
    small_sk = secret.binary(n)
    glwe_sk = secret.binary(k * N)
    rows = []
    if not g:
        for i in range(n):
            rows += _ggsw(mask, noise, glwe_sk, k, N, small_sk[i], P["pbs_base_log"], P["pbs_level"], P["glwe_noise"])
    else:
        # lwe_multi_bit_bootstrap_key_generation.rs:64-76: for each group, subset index s = 0 .. 2^g - 1, bit j of s
        # (from the most significant of the g bits) selects key bit j of the group; the GGSW encrypts the product of
        # (selected ? key bit : 1 - key bit)
        for grp in range(n // g):
            bits = [int(b) for b in small_sk[grp * g:(grp + 1) * g]]
            for s in range(1 << g):
                prod = 1
                for j in range(g):
                    sel = (s >> (g - 1 - j)) & 1
                    prod *= bits[j] if sel else 1 - bits[j]
                rows += _ggsw(mask, noise, glwe_sk, k, N, prod, P["pbs_base_log"], P["pbs_level"], P["glwe_noise"])
    bsk = np.concatenate(rows)
    inputs = []
    for m in messages:  # allocate_and_encrypt_new_lwe_ciphertext: mask, then one noise sample
        a = mask.uniform_u64(n)
        body = tuniform(noise, P["lwe_noise"], 1) + a[small_sk == 1].sum(dtype=np.uint64) + np.array([m * DELTA], dtype=np.uint64)
        inputs.append(np.concatenate([a, body]))
    return dict(small_sk=small_sk, glwe_sk=glwe_sk, bsk=bsk, inputs=np.stack(inputs))


def phase(ct, sk):
    return int(orc.lwe_decrypt(ct, sk))


def decode(ph):
    return ((ph + DELTA // 2) // DELTA) % MSG_MODULUS


def phase_distance(a, b):
    d = (a - b) & M64
    return min(d, (1 << 64) - d)
