"""Regenerates the reference's golden PBS test vectors (apps/test-vectors, toy and valid_params_128) and
hashes them, so the CPU oracle is pinned to bytes the reference itself produced.

What is restated here (test infrastructure, CPU only):
  * tfhe-csprng's generator: AES-128 in counter mode over a linear byte table, key = seed as
    little-endian u128, block t = AES_k(t as little-endian u128), started at table index 0
    (tfhe-csprng/src/generators/aes_ctr/{mod.rs:219-226,generic.rs:84-107,states.rs:87-122},
    implem/soft/block_cipher.rs:27-40,70-80); forks hand consecutive exact-size byte ranges
    to their children (generic.rs:142-176), so generation is sequential in stream order.
  * sampling: binary key = (byte & 1) per element (commons/math/random/uniform_binary.rs:9-21),
    uniform u64 = 8 bytes little-endian (uniform.rs:15-23); Gaussian noise = Marsaglia polar
    method on i64 pairs, first output, FromTorus (gaussian.rs:40-69,151-163; oracle/tfhe_oracle_kat.c);
    the noise generator is keyed by DeterministicSeeder(RAND_SEED).seed() (generators/seeder.rs:48-51,
    encryption/mod.rs) and forked children reserve 16*ceil(-128/log2(1-pi/4)) bytes per sample
    (noise_random_generator.rs:33-57).
  * the generation order of apps/test-vectors/src/main.rs:121-365: secret generator and the
    encryption generator's mask stream are both keyed with RAND_SEED = 0x74666865.
  * serde/ciborium encoding of the entities (struct -> definite map with text keys in field
    order, Vec<u64> -> definite array, newtypes -> their integer; field orders from
    cc/entities/{lwe_secret_key,lwe_ciphertext,glwe_ciphertext,lwe_keyswitch_key,
    lwe_bootstrap_key,ggsw_ciphertext_list}.rs and commons/ciphertext_modulus.rs:48-54).
Everything between key material and hashes (keyswitch, modulus switch, blind rotation with exact
polynomial products, sample extraction) is computed by the ORACLE under test.
"""
import hashlib
import math
import struct

import numpy as np

from . import oracle as orc

RAND_SEED = 0x74666865
MSG_A, MSG_B = 4, 3
M64 = (1 << 64) - 1

# ------------------------------------------------------------------ AES-128 (FIPS-197), encrypt only
_SBOX = [0] * 256


def _init_sbox():
    p = q = 1
    while True:
        p = p ^ ((p << 1) & 0xFF) ^ (0x1B if p & 0x80 else 0)
        q ^= (q << 1) & 0xFF
        q ^= (q << 2) & 0xFF
        q ^= (q << 4) & 0xFF
        if q & 0x80:
            q ^= 0x09
        x = q ^ ((q << 1 | q >> 7) & 0xFF) ^ ((q << 2 | q >> 6) & 0xFF) ^ ((q << 3 | q >> 5) & 0xFF) ^ \
            ((q << 4 | q >> 4) & 0xFF)
        _SBOX[p] = (x ^ 0x63) & 0xFF
        if p == 1:
            break
    _SBOX[0] = 0x63


_init_sbox()


def _xtime(a):
    return ((a << 1) ^ 0x1B) & 0xFF if a & 0x80 else a << 1


class Aes128:
    def __init__(self, key: bytes):
        assert len(key) == 16
        w = [list(key[4 * i:4 * i + 4]) for i in range(4)]
        rcon = 1
        for i in range(4, 44):
            t = list(w[i - 1])
            if i % 4 == 0:
                t = t[1:] + t[:1]
                t = [_SBOX[b] for b in t]
                t[0] ^= rcon
                rcon = _xtime(rcon)
            w.append([a ^ b for a, b in zip(w[i - 4], t)])
        self.rk = [sum((w[4 * r + c] for c in range(4)), []) for r in range(11)]

    def encrypt_block(self, block: bytes) -> bytes:
        s = [b ^ k for b, k in zip(block, self.rk[0])]
        for rnd in range(1, 11):
            s = [_SBOX[b] for b in s]
            # shift rows (state is column-major: s[4*c + r])
            s = [s[4 * ((c + r) % 4) + r] for c in range(4) for r in range(4)]
            if rnd != 10:
                t = []
                for c in range(4):
                    a = s[4 * c:4 * c + 4]
                    x = a[0] ^ a[1] ^ a[2] ^ a[3]
                    t += [a[0] ^ x ^ _xtime(a[0] ^ a[1]), a[1] ^ x ^ _xtime(a[1] ^ a[2]),
                          a[2] ^ x ^ _xtime(a[2] ^ a[3]), a[3] ^ x ^ _xtime(a[3] ^ a[0])]
                s = t
            s = [b ^ k for b, k in zip(s, self.rk[rnd])]
        return bytes(s)


class CsprngStream:
    """Linear byte stream of tfhe-csprng's AES-CTR generator seeded with Seed(seed)."""

    def __init__(self, seed: int):
        self.aes = Aes128(seed.to_bytes(16, "little"))
        self.ctr = 0
        self.buf = b""

    def take(self, n: int) -> bytes:
        while len(self.buf) < n:
            self.buf += self.aes.encrypt_block(self.ctr.to_bytes(16, "little"))
            self.ctr += 1
        out, self.buf = self.buf[:n], self.buf[n:]
        return out

    def binary(self, n):
        return np.frombuffer(self.take(n), dtype=np.uint8).astype(np.uint64) & np.uint64(1)

    def uniform_u64(self, n):
        return np.frombuffer(self.take(8 * n), dtype="<u8").astype(np.uint64)


# ------------------------------------------------------------------ minimal CBOR (ciborium-compatible subset)
def _head(major, v):
    if v < 24:
        return bytes([major << 5 | v])
    if v < 1 << 8:
        return bytes([major << 5 | 24, v])
    if v < 1 << 16:
        return bytes([major << 5 | 25]) + struct.pack(">H", v)
    if v < 1 << 32:
        return bytes([major << 5 | 26]) + struct.pack(">I", v)
    return bytes([major << 5 | 27]) + struct.pack(">Q", v)


def _cbor_u64_array_body(a) -> bytes:
    """Vectorised shortest-form unsigned encoding of every element (same bytes as _head(0, v))."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    be = a.astype(">u8").view(np.uint8).reshape(-1, 8)
    nbytes = np.where(a < 24, 0, np.where(a < 1 << 8, 1, np.where(a < 1 << 16, 2, np.where(a < 1 << 32, 4, 8))))
    first = np.where(a < 24, a, np.where(a < 1 << 8, 24, np.where(a < 1 << 16, 25, np.where(a < 1 << 32, 26, 27))))
    rec = np.zeros((len(a), 9), dtype=np.uint8)
    rec[:, 0] = first.astype(np.uint8)
    keep = np.zeros((len(a), 9), dtype=bool)
    keep[:, 0] = True
    for j in range(8):  # payload byte j of a w-byte big-endian value is be[:, 8 - w + j]
        sel = nbytes > j
        rec[sel, 1 + j] = be[sel, (8 - nbytes[sel] + j)]
        keep[:, 1 + j] = sel
    return rec[keep].tobytes()


def cbor(obj) -> bytes:
    if isinstance(obj, (int, np.integer)):
        return _head(0, int(obj))
    if isinstance(obj, str):
        b = obj.encode()
        return _head(3, len(b)) + b
    if isinstance(obj, np.ndarray) and len(obj) > 4096:
        return _head(4, len(obj)) + _cbor_u64_array_body(obj)
    if isinstance(obj, (list, np.ndarray)):
        return _head(4, len(obj)) + b"".join(_head(0, int(v)) for v in obj)
    if isinstance(obj, dict):  # insertion order = struct field order
        return _head(5, len(obj)) + b"".join(cbor(k) + cbor(v) for k, v in obj.items())
    raise TypeError(type(obj))


def _modulus(native=True, modulus=0):
    return {"modulus": 0 if native else modulus, "scalar_bits": 64}


def ser_lwe_secret_key(data):
    return cbor({"data": data})


def ser_lwe_ciphertext(data, native=True, modulus=0):
    return cbor({"data": data, "ciphertext_modulus": _modulus(native, modulus)})


def ser_glwe_ciphertext(data, polynomial_size):
    return cbor({"data": data, "polynomial_size": polynomial_size, "ciphertext_modulus": _modulus()})


def ser_ksk(data, base_log, level, output_lwe_size):
    return cbor({"data": data, "decomp_base_log": base_log, "decomp_level_count": level,
                 "output_lwe_size": output_lwe_size, "ciphertext_modulus": _modulus()})


def ser_bsk(data, glwe_size, polynomial_size, base_log, level):
    return cbor({"ggsw_list": {"data": data, "glwe_size": glwe_size, "polynomial_size": polynomial_size,
                               "decomp_base_log": base_log, "decomp_level_count": level,
                               "ciphertext_modulus": _modulus()}})


# ------------------------------------------------------------------ the generation pipeline
TOY = dict(n=10, k=1, N=256, pbs_base_log=24, pbs_level=1, ks_base_log=37, ks_level=1, msg_bits=4,
           lwe_std=0.0, glwe_std=0.0)
# apps/test-vectors/src/main.rs:17-25 (VALID_*): a production-size set with Gaussian noise
VALID = dict(n=833, k=1, N=2048, pbs_base_log=23, pbs_level=1, ks_base_log=3, ks_level=5, msg_bits=4,
             lwe_std=3.6158408373309336e-06, glwe_std=2.845267479601915e-15)

# Forked noise generators reserve, per sample, 16 bytes x ceil(-128 / log2(1 - pi/4)) attempts
# (commons/generators/encryption/{mod.rs:23,noise_random_generator.rs:33-57}; gaussian.rs:71-89).
NOISE_BYTES_PER_FORKED_SAMPLE = 16 * math.ceil(-128.0 / math.log2(1.0 - math.pi / 4.0))


class FastStream:
    """CsprngStream with random access, generated by the oracle library's C AES (same byte table)."""

    def __init__(self, seed: int):
        self.seed, self.pos = seed, 0

    def take(self, n):
        out = orc.csprng_bytes(self.seed, self.pos, n)
        self.pos += n
        return out

    def binary(self, n):
        return self.take(n).astype(np.uint64) & np.uint64(1)

    def uniform_u64(self, n):
        return self.take(8 * n).view("<u8").astype(np.uint64)

    def gaussian(self, std, count, forked):
        """`count` noise samples.  A forked child owns count*NOISE_BYTES_PER_FORKED_SAMPLE bytes and leaves
        the unused tail behind; an unforked draw advances by what the rejection loop consumed."""
        out, used = orc.csprng_gaussian_u64(self.seed, self.pos, std, count)
        self.pos += count * NOISE_BYTES_PER_FORKED_SAMPLE if forked else used
        return out


def _lwe_encrypt(mask_stream, noise_stream, sk, pt, std, forked):
    """cc/algorithms/lwe_encryption.rs: uniform mask, body = <mask, sk> + noise + plaintext."""
    mask = mask_stream.uniform_u64(len(sk))
    body = noise_stream.gaussian(std, 1, forked)                       # wrapping u64 array arithmetic
    body += mask[sk == 1].sum(dtype=np.uint64) + np.array([pt], dtype=np.uint64)
    return np.concatenate([mask, body])


def _glwe_encrypt(mask_stream, noise_stream, glwe_sk, k, N, body_pt, std):
    """body = sum_j A_j * S_j + noise + plaintext (negacyclic); always a forked child (GGSW row)."""
    mask = mask_stream.uniform_u64(k * N)
    body = np.array(body_pt, dtype=np.uint64) + noise_stream.gaussian(std, N, True)
    for j in range(k):
        orc.negacyclic_mul_add(body, glwe_sk[j * N:(j + 1) * N].astype(np.int64), mask[j * N:(j + 1) * N])
    return np.concatenate([mask, body])


def generate_vectors(P):
    """Returns ({name: cbor bytes}, intermediates) for one parameter set of apps/test-vectors
    (the files whose bits do not depend on the f64 FFT)."""
    n, k, N = P["n"], P["k"], P["N"]
    log_delta = 64 - P["msg_bits"] - 1
    secret = FastStream(RAND_SEED)
    mask = FastStream(RAND_SEED)            # EncryptionRandomGenerator::new(Seed(RAND_SEED), ..).mask
    # .noise is keyed by DeterministicSeeder(Seed(RAND_SEED)).seed() = the first uniform u128 of that stream
    noise = FastStream(int.from_bytes(FastStream(RAND_SEED).take(16).tobytes(), "little"))
    out = {}

    glwe_sk = secret.binary(k * N)          # GlweSecretKey::generate_new_binary
    small_sk = secret.binary(n)             # LweSecretKey::generate_new_binary
    out["large_lwe_secret_key"] = ser_lwe_secret_key(glwe_sk)
    out["small_lwe_secret_key"] = ser_lwe_secret_key(small_sk)

    lwe_a = _lwe_encrypt(mask, noise, glwe_sk, MSG_A << log_delta, P["glwe_std"], False)
    lwe_b = _lwe_encrypt(mask, noise, glwe_sk, MSG_B << log_delta, P["glwe_std"], False)
    out["lwe_a"] = ser_lwe_ciphertext(lwe_a)
    out["lwe_b"] = ser_lwe_ciphertext(lwe_b)
    out["lwe_sum"] = ser_lwe_ciphertext(lwe_a + lwe_b)
    out["lwe_prod"] = ser_lwe_ciphertext(lwe_a * np.uint64(MSG_B))

    # keyswitch key: block i, level index (level l first) encrypts s_i * 2^(64 - base_log*level);
    # each block is an encrypt_lwe_ciphertext_list, i.e. one forked child per level
    ksk = []
    for i in range(k * N):
        for lvl in range(P["ks_level"], 0, -1):
            pt = (int(glwe_sk[i]) << (64 - P["ks_base_log"] * lvl)) & M64
            ksk.append(_lwe_encrypt(mask, noise, small_sk, pt, P["lwe_std"], True))
    ksk = np.concatenate(ksk)
    out["ksk"] = ser_ksk(ksk, P["ks_base_log"], P["ks_level"], n + 1)

    lwe_ks = orc.keyswitch(lwe_a, ksk, k * N, n, P["ks_base_log"], P["ks_level"])          # ORACLE
    out["lwe_ks"] = ser_lwe_ciphertext(lwe_ks)

    # bootstrap key: GGSW_i of s_i; level l first; row r < k encrypts -S_r*m*q/B^lvl, row k encrypts m*q/B^lvl
    bsk = []
    for i in range(n):
        for lvl in range(P["pbs_level"], 0, -1):
            factor = ((-int(small_sk[i])) << (64 - P["pbs_base_log"] * lvl)) & M64
            for row in range(k + 1):
                body = np.zeros(N, dtype=np.uint64)
                if row < k:
                    body = (glwe_sk[row * N:(row + 1) * N] * np.uint64(factor)).astype(np.uint64)
                else:
                    body[0] = (-factor) & M64
                bsk.append(_glwe_encrypt(mask, noise, glwe_sk, k, N, body, P["glwe_std"]))
    bsk = np.concatenate(bsk)
    out["bsk"] = ser_bsk(bsk, k + 1, N, P["pbs_base_log"], P["pbs_level"])

    log_mod = (2 * N).bit_length() - 1
    msed = orc.lwe_modulus_switch(lwe_ks, log_mod, 0)                                       # ORACLE
    out["lwe_ms"] = ser_lwe_ciphertext(msed << np.uint64(64 - log_mod), native=False, modulus=1 << log_mod)

    p = 1 << P["msg_bits"]
    # the f64 vectors: the reference's own transform in the configuration the vectors were made with
    # (tfhe-fft radix-4 DIF plan forced by `experimental-force_fft_algo_dif4`, x86 conversion paths),
    # restated in oracle/tfhe_oracle_dif4.c
    bsk_f = orc.dif4_convert_bsk(bsk, n, k, N, P["pbs_level"])                              # ORACLE
    for name, f in (("id", lambda x: x), ("spec", lambda x: (2 * x) % p)):
        lut = orc.generate_lut(k, N, p, 1 << log_delta, f)                                  # ORACLE
        acc = orc.blind_rotate_exact(lut, msed, bsk, n, k, N, P["pbs_base_log"], P["pbs_level"])  # ORACLE
        out[f"glwe_after_{name}_br_karatsuba"] = ser_glwe_ciphertext(acc, N)
        out[f"lwe_after_{name}_pbs_karatsuba"] = ser_lwe_ciphertext(orc.sample_extract(acc, k, N, 0))  # ORACLE
        acc_f = orc.dif4_blind_rotate(lut, msed, bsk_f, n, k, N, P["pbs_base_log"], P["pbs_level"])  # ORACLE
        out[f"glwe_after_{name}_br"] = ser_glwe_ciphertext(acc_f, N)
        out[f"lwe_after_{name}_pbs"] = ser_lwe_ciphertext(orc.sample_extract(acc_f, k, N, 0))      # ORACLE
    return out, dict(glwe_sk=glwe_sk, small_sk=small_sk, lwe_a=lwe_a, ksk=ksk, bsk=bsk, lwe_ks=lwe_ks, msed=msed)


def generate_toy_vectors():
    return generate_vectors(TOY)


def generate_valid_vectors():
    return generate_vectors(VALID)


def sha256_hex(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()
