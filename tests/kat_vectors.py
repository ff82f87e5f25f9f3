"""Regenerates the reference's golden PBS test vectors (apps/test-vectors, toy parameter set) and
hashes them, so the CPU oracle is pinned to bytes the reference itself produced.

What is restated here (test infrastructure, CPU only):
  * tfhe-csprng's generator: AES-128 in counter mode over a linear byte table, key = seed as
    little-endian u128, block t = AES_k(t as little-endian u128), started at table index 0
    (tfhe-csprng/src/generators/aes_ctr/{mod.rs:219-226,generic.rs:84-107,states.rs:87-122},
    implem/soft/block_cipher.rs:27-40,70-80); forks hand consecutive exact-size byte ranges
    to their children (generic.rs:142-176), so generation is sequential in stream order.
  * sampling: binary key = (byte & 1) per element (commons/math/random/uniform_binary.rs:9-21),
    uniform u64 = 8 bytes little-endian (uniform.rs:15-23); the toy set's Gaussian noise has
    standard deviation 0, i.e. the noise term is exactly 0 (apps/test-vectors/src/main.rs:29-30).
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


def cbor(obj) -> bytes:
    if isinstance(obj, (int, np.integer)):
        return _head(0, int(obj))
    if isinstance(obj, str):
        b = obj.encode()
        return _head(3, len(b)) + b
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
TOY = dict(n=10, k=1, N=256, pbs_base_log=24, pbs_level=1, ks_base_log=37, ks_level=1, msg_bits=4)


def _lwe_encrypt_noiseless(mask_stream, sk, pt):
    mask = mask_stream.uniform_u64(len(sk))
    body = (int(np.sum(mask[sk == 1].astype(object))) + pt) & M64
    return np.concatenate([mask, np.array([body], dtype=np.uint64)])


def _glwe_encrypt_noiseless(mask_stream, glwe_sk, k, N, body_pt):
    """body = sum_j A_j * S_j + plaintext (negacyclic), noise 0."""
    mask = mask_stream.uniform_u64(k * N)
    body = np.array(body_pt, dtype=np.uint64).copy()
    for j in range(k):
        orc.negacyclic_mul_add(body, glwe_sk[j * N:(j + 1) * N].astype(np.int64), mask[j * N:(j + 1) * N], naive=True)
    return np.concatenate([mask, body])


def generate_toy_vectors():
    """Returns {name: cbor bytes} for data/toy_params (files whose bits do not depend on f64)."""
    P = TOY
    n, k, N = P["n"], P["k"], P["N"]
    log_delta = 64 - P["msg_bits"] - 1
    secret = CsprngStream(RAND_SEED)
    mask = CsprngStream(RAND_SEED)          # EncryptionRandomGenerator::new(Seed(RAND_SEED), ..).mask
    out = {}

    glwe_sk = secret.binary(k * N)          # GlweSecretKey::generate_new_binary
    small_sk = secret.binary(n)             # LweSecretKey::generate_new_binary
    out["large_lwe_secret_key"] = ser_lwe_secret_key(glwe_sk)
    out["small_lwe_secret_key"] = ser_lwe_secret_key(small_sk)

    lwe_a = _lwe_encrypt_noiseless(mask, glwe_sk, MSG_A << log_delta)
    lwe_b = _lwe_encrypt_noiseless(mask, glwe_sk, MSG_B << log_delta)
    out["lwe_a"] = ser_lwe_ciphertext(lwe_a)
    out["lwe_b"] = ser_lwe_ciphertext(lwe_b)
    out["lwe_sum"] = ser_lwe_ciphertext(lwe_a + lwe_b)
    out["lwe_prod"] = ser_lwe_ciphertext(lwe_a * np.uint64(MSG_B))

    # keyswitch key: block i, level index (level l first) encrypts s_i * 2^(64 - base_log*level)
    ksk = []
    for i in range(k * N):
        for lvl in range(P["ks_level"], 0, -1):
            pt = (int(glwe_sk[i]) << (64 - P["ks_base_log"] * lvl)) & M64
            ksk.append(_lwe_encrypt_noiseless(mask, small_sk, pt))
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
                bsk.append(_glwe_encrypt_noiseless(mask, glwe_sk, k, N, body))
    bsk = np.concatenate(bsk)
    out["bsk"] = ser_bsk(bsk, k + 1, N, P["pbs_base_log"], P["pbs_level"])

    log_mod = (2 * N).bit_length() - 1
    msed = orc.lwe_modulus_switch(lwe_ks, log_mod, 0)                                       # ORACLE
    out["lwe_ms"] = ser_lwe_ciphertext(msed << np.uint64(64 - log_mod), native=False, modulus=1 << log_mod)

    p = 1 << P["msg_bits"]
    for name, f in (("id", lambda x: x), ("spec", lambda x: (2 * x) % p)):
        lut = orc.generate_lut(k, N, p, 1 << log_delta, f)                                  # ORACLE
        acc = orc.blind_rotate_exact(lut, msed, bsk, n, k, N, P["pbs_base_log"], P["pbs_level"])  # ORACLE
        out[f"glwe_after_{name}_br_karatsuba"] = ser_glwe_ciphertext(acc, N)
        out[f"lwe_after_{name}_pbs_karatsuba"] = ser_lwe_ciphertext(orc.sample_extract(acc, k, N, 0))  # ORACLE
    return out, dict(glwe_sk=glwe_sk, small_sk=small_sk, lwe_a=lwe_a, ksk=ksk, bsk=bsk, lwe_ks=lwe_ks, msed=msed)


def sha256_hex(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()
