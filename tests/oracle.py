"""ctypes binding of the CPU oracle (oracle/libtfhe_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.normpath(os.path.join(_HERE, os.pardir, "oracle"))
_LIB_PATH = os.path.join(ORACLE_DIR, "libtfhe_oracle.so")


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libtfhe_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

u32, u64, i64, f64 = C.c_uint32, C.c_uint64, C.c_int64, C.c_double
P = C.c_void_p


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        L = _lib
        L.orc_modulus_switch.restype = u64
        L.orc_modulus_switch.argtypes = [u64, u32]
        L.orc_centered_ms_body_correction.restype = u64
        L.orc_centered_ms_body_correction.argtypes = [P, u32, u32]
        L.orc_closest_representable.restype = u64
        L.orc_closest_representable.argtypes = [u64, u32, u32]
        L.orc_decomp_init_state.restype = u64
        L.orc_decomp_init_state.argtypes = [u64, u32, u32]
        L.orc_decompose.argtypes = [u64, u32, u32, P]
        for f in ("orc_gl_add", "orc_gl_sub", "orc_gl_mul", "orc_gl_pow"):
            getattr(L, f).restype = u64
            getattr(L, f).argtypes = [u64, u64]
        L.orc_gl_primitive_root_2N.restype = u64
        L.orc_gl_primitive_root_2N.argtypes = [u32]
        L.orc_modswitch_pow2_to_prime.restype = u64
        L.orc_modswitch_pow2_to_prime.argtypes = [u64]
        L.orc_modswitch_prime_to_pow2.restype = u64
        L.orc_modswitch_prime_to_pow2.argtypes = [u64]
        L.orc_f64_to_i64_sat.restype = i64
        L.orc_f64_to_i64_sat.argtypes = [f64]
        L.orc_from_torus.restype = u64
        L.orc_from_torus.argtypes = [f64]
        L.orc_lwe_decrypt.restype = u64
        L.orc_rng_next.restype = u64
        L.orc_rng_tuniform.restype = i64
        L.orc_max_threads.restype = u32
        L.orc_csprng_gaussian_u64.restype = u64
        L.orc_csprng_gaussian_u64.argtypes = [P, u64, f64, u64, P]
        L.orc_csprng_bytes.argtypes = [P, u64, u64, P]
    return _lib


def _p(a):
    return a.ctypes.data_as(P)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class Rng:
    def __init__(self, seed):
        self.state = (u64 * 4)()
        lib().orc_rng_seed(self.state, u64(seed))

    def next(self):
        return lib().orc_rng_next(self.state)

    def tuniform(self, b):
        return lib().orc_rng_tuniform(self.state, u32(b))

    def binary_key(self, n):
        out = np.zeros(n, dtype=np.uint64)
        lib().orc_gen_binary_key(self.state, _p(out), u32(n))
        return out

    def uniform(self, n):
        return np.array([self.next() for _ in range(n)], dtype=np.uint64)


# ----------------------------------------------------------------- small pieces
def modulus_switch(x, log_modulus):
    return lib().orc_modulus_switch(u64(int(x)), u32(log_modulus))


def lwe_modulus_switch(lwe, log_modulus, ms_type):
    lwe = _u64(lwe)
    out = np.zeros_like(lwe)
    lib().orc_lwe_modulus_switch(_p(lwe), u32(len(lwe) - 1), u32(log_modulus), u32(ms_type), _p(out))
    return out


def centered_ms_body_correction(lwe, log_modulus):
    lwe = _u64(lwe)
    return lib().orc_centered_ms_body_correction(_p(lwe), u32(len(lwe) - 1), u32(log_modulus))


def decompose(x, base_log, level):
    out = np.zeros(level, dtype=np.int64)
    lib().orc_decompose(u64(int(x)), u32(base_log), u32(level), _p(out))
    return out


def monomial(op, poly, degree):
    poly = _u64(poly)
    out = np.zeros_like(poly)
    getattr(lib(), "orc_monomial_" + op)(_p(out), _p(poly), u32(len(poly)), u64(int(degree)))
    return out


def sample_extract(glwe, k, N, nth=0):
    glwe = _u64(glwe)
    out = np.zeros(k * N + 1, dtype=np.uint64)
    lib().orc_sample_extract(_p(out), _p(glwe), u32(k), u32(N), u32(nth))
    return out


def keyswitch(lwe_in, ksk, n_in, n_out, base_log, level):
    lwe_in = _u64(lwe_in)
    out = np.zeros(n_out + 1, dtype=np.uint64)
    lib().orc_keyswitch(_p(out), _p(lwe_in), _p(ksk), u32(n_in), u32(n_out), u32(base_log), u32(level))
    return out


def keyswitch_64_32(lwe_in, ksk32, n_in, n_out, base_log, level):
    """u64 ciphertext, u32 key -> u32 ciphertext (KS32)."""
    lwe_in = np.ascontiguousarray(lwe_in, dtype=np.uint64)
    ksk32 = np.ascontiguousarray(ksk32, dtype=np.uint32)
    out = np.zeros(n_out + 1, dtype=np.uint32)
    lib().orc_keyswitch_64_32(_p(out), _p(lwe_in), _p(ksk32), u32(n_in), u32(n_out), u32(base_log), u32(level))
    return out


def keyswitch_batch(lwe_in, ksk, n_in, n_out, base_log, level, threads=0):
    lwe_in = _u64(lwe_in).reshape(-1, n_in + 1)
    out = np.zeros((lwe_in.shape[0], n_out + 1), dtype=np.uint64)
    lib().orc_keyswitch_batch(_p(out), _p(lwe_in), _p(ksk), u32(n_in), u32(n_out), u32(base_log),
                              u32(level), u32(lwe_in.shape[0]), u32(threads))
    return out


def generate_lut(k, N, message_modulus, delta, f):
    table = np.array([f(i) % (1 << 64) for i in range(message_modulus)], dtype=np.uint64)
    out = np.zeros((k + 1) * N, dtype=np.uint64)
    lib().orc_generate_lut(_p(out), u32(k), u32(N), u32(message_modulus), u64(delta), _p(table))
    return out


def negacyclic_mul_add(out, small, big, naive=False):
    small = np.ascontiguousarray(small, dtype=np.int64)
    big = _u64(big)
    fn = lib().orc_negacyclic_mul_add_naive if naive else lib().orc_negacyclic_mul_add
    fn(_p(out), _p(small), _p(big), u32(len(big)))
    return out


# ------------------------------------------------------------------ keys / crypto
def lwe_encrypt(rng, sk, plaintext, noise_log2):
    ct = np.zeros(len(sk) + 1, dtype=np.uint64)
    lib().orc_lwe_encrypt(rng.state, _p(ct), _p(sk), u32(len(sk)), u64(int(plaintext)), u32(noise_log2))
    return ct


def lwe_decrypt(ct, sk):
    ct = _u64(ct)
    return lib().orc_lwe_decrypt(_p(ct), _p(sk), u32(len(sk)))


def gen_bsk(seed, lwe_sk, glwe_sk, k, N, base_log, level, noise_log2):
    n = len(lwe_sk)
    bsk = np.zeros(n * level * (k + 1) * (k + 1) * N, dtype=np.uint64)
    lib().orc_gen_bsk(u64(seed), _p(bsk), _p(lwe_sk), u32(n), _p(glwe_sk), u32(k), u32(N),
                      u32(base_log), u32(level), u32(noise_log2))
    return bsk


def gen_multi_bit_bsk(seed, lwe_sk, glwe_sk, k, N, base_log, level, g, noise_log2):
    n = len(lwe_sk)
    bsk = np.zeros((n // g) * (1 << g) * level * (k + 1) * (k + 1) * N, dtype=np.uint64)
    lib().orc_gen_multi_bit_bsk(u64(seed), _p(bsk), _p(lwe_sk), u32(n), _p(glwe_sk), u32(k), u32(N),
                                u32(base_log), u32(level), u32(g), u32(noise_log2))
    return bsk


def gen_ksk(seed, sk_in, sk_out, base_log, level, noise_log2):
    ksk = np.zeros(len(sk_in) * level * (len(sk_out) + 1), dtype=np.uint64)
    lib().orc_gen_ksk(u64(seed), _p(ksk), _p(sk_in), u32(len(sk_in)), _p(sk_out), u32(len(sk_out)),
                      u32(base_log), u32(level), u32(noise_log2))
    return ksk


# ------------------------------------------------------------------------ engines
def convert_bsk_ntt(bsk_std, n, k, N, level):
    out = np.zeros_like(bsk_std)
    lib().orc_convert_bsk_ntt(_p(out), _p(bsk_std), u32(n), u32(k), u32(N), u32(level))
    return out


def convert_bsk_fft(bsk_std, n, k, N, level):
    out = np.zeros(len(bsk_std), dtype=np.float64)
    lib().orc_convert_bsk_fft(_p(out), _p(bsk_std), u32(n), u32(k), u32(N), u32(level))
    return out


ENGINE_EXACT, ENGINE_NTT, ENGINE_FFT = 0, 1, 2


def pbs_batch(engine, lwe_in, lut, bsk, n, k, N, base_log, level, ms_type, threads=0):
    lwe_in = _u64(lwe_in).reshape(-1, n + 1)
    lut = _u64(lut)
    out = np.zeros((lwe_in.shape[0], k * N + 1), dtype=np.uint64)
    lib().orc_pbs_batch(u32(engine), _p(out), _p(lwe_in), _p(lut), _p(bsk), u32(n), u32(k), u32(N),
                        u32(base_log), u32(level), u32(ms_type), u32(lwe_in.shape[0]), u32(threads))
    return out


def convert_multi_bit_bsk_fft(bsk_std, n, k, N, level, g):
    """standard-domain multi-bit key -> Fourier domain (what the f64 multi-bit engine consumes)"""
    bsk_std = _u64(bsk_std)
    out = np.zeros(bsk_std.size, dtype=np.float64)
    lib().orc_convert_multi_bit_bsk_fft(_p(out), _p(bsk_std), u32(n), u32(k), u32(N), u32(level), u32(g))
    return out


_mb_fft_cache = {}


def pbs_multi_bit(engine, lwe_in, lut, bsk_std, n, k, N, base_log, level, g, threads=0):
    """engine EXACT: integer-domain keybundle from the standard key; FFT: Fourier-domain combine — the key is
    converted once per key array (cached by identity) and the batch runs under OpenMP."""
    lwe_in = _u64(lwe_in).reshape(-1, n + 1)
    lut = _u64(lut)
    out = np.zeros((lwe_in.shape[0], k * N + 1), dtype=np.uint64)
    if engine == ENGINE_FFT:
        key = (id(bsk_std), n, k, N, level, g)
        if key not in _mb_fft_cache:
            _mb_fft_cache.clear()   # one converted key at a time (320 MB at production size)
            _mb_fft_cache[key] = (bsk_std, convert_multi_bit_bsk_fft(bsk_std, n, k, N, level, g))
        bsk_f = _mb_fft_cache[key][1]
        lib().orc_pbs_multi_bit_fft_batch(_p(out), _p(lwe_in), _p(lut), _p(bsk_f), u32(n), u32(k), u32(N),
                                          u32(base_log), u32(level), u32(g), u32(lwe_in.shape[0]), u32(threads))
        return out
    bsk_std = _u64(bsk_std)
    for i in range(lwe_in.shape[0]):
        lib().orc_pbs_multi_bit_exact(_p(out[i]), _p(lwe_in[i]), _p(lut), _p(bsk_std), u32(n), u32(k), u32(N),
                                      u32(base_log), u32(level), u32(g))
    return out


def monomial_table(N):
    z = np.zeros(4 * N, dtype=np.float64)
    lib().orc_monomial_table(u32(N), _p(z))
    return z


def monomial_fourier(N, degree, z=None):
    z = monomial_table(N) if z is None else z
    m = np.zeros(N, dtype=np.float64)
    lib().orc_monomial_fourier(u32(N), C.c_uint64(int(degree)), _p(z), _p(m))
    return m


def multi_bit_modulus_switch(lwe, log_modulus, g):
    lwe = _u64(lwe)
    n = len(lwe) - 1
    deg = np.zeros((n // g) * (1 << g), dtype=np.uint64)
    body = u64(0)
    lib().orc_multi_bit_modulus_switch(_p(lwe), u32(n), u32(log_modulus), u32(g), _p(deg), C.byref(body))
    return deg, body.value


# ------------------------------------------------------------------- transforms
def fft_tables(N):
    fwd = np.zeros(N, dtype=np.float64)
    inv = np.zeros(N, dtype=np.float64)
    untw = np.zeros(N, dtype=np.float64)
    lib().orc_fft_tables(u32(N), _p(fwd), _p(inv), _p(untw))
    return fwd, inv, untw


def fft_forward_int(digits):
    digits = np.ascontiguousarray(digits, dtype=np.int64)
    out = np.zeros(len(digits), dtype=np.float64)
    lib().orc_fft_forward_int(_p(out), _p(digits), u32(len(digits)))
    return out


def fft_forward_torus(poly):
    poly = _u64(poly)
    out = np.zeros(len(poly), dtype=np.float64)
    lib().orc_fft_forward_torus(_p(out), _p(poly), u32(len(poly)))
    return out


def fft_forward_f64(compressed):
    x = np.ascontiguousarray(compressed, dtype=np.float64)
    out = np.zeros(len(x), dtype=np.float64)
    lib().orc_fft_forward_f64(_p(out), _p(x), u32(len(x)))
    return out


def fft_polynomial_mul_f64(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    out = np.zeros(len(a), dtype=np.float64)
    lib().orc_fft_polynomial_mul_f64(_p(out), _p(a), _p(b), u32(len(a)))
    return out


def fft_backward_add(poly, fourier):
    poly = _u64(poly).copy()
    f = np.ascontiguousarray(fourier, dtype=np.float64).copy()
    lib().orc_fft_backward_add(_p(poly), _p(f), u32(len(poly)))
    return poly


def ntt_forward(data):
    data = _u64(data).copy()
    lib().orc_ntt_forward(_p(data), u32(len(data)))
    return data


def ntt_inverse(data, normalize=True):
    data = _u64(data).copy()
    if normalize:
        lib().orc_ntt_normalize(_p(data), u32(len(data)))
    lib().orc_ntt_inverse(_p(data), u32(len(data)))
    return data


def blind_rotate_exact(lut, msed, bsk_std, n, k, N, base_log, level):
    """acc <- blind rotation of `lut` by the modulus-switched LWE `msed` (n+1 words), exact products."""
    acc = _u64(lut).copy()
    msed = _u64(msed)
    bsk_std = _u64(bsk_std)
    lib().orc_blind_rotate_exact(_p(acc), _p(msed), _p(bsk_std), u32(n), u32(k), u32(N), u32(base_log), u32(level))
    return acc


def dif4_convert_bsk(bsk_std, n, k, N, level):
    """the reference's own key conversion in its golden-vector configuration (tfhe_oracle_dif4.c)"""
    bsk_std = _u64(bsk_std)
    out = np.zeros(bsk_std.size, dtype=np.float64)
    lib().orc_dif4_convert_bsk(_p(out), _p(bsk_std), u32(n), u32(k), u32(N), u32(level))
    return out


def dif4_blind_rotate(lut, msed, bsk_f, n, k, N, base_log, level):
    lut, msed = _u64(lut), _u64(msed)
    acc = np.zeros((k + 1) * N, dtype=np.uint64)
    lib().orc_dif4_blind_rotate(_p(acc), _p(lut), _p(msed), _p(bsk_f), u32(n), u32(k), u32(N), u32(base_log),
                                u32(level))
    return acc


def dif4_fft(buf, N, fwd=True):
    buf = np.ascontiguousarray(buf, dtype=np.float64).copy()
    lib().orc_dif4_fft(_p(buf), u32(N), C.c_int(1 if fwd else 0))
    return buf


def csprng_bytes(seed, offset, n):
    """Bytes [offset, offset+n) of tfhe-csprng's AES-CTR byte table for Seed(seed)."""
    key = np.frombuffer(int(seed).to_bytes(16, "little"), dtype=np.uint8).copy()
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_csprng_bytes(_p(key), u64(offset), u64(n), _p(out))
    return out


def csprng_gaussian_u64(seed, offset, std, count):
    """`count` Gaussian torus samples read sequentially from byte `offset`; returns (samples, bytes used)."""
    key = np.frombuffer(int(seed).to_bytes(16, "little"), dtype=np.uint8).copy()
    out = np.zeros(count, dtype=np.uint64)
    used = lib().orc_csprng_gaussian_u64(_p(key), u64(offset), f64(std), u64(count), _p(out))
    return out, int(used)
