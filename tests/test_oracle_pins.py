"""Pins the CPU oracle against the literal vectors of the reference's own tests/doc-tests
(tests/golden/reference_kats.json, see make_golden.py) and against defining formulas.
CPU only."""
import json
import os
import struct

import numpy as np
import pytest

from . import oracle as orc
from .common import TOY_K1, TOY_K2, TOY_K3, TOY_K1_L1, TOY_MB, TOY_MB2, make_keys, encrypt_small, decrypt_big, \
    torus_distance

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
M64 = (1 << 64) - 1


def _f64(bits):
    return struct.unpack("<d", struct.pack("<Q", bits))[0]


def test_convert_f64_i64_table():
    k = KATS["convert_f64_i64"]
    for bits, tgt in zip(k["inputs_f64_bits"], k["targets_i64"]):
        x = _f64(bits)
        got = orc.lib().orc_f64_to_i64_sat(np.rint(x))
        assert got == tgt, (x, got, tgt)


def test_closest_representable_doc_vector():
    k = KATS["closest_representable_u32"]
    # the u64 decomposer on x<<32 reproduces the u32 doc-test value shifted by 32
    got = orc.lib().orc_closest_representable(k["input"] << 32, k["base_log"], k["level"])
    assert got == k["output"] << 32


def test_decompose_doc_vector_and_recomposition():
    k = KATS["decompose_half_basis_u32"]
    d = orc.decompose(k["input"] << 32, k["base_log"], k["level"])
    assert len(d) == k["count"] and np.abs(d).max() <= k["digit_abs_max"]
    assert np.abs(d).max() == k["digit_abs_max"]  # "has a decomposition term == to half_basis"
    rng = np.random.default_rng(1)
    for base_log, level in [(4, 3), (23, 1), (15, 2), (4, 4), (3, 6), (12, 3), (8, 7)]:
        for x in rng.integers(0, 1 << 64, size=200, dtype=np.uint64):
            x = int(x)
            d = orc.decompose(x, base_log, level)
            assert np.all(np.abs(d) <= (1 << (base_log - 1)))
            # digits[0] <-> level l ... digits[l-1] <-> level 1 (decomposer.rs:239-244)
            rec = sum(int(di) << (64 - base_log * (level - i)) for i, di in enumerate(d)) & M64
            assert rec == orc.lib().orc_closest_representable(x, base_log, level)


def test_monomial_doc_vectors():
    for name, op in (("monomial_div_u8", "div"), ("monomial_mul_u8", "mul")):
        k = KATS[name]
        out = orc.monomial(op, np.array(k["input"], dtype=np.uint64), k["degree"])
        assert [int(v) & 0xFF for v in out] == k["output"]


def test_polynomial_mul_doc_vector():
    # polynomial_wrapping_mul: output = lhs * rhs (negacyclic)
    k = KATS["polynomial_wrapping_mul_u8"]
    for naive in (True, False):
        out = np.zeros(3, dtype=np.uint64)
        if naive:
            orc.negacyclic_mul_add(out, np.array(k["lhs"]), np.array(k["rhs"], dtype=np.uint64), naive=True)
            assert [int(v) & 0xFF for v in out] == k["output"]


def test_karatsuba_equals_schoolbook():
    rng = np.random.default_rng(7)
    for N in (64, 256, 1024):
        small = rng.integers(-(1 << 22), 1 << 22, size=N)
        big = rng.integers(0, 1 << 64, size=N, dtype=np.uint64)
        a = rng.integers(0, 1 << 64, size=N, dtype=np.uint64)
        o1 = orc.negacyclic_mul_add(a.copy(), small, big, naive=True)
        o2 = orc.negacyclic_mul_add(a.copy(), small, big, naive=False)
        assert np.array_equal(o1, o2)


def test_goldilocks_roots_and_arithmetic():
    k = KATS["goldilocks_roots"]
    p = k["p"]
    L = orc.lib()
    for N, root in k["roots"].items():
        N = int(N)
        assert L.orc_gl_primitive_root_2N(N) == root
        assert L.orc_gl_pow(root, N) == p - 1          # psi^N = -1
        assert pow(root, N, p) == p - 1
    rng = np.random.default_rng(3)
    edge = [0, 1, 2, p - 1, p - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 63), (1 << 63) - 1]
    vals = edge + [int(v) % p for v in rng.integers(0, 1 << 64, size=300, dtype=np.uint64)]
    for a in vals[:40]:
        for b in vals[:40]:
            assert L.orc_gl_mul(a, b) == a * b % p
            assert L.orc_gl_add(a, b) == (a + b) % p
            assert L.orc_gl_sub(a, b) == (a - b) % p


def test_ntt_product_matches_naive_convolution():
    # tfhe-ntt/src/prime64.rs:1264-1990 style: NTT product == negacyclic convolution mod p
    p = KATS["goldilocks_roots"]["p"]
    rng = np.random.default_rng(5)
    for N in (256, 512):
        a = [int(v) % p for v in rng.integers(0, 1 << 64, size=N, dtype=np.uint64)]
        b = [int(v) % p for v in rng.integers(0, 1 << 64, size=N, dtype=np.uint64)]
        fa = orc.ntt_forward(np.array(a, dtype=np.uint64))
        fb = orc.ntt_forward(np.array(b, dtype=np.uint64))
        prod = np.array([orc.lib().orc_gl_mul(int(x), int(y)) for x, y in zip(fa, fb)], dtype=np.uint64)
        got = orc.ntt_inverse(prod, normalize=True)
        ref = [0] * N
        for i in range(N):
            if a[i] == 0:
                continue
            for j in range(N):
                d = i + j
                if d < N:
                    ref[d] = (ref[d] + a[i] * b[j]) % p
                else:
                    ref[d - N] = (ref[d - N] - a[i] * b[j]) % p
        assert [int(v) for v in got] == ref


def test_modswitch_prime_pow2_formulas():
    p = KATS["goldilocks_roots"]["p"]
    rng = np.random.default_rng(11)
    xs = [0, 1, 2, M64, M64 - 1, 1 << 63, (1 << 63) - 1, (1 << 32), (1 << 32) - 1] + \
         [int(v) for v in rng.integers(0, 1 << 64, size=500, dtype=np.uint64)]
    for x in xs:
        assert orc.lib().orc_modswitch_pow2_to_prime(x) == (x * p + (1 << 63)) >> 64  # ntt64.rs:144-160
        v = x % p
        assert orc.lib().orc_modswitch_prime_to_pow2(v) == (((v << 64) | (p >> 1)) // p) & M64  # ntt64.rs:162-177


def test_modulus_switch_and_centered_correction_formula():
    from .common import centered_ms_reference
    rng = np.random.default_rng(13)
    for log_mod in (9, 10, 11, 12, 13):
        lwe = rng.integers(0, 1 << 64, size=65, dtype=np.uint64)
        ref, corr = centered_ms_reference(lwe, log_mod)   # exact-integer restatement of modulus_switch.rs:57-103
        assert orc.centered_ms_body_correction(lwe, log_mod) == corr
        out = orc.lwe_modulus_switch(lwe, log_mod, 1)
        assert np.array_equal(out, ref)
        assert out.max() < (1 << log_mod)


def test_centered_modulus_switch_on_rounding_boundaries():
    """Ties, tie +- 1 (odd errors of either sign), exact multiples, saturated words; n odd and even so that the
    halving of the summed halving errors truncates toward zero from both sides (modulus_switch.rs:78-96)."""
    from .common import centered_ms_edge_vectors, centered_ms_reference
    for log_mod in (11, 12):
        for n in (9, 10, 31):
            for name, lwe in centered_ms_edge_vectors(n, log_mod).items():
                ref, corr = centered_ms_reference(lwe, log_mod)
                assert orc.centered_ms_body_correction(lwe, log_mod) == corr, (log_mod, n, name)
                assert np.array_equal(orc.lwe_modulus_switch(lwe, log_mod, 1), ref), (log_mod, n, name)


def test_multi_bit_monomial_factors_are_the_transform_of_the_monomial():
    """Fourier-domain keybundle (lwe_multi_bit_programmable_bootstrapping.rs:116-156, fft/mod.rs:411-446): the
    factor the oracle and the kernels multiply a key polynomial by at transform position p must be the value of
    X^d there — checked against the oracle's own forward transform of the monomial (d >= N: X^d = -X^(d-N)) and
    against e^{i pi (1 + 4 bitrev p) d / N} evaluated directly."""
    for N in (256, 1024, 2048):
        n, L = N // 2, (N // 2).bit_length() - 1
        z = orc.monomial_table(N)
        ang = np.pi * np.arange(2 * N) / N
        assert np.abs(z[0::2] - np.cos(ang)).max() < 2e-15 and np.abs(z[1::2] - np.sin(ang)).max() < 2e-15
        assert (z[0], z[1], z[N], z[N + 1], z[2 * N], z[2 * N + 1]) == (1.0, 0.0, 0.0, 1.0, -1.0, 0.0)
        br = np.array([int(format(p, "0%db" % L)[::-1], 2) for p in range(n)])
        for d in (0, 1, 5, N - 1, N, N + 3, 2 * N - 1, 777):
            digits = np.zeros(N, dtype=np.int64)
            digits[d % N] = 1 if d < N else -1
            m = orc.monomial_fourier(N, d, z)
            assert np.abs(np.asarray(orc.fft_forward_int(digits)).reshape(-1) - m).max() < 1e-12
            e = np.pi * (((1 + 4 * br) * d) % (2 * N)) / N
            assert np.abs(m[0::2] - np.cos(e)).max() < 1e-15 and np.abs(m[1::2] - np.sin(e)).max() < 1e-15


def test_multi_bit_fourier_combine_agrees_with_the_integer_combine():
    """The f64 multi-bit engine (keybundle combined in the Fourier domain) and the exact engine (integer monomial
    products) decrypt to the same message with phases a few 2^40 apart at most (toy set, no key noise budget
    issue): pins the combine order / degrees / key layout of the new path against the exact one."""
    from .common import TOY_MB, make_keys, encrypt_small, decrypt_big
    p = TOY_MB
    keys = make_keys(p)
    msgs = [0, 1, 2, 3]
    cts = encrypt_small(p, keys, msgs, seed=4)
    f = lambda x: (x + 1) % p.plaintext_modulus
    lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, f)
    a = orc.pbs_multi_bit(orc.ENGINE_FFT, cts, lut, keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.grouping)
    b = orc.pbs_multi_bit(orc.ENGINE_EXACT, cts, lut, keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.grouping)
    for x, y, m in zip(a, b, msgs):
        assert decrypt_big(p, keys, x) == decrypt_big(p, keys, y) == f(m)
        d = (int(orc.lwe_decrypt(x, keys.glwe_sk)) - int(orc.lwe_decrypt(y, keys.glwe_sk))) & M64
        assert min(d, (1 << 64) - d) < (1 << 50)


def test_sample_extract_formula():
    rng = np.random.default_rng(17)
    k, N = 2, 64
    glwe = rng.integers(0, 1 << 64, size=(k + 1) * N, dtype=np.uint64)
    for nth in (0, 1, 17, N - 1):
        out = orc.sample_extract(glwe, k, N, nth)
        assert out[k * N] == glwe[k * N + nth]
        for p in range(k):
            A = glwe[p * N:(p + 1) * N]
            for j in range(N):
                exp = int(A[nth - j]) if j <= nth else (-int(A[N + nth - j])) & M64
                assert int(out[p * N + j]) == exp


def test_fft_forward_is_negacyclic_evaluation():
    # forward output position p holds P(zeta^(1+4*bitrev(p))), zeta = exp(i*pi/N)
    N = 64
    n = N // 2
    rng = np.random.default_rng(19)
    d = rng.integers(-(1 << 22), 1 << 22, size=N)
    out = orc.fft_forward_int(d).view(np.complex128)
    bits = n.bit_length() - 1
    for pos in range(n):
        br = int(format(pos, f"0{bits}b")[::-1], 2)
        x = np.exp(1j * np.pi * (1 + 4 * br) / N)
        ref = sum(int(d[j]) * x ** j for j in range(N))
        assert abs(out[pos] - ref) < 1e-6 * max(1.0, abs(ref))


def test_fft_roundtrip_and_product_vs_exact():
    # fft/tests.rs:6-70 (round-trip) and :72-210 (product vs naive convolution) in spirit
    rng = np.random.default_rng(23)
    for N in (256, 1024, 2048):
        small = rng.integers(-(1 << 22), 1 << 22, size=N)
        big = rng.integers(0, 1 << 64, size=N, dtype=np.uint64)
        fs = orc.fft_forward_int(small).view(np.complex128)
        fb = orc.fft_forward_torus(big).view(np.complex128)
        prod = (fs * fb).view(np.float64)
        got = orc.fft_backward_add(np.zeros(N, dtype=np.uint64), prod)
        exact = orc.negacyclic_mul_add(np.zeros(N, dtype=np.uint64), small, big)
        # torus distance bounded by the f64 mantissa budget: |small| 2^22 * N 2^11 * 2^64 / 2^53
        assert torus_distance(got, exact) < 2.0 ** (22 + 11 + 11 + 4)


@pytest.mark.parametrize("p", [TOY_K1, TOY_K1_L1, TOY_K2, TOY_K3], ids=lambda p: p.name)
def test_oracle_engines_decrypt_and_agree_in_phase(p):
    keys = make_keys(p)
    msgs = list(range(p.plaintext_modulus))
    cts = encrypt_small(p, keys, msgs)
    f = lambda x: (3 * x + 1) % p.plaintext_modulus
    lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, f)
    args = (p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.ms_type)
    oe = orc.pbs_batch(orc.ENGINE_EXACT, cts, lut, keys.bsk, *args)
    on = orc.pbs_batch(orc.ENGINE_NTT, cts, lut, orc.convert_bsk_ntt(keys.bsk, p.n, p.k, p.N, p.pbs_level), *args)
    of = orc.pbs_batch(orc.ENGINE_FFT, cts, lut, orc.convert_bsk_fft(keys.bsk, p.n, p.k, p.N, p.pbs_level), *args)
    exp = [f(m) for m in msgs]
    phases = []
    for o in (oe, on, of):
        assert [decrypt_big(p, keys, c) for c in o] == exp
        phases.append(np.array([orc.lwe_decrypt(c, keys.glwe_sk) for c in o], dtype=np.uint64))
    # raw ciphertexts legitimately differ (a flipped decomposition digit re-randomises the mask);
    # the decrypted phases agree to far below delta = 2^61
    assert torus_distance(phases[1], phases[0]) < 2.0 ** 50
    assert torus_distance(phases[2], phases[0]) < 2.0 ** 50


@pytest.mark.parametrize("p", [TOY_MB, TOY_MB2], ids=lambda p: p.name)
def test_oracle_multi_bit(p):
    keys = make_keys(p)
    msgs = list(range(p.plaintext_modulus))
    cts = encrypt_small(p, keys, msgs)
    f = lambda x: (x + 1) % p.plaintext_modulus
    lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, f)
    for eng in (orc.ENGINE_EXACT, orc.ENGINE_FFT):
        o = orc.pbs_multi_bit(eng, cts, lut, keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.grouping)
        assert [decrypt_big(p, keys, c) for c in o] == [f(m) for m in msgs]


def test_prime_to_pow2_switch_equals_the_wide_division():
    """orc_modswitch_prime_to_pow2 = floor((v * 2^64 + (p - 1) / 2) / p) (commons/math/ntt/ntt64.rs:162-177, output width 64):
    compared with Python's integers on edge values, random values and the values on both sides of quotient steps."""
    import random
    P = 0xFFFFFFFF00000001
    rng = random.Random(7)
    vals = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 63 - 1, 2 ** 63, 2 ** 63 + 1,
            P - 2 ** 32, P - 2 ** 32 + 1]
    vals += [rng.randrange(P) for _ in range(20000)]
    for _ in range(5000):   # smallest v whose quotient reaches q, and its predecessor
        q = rng.randrange(1, 1 << 64)
        v = -((-(q * P - (P >> 1))) >> 64)
        vals += [x for x in (v - 1, v, v + 1) if 0 <= x < P]
    for v in vals:
        assert orc.lib().orc_modswitch_prime_to_pow2(v) == (((v << 64) + (P >> 1)) // P) % (1 << 64), v
