"""Ties the three parts of the path for which the reference ships no bytes — the centered-mean modulus switch, the
NTT-bnf engine and the multi-bit PBS — to the pieces that ARE pinned to its golden vectors (tests/test_reference_kat.py:
the regenerated valid_params_128 keys and ciphertexts, n = 833, N = 2048, PBS 23 x 1, whose Karatsuba-path outputs
reproduce the reference's SHA-256 digests)."""
import dataclasses

import numpy as np
import pytest

import tfhe_rs_amd  # noqa: F401
from tfhe_rs_amd import core_crypto_gpu as gpu

from . import kat_vectors as kv
from . import oracle as orc
from .common import C1, centered_ms_edge_vectors, centered_ms_reference, make_keys
from .harness import Ctx, oracle_pbs, use_backend
from .test_reference_kat import PARAMS, _vectors

BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _phase_distance(a, b):
    d = (int(a) - int(b)) % (1 << 64)
    return min(d, (1 << 64) - d)


@pytest.mark.gpu
def test_ntt_engines_stay_within_the_key_modswitch_bound_of_the_pinned_exact_path():
    """NTT-bnf (commons/math/ntt/ntt64.rs:144-245) = the exact integer products with every key word switched to the
    prime, x -> round(x P / 2^64), and every product switched back: a key word moves by at most 1/2 (units of 2^-64
    of the torus), so ONE CMUX moves an output coefficient by at most (k+1) l N (B/2) / 2 + 1 = 2^33 + 1, and by
    about (B / sqrt 12) sqrt((k+1) l N / 12) = 2^25.4 RMS (key roundings independent of the digits).  Checked where
    the bound applies word for word: bootstraps of ONE mask element against the first GGSW of the reference's
    regenerated bsk.cbor (n = 1; with more CMUXes a coefficient moved by 2^25 flips later decomposition digits and the
    two paths part by whole noise realisations — which is why longer chains are gated on the decrypted phase, 2^50,
    in test_reference_kat.py).  Both implementations of the engine must give the same words."""
    _, m = _vectors("valid")
    P = PARAMS["valid"]
    k, N = P["k"], P["N"]
    ggsw0 = np.ascontiguousarray(m["bsk"][:(k + 1) ** 2 * P["pbs_level"] * N])
    use_backend("hip")
    st = gpu.CudaStreams.new_single_gpu(0)
    B = 64
    rng = np.random.default_rng(41)
    cts = rng.integers(0, 1 << 64, size=(B, 2), dtype=np.uint64)      # (mask element, body): any rotation pair
    idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), st)
    lidx = gpu.CudaVec.from_cpu_async(np.zeros(B, dtype=np.uint64), st)
    d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts, st)
    lut = rng.integers(0, 1 << 64, size=(k + 1) * N, dtype=np.uint64)  # a full-width accumulator: every digit in play
    d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(lut, k, N, st)
    outs = {}
    for e in ("exact64", "ntt64", "ntt64_split"):
        bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(ggsw0, 1, k, N, P["pbs_base_log"], P["pbs_level"], st, engine=e)
        d_o = gpu.CudaLweCiphertextList.new(k * N, B, st)
        gpu.cuda_programmable_bootstrap_lwe_ciphertext(d_in, d_o, d_lut, lidx, idx, idx, bsk, st)
        outs[e] = d_o.to_lwe_ciphertext_list(st)
    assert np.array_equal(outs["ntt64"], outs["ntt64_split"])
    diff = (outs["ntt64"] - outs["exact64"]).astype(np.int64)          # wrapping difference, signed
    assert np.abs(diff).max() <= (1 << 33) + 1, int(np.abs(diff).max()).bit_length()
    live = diff[np.any(diff != 0, axis=1)]                             # rows whose mask element switched to 0 rotate nothing
    assert len(live) >= B - 2
    rms = float(np.sqrt(np.mean(live.astype(np.float64) ** 2)))
    assert 2.0 ** 24.4 < rms < 2.0 ** 26.4, np.log2(rms)


@pytest.mark.gpu
@pytest.mark.parametrize("g,base_log,level", [(2, 23, 1), (3, 15, 2), (4, 22, 1)])
def test_multi_bit_on_the_pinned_secret_keys_lands_on_the_pinned_message(g, base_log, level):
    """Multi-bit bootstrap keys (lwe_multi_bit_bootstrap_key_generation.rs:64-76) built FROM THE REGENERATED secret
    keys of the reference's golden vectors (the first 828 bits of small_lwe_secret_key — 833 is divisible by none of
    2, 3, 4 — and large_lwe_secret_key), message and LUT encodings of apps/test-vectors: the multi-bit PBS on the
    MI355X must decrypt, under the reference's big key, to what the pinned classic result decrypts to, and sit
    within the noise of a bootstrap of it in phase."""
    _, m = _vectors("valid")
    P = PARAMS["valid"]
    k, N = P["k"], P["N"]
    n = 828
    small, big = np.ascontiguousarray(m["small_sk"][:n]), m["glwe_sk"]
    bsk_h = orc.gen_multi_bit_bsk(0x6d62 + g, small, big, k, N, base_log, level, g, 17)
    use_backend("hip")
    st = gpu.CudaStreams.new_single_gpu(0)
    bsk = gpu.CudaLweMultiBitBootstrapKey.from_lwe_multi_bit_bootstrap_key(bsk_h, n, k, N, base_log, level, g, st)
    B = 6   # latency path (<= 128 LWEs); the throughput kernels are bit-identical to it (test_backend_parity.py)
    rng = orc.Rng(0x6d63)
    cts = np.stack([orc.lwe_encrypt(rng, small, kv.MSG_A << 59, 45) for _ in range(B)])
    d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts, st)
    idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), st)
    lidx = gpu.CudaVec.from_cpu_async(np.zeros(B, dtype=np.uint64), st)
    p = 1 << P["msg_bits"]
    for f in (lambda x: x, lambda x: (2 * x) % p):
        lut = orc.generate_lut(k, N, p, 1 << 59, f)
        # the pinned classic result for this LUT (its SHA-256 is the reference's, test_reference_kat.py)
        pinned = orc.pbs_batch(orc.ENGINE_EXACT, m["lwe_ks"][None, :], lut, m["bsk"], P["n"], k, N, P["pbs_base_log"],
                               P["pbs_level"], 0)[0]
        want_phase = int(orc.lwe_decrypt(pinned, big))
        want = ((want_phase + (1 << 58)) >> 59) % 32
        assert want == f(kv.MSG_A)
        d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(lut, k, N, st)
        d_out = gpu.CudaLweCiphertextList.new(k * N, B, st)
        gpu.cuda_multi_bit_programmable_bootstrap_lwe_ciphertext(d_in, d_out, d_lut, lidx, idx, idx, bsk, st)
        for o in d_out.to_lwe_ciphertext_list(st):
            ph = int(orc.lwe_decrypt(o, big))
            assert ((ph + (1 << 58)) >> 59) % 32 == want
            assert _phase_distance(ph, want_phase) < (1 << 56)   # two bootstraps' noise apart, far below delta / 2 = 2^58


@pytest.mark.parametrize("kind", BACKENDS)
def test_centered_mean_switch_on_boundary_masks_at_production_length(kind):
    """modulus_switch.rs:57-103 on masks of the PRODUCTION length (n = 918 and the odd 917) that sit on its rounding
    boundaries — all ties, tie +- 1, alternating signs, saturated words: the helper kernel against the exact-integer
    restatement (common.centered_ms_reference, independent of the C oracle), every pattern x every body."""
    use_backend(kind)
    st = gpu.CudaStreams.new_single_gpu(0)
    for n in (918, 917):
        for name, lwe in centered_ms_edge_vectors(n, 12, seed=9).items():
            d_in = gpu.CudaVec.from_cpu_async(lwe, st)
            d_out = gpu.CudaVec(n + 1, st)
            gpu.cuda_modulus_switch_ciphertext(d_out, d_in, n, 12, True, st)
            want, _ = centered_ms_reference(lwe, 12)
            assert np.array_equal(d_out.copy_to_cpu(st), want), (n, name)


@pytest.mark.gpu
def test_production_pbs_kernels_on_boundary_masks_equal_the_oracle():
    """The same boundary masks as whole ciphertexts through the PARAM_MESSAGE_2_CARRY_2 bootstrap (n = 918, centered
    switch computed in each kernel's own prologue): throughput kernel, latency kernel and generic kernel against the
    oracle, every word."""
    p = C1
    keys = make_keys(p)
    c = Ctx("hip", p, keys, "fft64")
    vecs = centered_ms_edge_vectors(p.n, p.log2N2, seed=9)
    names = [k for k in sorted(vecs) if k.endswith("body1") or k.endswith("body3")]
    cts = np.stack([vecs[k] for k in names])
    lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, lambda x: (x + 3) % p.plaintext_modulus)
    ref = oracle_pbs(p, keys, "fft64", cts, lut)
    try:
        for which in (2, 3, 1):
            c.lib.hip_backend_set_fft_kernel(which)
            out = c.pbs(cts, lut)
            bad = [names[i] for i in range(len(names)) if not np.array_equal(out[i], ref[i])]
            assert not bad, (which, bad)
    finally:
        c.lib.hip_backend_set_fft_kernel(0)
