"""Pins the oracle to bytes produced by the reference itself: regenerates ALL 36 of the reference's golden
test vectors (apps/test-vectors; BOTH parameter sets: toy n=10/N=256 without noise, and valid_params_128
n=833/N=2048/PBS 23x1/KS 3x5 with Gaussian noise: KS -> MS -> blind rotation -> sample extract, identity and
2x LUTs; the blind rotation once with exact integer products (`*_karatsuba`) and once with the reference's f64
transform in the configuration the vectors were made with — tfhe-fft's radix-4 DIF plan, x86 conversion
paths, oracle/tfhe_oracle_dif4.c) and compares SHA-256 digests with apps/test-vectors/checksums.sha256
(transcribed into tests/golden/reference_kats.json).
The key material comes from a restatement of tfhe-csprng (tests/kat_vectors.py); everything after
it — keyswitch, modulus switch, blind rotation, sample extraction — is the oracle under test."""
import json
import os

import numpy as np
import pytest

from . import kat_vectors as kv
from . import oracle as orc

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
SUMS = {"toy": KATS["test_vector_sha256_toy"]["sha256"], "valid": KATS["test_vector_sha256_valid"]["sha256"]}
PARAMS = {"toy": kv.TOY, "valid": kv.VALID}

INTEGER_FILES = ["large_lwe_secret_key", "small_lwe_secret_key", "lwe_a", "lwe_b", "lwe_sum", "lwe_prod", "ksk",
                 "lwe_ks", "bsk", "lwe_ms", "glwe_after_id_br_karatsuba", "lwe_after_id_pbs_karatsuba",
                 "glwe_after_spec_br_karatsuba", "lwe_after_spec_pbs_karatsuba"]
# the f64 vectors: twist factors, conversions, roundings, multiply-accumulate forms and blind-rotation order of the
# reference's f64 engine, pinned to its bytes
F64_FILES = ["glwe_after_id_br", "lwe_after_id_pbs", "glwe_after_spec_br", "lwe_after_spec_pbs"]


_CACHE = {}


def _vectors(which):
    if which not in _CACHE:
        _CACHE[which] = kv.generate_vectors(PARAMS[which])
    return _CACHE[which]


@pytest.fixture(scope="module")
def vectors():
    return _vectors("toy")


def test_aes128_fips197_vector():
    ct = kv.Aes128(bytes(range(16))).encrypt_block(bytes.fromhex("00112233445566778899aabbccddeeff"))
    assert ct.hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"   # FIPS-197 appendix C.1


def test_c_and_python_csprng_agree():
    py = kv.CsprngStream(kv.RAND_SEED).take(200)
    assert orc.csprng_bytes(kv.RAND_SEED, 0, 200).tobytes() == py
    assert orc.csprng_bytes(kv.RAND_SEED, 37, 50).tobytes() == py[37:87]


def test_every_reference_digest_is_covered():
    for which in ("toy", "valid"):
        assert sorted(INTEGER_FILES + F64_FILES) == sorted(SUMS[which]), "a golden vector is not regenerated"


@pytest.mark.parametrize("which", ["toy", "valid"])
@pytest.mark.parametrize("name", INTEGER_FILES + F64_FILES)
def test_regenerated_vector_matches_reference_sha256(which, name):
    out, _ = _vectors(which)
    assert kv.sha256_hex(out[name]) == SUMS[which][name], f"{which}/{name}.cbor differs from the reference's"


@pytest.mark.parametrize("which", ["toy", "valid"])
def test_f64_path_agrees_in_phase_with_the_pinned_exact_path(which):
    """The fixed-order f64 path the GPU implements differs from the reference's dif4 plan only in
    the order of the butterflies: it must decrypt to the same message and sit within 2^50 in phase of the pinned
    exact result AND of the pinned dif4 result (whose bytes equal the reference's, test above)."""
    _, m = _vectors(which)
    P = PARAMS[which]
    n, k, N = P["n"], P["k"], P["N"]
    p = 1 << P["msg_bits"]
    bsk_f = orc.convert_bsk_fft(m["bsk"], n, k, N, P["pbs_level"])
    for f in (lambda x: x, lambda x: (2 * x) % p):
        lut = orc.generate_lut(k, N, p, 1 << 59, f)
        out_f = orc.pbs_batch(orc.ENGINE_FFT, m["lwe_ks"][None, :], lut, bsk_f, n, k, N, P["pbs_base_log"],
                              P["pbs_level"], 0)[0]
        out_e = orc.pbs_batch(orc.ENGINE_EXACT, m["lwe_ks"][None, :], lut, m["bsk"], n, k, N, P["pbs_base_log"],
                              P["pbs_level"], 0)[0]
        pf = int(orc.lwe_decrypt(out_f, m["glwe_sk"]))
        pe = int(orc.lwe_decrypt(out_e, m["glwe_sk"]))
        d = (pf - pe) % (1 << 64)
        assert min(d, (1 << 64) - d) < (1 << 50)
        acc_r = orc.dif4_blind_rotate(lut, m["msed"], orc.dif4_convert_bsk(m["bsk"], n, k, N, P["pbs_level"]), n, k, N,
                                      P["pbs_base_log"], P["pbs_level"])
        pr = int(orc.lwe_decrypt(orc.sample_extract(acc_r, k, N, 0), m["glwe_sk"]))
        d = (pf - pr) % (1 << 64)
        assert min(d, (1 << 64) - d) < (1 << 50)
        assert ((pe + (1 << 58)) >> 59) % 32 == f(kv.MSG_A)


# ------------------------------------------------------------------ the BACKEND against the reference's bytes
# Same golden vectors, but keyswitch / modulus switch / blind rotation + sample extraction are now
# computed by the backend through the C ABI ([emu] = kernel sources on the host, [hip] = MI355X)
# and the SHA-256 of ITS outputs must equal the reference's checksums.
BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("which", ["toy", "valid"])
@pytest.mark.parametrize("kind", BACKENDS)
def test_backend_reproduces_reference_golden_vectors(kind, which):
    from tfhe_rs_amd import core_crypto_gpu as gpu
    from .harness import use_backend
    _, m = _vectors(which)
    P = PARAMS[which]
    SUMS = globals()["SUMS"][which]
    n, k, N = P["n"], P["k"], P["N"]
    lib = use_backend(kind)
    st = gpu.CudaStreams.new_single_gpu(0)

    # keyswitch (big -> small) of lwe_a with the regenerated ksk  ==> lwe_ks.cbor
    ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(m["ksk"], k * N, n, P["ks_base_log"], P["ks_level"], st)
    d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(m["lwe_a"][None, :], st)
    d_ks = gpu.CudaLweCiphertextList.new(n, 1, st)
    idx = gpu.CudaVec.from_cpu_async(np.zeros(1, dtype=np.uint64), st)
    gpu.cuda_keyswitch_lwe_ciphertext(ksk, d_in, d_ks, idx, idx, True, st)
    lwe_ks = d_ks.to_lwe_ciphertext_list(st)[0]
    assert kv.sha256_hex(kv.ser_lwe_ciphertext(lwe_ks)) == SUMS["lwe_ks"]

    # modulus switch helper  ==> lwe_ms.cbor (values re-aligned on the MSBs as the reference stores them)
    log_mod = (2 * N).bit_length() - 1
    d_ms = gpu.CudaVec(n + 1, st)
    gpu.cuda_modulus_switch_ciphertext(d_ms, d_ks.d_vec, n, log_mod, False, st)
    msed = d_ms.copy_to_cpu(st)
    assert kv.sha256_hex(kv.ser_lwe_ciphertext(msed << np.uint64(64 - log_mod), native=False,
                                               modulus=1 << log_mod)) == SUMS["lwe_ms"]

    if kind == "emu" and which == "valid":
        return  # 833 exact N=2048 products per output are minutes on the host emulation; the MI355X runs them
    # PBS with the exact engine (MS -> blind rotation -> sample extract) ==> lwe_after_*_pbs_karatsuba.cbor
    bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(m["bsk"], n, k, N, P["pbs_base_log"], P["pbs_level"], st,
                                                         ms_noise_reduction=False, engine="exact64")
    p = 1 << P["msg_bits"]
    for name, f in (("id", lambda x: x), ("spec", lambda x: (2 * x) % p)):
        lut = orc.generate_lut(k, N, p, 1 << 59, f)
        d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(lut, k, N, st)
        d_out = gpu.CudaLweCiphertextList.new(k * N, 1, st)
        gpu.cuda_programmable_bootstrap_lwe_ciphertext(d_ks, d_out, d_lut, idx, idx, idx, bsk, st)
        assert lib.hip_backend_last_pbs_kernel() == 5
        out = d_out.to_lwe_ciphertext_list(st)[0]
        assert kv.sha256_hex(kv.ser_lwe_ciphertext(out)) == SUMS[f"lwe_after_{name}_pbs_karatsuba"]
        # the reference's f64 vector: same pipeline with the reference-order f64 engine (tfhe-fft's dif4 plan and
        # the x86 conversion forms, operation for operation on the GPU) ==> lwe_after_*_pbs.cbor
        b_ref = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(m["bsk"], n, k, N, P["pbs_base_log"], P["pbs_level"],
                                                               st, engine="ref64")
        d_or = gpu.CudaLweCiphertextList.new(k * N, 1, st)
        gpu.cuda_programmable_bootstrap_lwe_ciphertext(d_ks, d_or, d_lut, idx, idx, idx, b_ref, st)
        assert lib.hip_backend_last_pbs_kernel() == 11
        assert kv.sha256_hex(kv.ser_lwe_ciphertext(d_or.to_lwe_ciphertext_list(st)[0])) == SUMS[f"lwe_after_{name}_pbs"]
        # and the production engines land on the same message, within 2^50 in phase
        for engine in ("fft64", "ntt64"):
            b2 = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(m["bsk"], n, k, N, P["pbs_base_log"],
                                                                P["pbs_level"], st, engine=engine)
            d_o2 = gpu.CudaLweCiphertextList.new(k * N, 1, st)
            gpu.cuda_programmable_bootstrap_lwe_ciphertext(d_ks, d_o2, d_lut, idx, idx, idx, b2, st)
            o2 = d_o2.to_lwe_ciphertext_list(st)[0]
            d = (int(orc.lwe_decrypt(o2, m["glwe_sk"])) - int(orc.lwe_decrypt(out, m["glwe_sk"]))) % (1 << 64)
            assert min(d, (1 << 64) - d) < (1 << 50), engine
