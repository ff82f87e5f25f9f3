"""The two reference entry points its noise tests use around the multi-bit blind rotation
(tfhe/src/integer/gpu/server_key/radix/tests_noise_distribution/utils/noise_simulation.rs:1186-1362):

  cuda_modulus_switch_multi_bit_64_async          (cuda/include/ciphertext.h:45-50, gpu/ffi.rs:914-936)
  {scratch_,,cleanup_}cuda_multi_bit_programmable_bootstrap_noise_tests_64[_async]
                                                  (cuda/include/pbs/programmable_bootstrap_multibit.h:44-60, gpu/ffi.rs:322-397)

driven the way `multi_bit_mod_switch` + `apply_generic_blind_rotation` drive them: the output buffer holds the input
ciphertext first, the switch writes behind it, the bootstrap reads both parts.  Checked against the oracle's multi-bit
switch (word for word) and multi-bit PBS (bit for bit), [emu] on the host build, [hip] on the MI355X."""
import ctypes as C

import numpy as np
import pytest

from tfhe_rs_amd import core_crypto_gpu as gpu

from . import oracle as orc
from .common import TOY_MB4_2048, TOY_MB_2048, decrypt_big, encrypt_small, make_keys
from .harness import oracle_pbs, use_backend

BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("n,g", [(918, 3), (920, 4), (918, 2), (918, 1), (12, 4)])
def test_multi_bit_modulus_switch_equals_oracle(kind, n, g):
    """size = the whole ciphertext (n + 1 words) as the reference's caller passes it: (n + 1) // g groups are switched;
    log_modulus is ignored in favour of log2(2 * degree) as in cuda/src/crypto/torus.cuh:148-159."""
    use_backend(kind)
    st = gpu.CudaStreams.new_single_gpu(0)
    rng = np.random.default_rng(100 * n + g)
    lwe = rng.integers(0, 1 << 64, size=n + 1, dtype=np.uint64)
    # sums that land exactly on the rounding boundary of the switch (bit 51 set, nothing below)
    lwe[0] = np.uint64((5 << 52) | (1 << 51))
    if g > 1:
        lwe[1] = np.uint64(1 << 51)
    groups = (n + 1) // g
    d_in = gpu.CudaVec.from_cpu_async(lwe, st)
    d_out = gpu.CudaVec(groups << g, st)
    gpu.cuda_modulus_switch_multi_bit_ciphertext(st, d_out, d_in, 7, 2048, g)   # a wrong log_modulus on purpose
    got = d_out.copy_to_cpu(st)
    want, _ = orc.multi_bit_modulus_switch(lwe[:n + 1] if g > 1 else np.append(lwe, np.uint64(0)), 12, g)
    if g == 1:   # n + 1 one-word groups: the body is switched as a group of its own
        want = want[:(n + 1) * 2]
    assert np.array_equal(got[:len(want)], want[:len(got)]) and len(got) == groups << g
    assert not got[::1 << g].any(), "subset 0 selects nothing: degree 0"


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("which", ["g3_l2", "g4_l1"])
def test_noise_tests_bootstrap_reads_the_switched_degrees(kind, which):
    """[lwe | degrees] -> the same bits as the standard multi-bit PBS and the oracle; a tampered degree changes the
    output (the degrees really come from the buffer), a tampered MASK word does not (the mask is not read again)."""
    p = TOY_MB_2048 if which == "g3_l2" else TOY_MB4_2048
    lib = use_backend(kind)
    keys = make_keys(p)
    st = gpu.CudaStreams.new_single_gpu(0)
    bsk = gpu.CudaLweMultiBitBootstrapKey.from_lwe_multi_bit_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log,
                                                                          p.pbs_level, p.grouping, st)
    f = lambda x: (7 * x + 2) % p.plaintext_modulus
    lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, f)
    d_lut = gpu.CudaVec.from_cpu_async(lut, st)
    zero = gpu.CudaVec.from_cpu_async(np.zeros(1, dtype=np.uint64), st)
    per, groups = 1 << p.grouping, p.n // p.grouping
    for m in (3, 9):
        ct = encrypt_small(p, keys, [m], seed=40 + m)[0]
        want = oracle_pbs(p, keys, "fft64", ct[None, :], lut)[0]
        # multi_bit_mod_switch: the input first, the switch output behind it (noise_simulation.rs:1318-1356)
        d_buf = gpu.CudaVec((per + 1) * (p.n + 1), st)
        d_buf.copy_from_cpu_async(ct, st)
        d_ct = gpu.CudaVec.from_cpu_async(ct, st)
        behind = gpu.CudaVec.__new__(gpu.CudaVec)   # a view of d_buf behind the ciphertext (as_mut_c_ptr(0).add(lwe_size))
        behind.ptr, behind.len, behind.dtype, behind.gpu_index = d_buf.ptr + (p.n + 1) * 8, groups * per, d_buf.dtype, d_buf.gpu_index
        gpu.cuda_modulus_switch_multi_bit_ciphertext(st, behind, d_ct, 12, p.N, p.grouping)
        behind.ptr = None  # not owned
        d_out = gpu.CudaVec(p.k * p.N + 1, st)

        def run():
            gpu.programmable_bootstrap_multi_bit_noise_tests(st, d_out, zero, d_lut, zero, d_buf, zero, bsk.d_vec, p.n, p.k,
                                                             p.N, p.pbs_base_log, p.pbs_level, p.grouping, 1)
            return d_out.copy_to_cpu(st)

        got = run()
        assert lib.hip_backend_last_pbs_kernel() == 4
        assert np.array_equal(got, want)
        assert decrypt_big(p, keys, got) == f(m)
        host = d_buf.copy_to_cpu(st)
        deg, _ = orc.multi_bit_modulus_switch(ct, 12, p.grouping)
        assert np.array_equal(host[p.n + 1:p.n + 1 + groups * per], deg)
        # a mask word of the leading ciphertext is NOT read by the keybundle any more ...
        host2 = host.copy()
        host2[1] ^= np.uint64(1 << 63)
        d_buf.copy_from_cpu_async(host2, st)
        assert np.array_equal(run(), want)
        # ... a degree is
        host3 = host.copy()
        host3[p.n + 1 + per + 1] = np.uint64((int(host3[p.n + 1 + per + 1]) + 1000) % (2 * p.N))
        d_buf.copy_from_cpu_async(host3, st)
        assert not np.array_equal(run(), want)
