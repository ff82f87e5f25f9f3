"""Error behaviour of the boundary: the reference's GPU backend reports misuse with a message on stderr and
abort() (backends/tfhe-cuda-common/cuda/include/device.h:13-41, PANIC / check_cuda_error), and its Rust
wrappers assert on mismatched dimensions before calling in (tfhe/src/core_crypto/gpu/algorithms/
lwe_programmable_bootstrapping.rs:29-86).  The library and the host mirror keep both behaviours.

Each misuse runs in its own interpreter (an abort must not take pytest down) against the host-emulation
build of the library — same ABI layer, no GPU needed."""
import os
import signal
import subprocess
import sys
import textwrap

import pytest

from .harness import EMU_LIB, build_emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = """
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
import tfhe_rs_amd
from tfhe_rs_amd import core_crypto_gpu as gpu, ffi
lib = ffi.default_library()
st = gpu.CudaStreams.new_single_gpu(0)
S, G = st.ptr[0], 0
""" % ROOT


def run(snippet):
    build_emu()
    env = dict(os.environ, TFHE_HIP_BACKEND_LIB=EMU_LIB)
    return subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(snippet)], env=env, capture_output=True,
                          text=True, timeout=300)


ABORTS = {
    "polynomial size not a power of two": ("""
        buf = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(S, G, C.byref(buf), 10, 1, 1000, 1, 4, True, 0)
        """, "polynomial_size 1000 not supported"),
    "polynomial size beyond the reference's GPU range": ("""
        buf = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(S, G, C.byref(buf), 10, 1, 32768, 1, 4, True, 0)
        """, "polynomial_size 32768 not supported"),
    "large ring with glwe_dimension above 1": ("""
        buf = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(S, G, C.byref(buf), 10, 2, 8192, 1, 4, True, 0)
        """, "supported with glwe_dimension 1 only"),
    "launch does not match its scratch": ("""
        buf = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(S, G, C.byref(buf), 10, 1, 256, 1, 4, True, 0)
        v = gpu.CudaVec(4 * 600, st)
        lib.cuda_programmable_bootstrap_64_async(S, G, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, buf,
                                                 10, 1, 512, 4, 1, 4, 1, 0)
        """, "PBS buffer parameters do not match"),
    "more samples than the scratch was sized for": ("""
        buf = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(S, G, C.byref(buf), 10, 1, 256, 1, 4, True, 0)
        v = gpu.CudaVec(4 * 600, st)
        lib.cuda_programmable_bootstrap_64_async(S, G, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, buf,
                                                 10, 1, 256, 4, 1, 5, 1, 0)
        """, "exceeds the scratch capacity"),
    "decomposition wider than the torus": ("""
        buf = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(S, G, C.byref(buf), 10, 1, 256, 4, 4, True, 0)
        v = gpu.CudaVec(4 * 600, st)
        lib.cuda_programmable_bootstrap_64_async(S, G, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, buf,
                                                 10, 1, 256, 16, 4, 4, 1, 0)
        """, "invalid decomposition"),
    "keyswitch decomposition wider than the torus": ("""
        v = gpu.CudaVec(4096, st)
        lib.cuda_keyswitch_lwe_ciphertext_vector_64_64_async(S, G, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, 16, 4, 16, 4, 1)
        """, "keyswitch: unsupported decomposition"),
    "multi-bit modulus switch outside degree 2048 (cuda/src/crypto/torus.cuh:641-650)": ("""
        v = gpu.CudaVec(4096, st)
        lib.cuda_modulus_switch_multi_bit_64_async(S, G, v.ptr, v.ptr, 919, 11, 1024, 3)
        """, "unsupported polynomial size"),
    "noise-tests multi-bit bootstrap with more than one ciphertext (programmable_bootstrap_multibit.cu:683-685)": ("""
        buf = C.c_void_p()
        lib.scratch_cuda_multi_bit_programmable_bootstrap_noise_tests_64_async(S, G, C.byref(buf), 1, 2048, 1, 2, True)
        v = gpu.CudaVec(4 * 4100, st)
        lib.cuda_multi_bit_programmable_bootstrap_noise_tests_64_async(S, G, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr,
                                                                       buf, 8, 1, 2048, 4, 22, 1, 2, 1, 0)
        """, "should be 1"),
    "noise-tests multi-bit bootstrap outside N = 2048 (programmable_bootstrap_multibit.cu:690-693)": ("""
        buf = C.c_void_p()
        lib.scratch_cuda_multi_bit_programmable_bootstrap_noise_tests_64_async(S, G, C.byref(buf), 1, 1024, 1, 1, True)
        v = gpu.CudaVec(4 * 4100, st)
        lib.cuda_multi_bit_programmable_bootstrap_noise_tests_64_async(S, G, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr, v.ptr,
                                                                       buf, 8, 1, 1024, 4, 22, 1, 1, 1, 0)
        """, "only polynomial size 2048 is supported"),
    "cleanup of a foreign buffer": ("""
        junk = (C.c_uint64 * 64)()
        p = C.c_void_p(C.addressof(junk))
        lib.cleanup_cuda_programmable_bootstrap_64(S, G, C.byref(p))
        """, "PBS buffer"),
    "launch on a size-only radix scratch": ("""
        s = ffi.CudaStreamsFFI((C.c_void_p * 1)(S), (C.c_uint32 * 1)(0), 1)
        mem = C.c_void_p()
        nbytes = lib.scratch_cuda_propagate_single_carry_64_inplace_async(
            s, C.byref(mem), ffi.CudaLweBootstrapKeyParamsFFI(12, 1, 2048, 23, 1, 2048, 1, 0),
            ffi.CudaLweKeyswitchKeyParamsFFI(2048, 12, 4, 4), 8, 4, 4, 0, False, 0)
        assert nbytes > 8 * 2049 * 8, nbytes          # the query reports the device bytes it would take
        v = gpu.CudaVec(8 * 2049, st)
        ct = ffi.CudaRadixCiphertextFFI(v.ptr, None, None, 8, 8, 2048)
        one = (C.c_void_p * 1)(v.ptr)
        lib.cuda_propagate_single_carry_64_inplace_async(s, C.byref(ct), None, None, mem, one, one, 0, 0)
        """, "created with allocate_gpu_memory=false"),
    "symbol outside the hot path": ("""
        lib.cdll.cuda_integer_div_rem_64_async()   # a link-compatibility stub (csrc/link_stubs.hip)
        """, "cuda_integer_div_rem_64_async: not part of the MI355X PBS backend"),
    "carry propagation with message_modulus 2 (MESSAGE_1_CARRY_3-class sets)": ("""
        s = ffi.CudaStreamsFFI((C.c_void_p * 1)(S), (C.c_uint32 * 1)(0), 1)
        mem = C.c_void_p()
        lib.scratch_cuda_propagate_single_carry_64_inplace_async(
            s, C.byref(mem), ffi.CudaLweBootstrapKeyParamsFFI(12, 1, 2048, 23, 1, 2048, 1, 0),
            ffi.CudaLweKeyswitchKeyParamsFFI(2048, 12, 4, 4), 8, 2, 8, 0, True, 0)
        """, "message_modulus 2 < 3 is not supported"),
    "signed-overflow flag without the operands of the addition (integer.cuh:2368-2370)": ("""
        s = ffi.CudaStreamsFFI((C.c_void_p * 1)(S), (C.c_uint32 * 1)(0), 1)
        mem = C.c_void_p()
        lib.scratch_cuda_propagate_single_carry_64_inplace_async(
            s, C.byref(mem), ffi.CudaLweBootstrapKeyParamsFFI(12, 1, 2048, 23, 1, 2048, 1, 0),
            ffi.CudaLweKeyswitchKeyParamsFFI(2048, 12, 4, 4), 8, 4, 4, 1, True, 0)
        v = gpu.CudaVec(9 * 2049, st)
        ct = ffi.CudaRadixCiphertextFFI(v.ptr, None, None, 8, 8, 2048)
        one = (C.c_void_p * 1)(v.ptr)
        lib.cuda_propagate_single_carry_64_inplace_async(s, C.byref(ct), C.byref(ct), C.byref(ct), mem, one, one, 1, 0)
        """, "single carry propagation is not supported for overflow, try using add_and_propagate_single_carry"),
    "carry propagation on blocks whose degrees exceed what its first bootstrap separates": ("""
        s = ffi.CudaStreamsFFI((C.c_void_p * 1)(S), (C.c_uint32 * 1)(0), 1)
        mem = C.c_void_p()
        lib.scratch_cuda_propagate_single_carry_64_inplace_async(
            s, C.byref(mem), ffi.CudaLweBootstrapKeyParamsFFI(12, 1, 2048, 23, 1, 2048, 1, 0),
            ffi.CudaLweKeyswitchKeyParamsFFI(2048, 12, 4, 4), 8, 4, 4, 0, True, 0)
        v = gpu.CudaVec(8 * 2049, st)
        deg = (C.c_uint64 * 8)(*([3] * 7 + [9]))      # 3 + 3 + 3: the sum of three clean blocks, not of two
        ct = ffi.CudaRadixCiphertextFFI(v.ptr, deg, None, 8, 8, 2048)
        one = (C.c_void_p * 1)(v.ptr)
        lib.cuda_propagate_single_carry_64_inplace_async(s, C.byref(ct), None, None, mem, one, one, 0, 0)
        """, "block 7 has degree 9, the carry propagation accepts at most 6"),
    "boolean multiplication on a scratch created for a product": ("""
        s = ffi.CudaStreamsFFI((C.c_void_p * 1)(S), (C.c_uint32 * 1)(0), 1)
        mem = C.c_void_p()
        lib.scratch_cuda_integer_mult_inplace_64_async(
            s, C.byref(mem), False, False, 4, 4, ffi.CudaLweBootstrapKeyParamsFFI(12, 1, 2048, 23, 1, 2048, 1, 0),
            ffi.CudaLweKeyswitchKeyParamsFFI(2048, 12, 4, 4), 2, True, 0)
        v = gpu.CudaVec(2 * 2049, st)
        ct = ffi.CudaRadixCiphertextFFI(v.ptr, None, None, 2, 2, 2048)
        one = (C.c_void_p * 1)(v.ptr)
        lib.cuda_integer_mult_inplace_64_async(s, C.byref(ct), False, C.byref(ct), True, one, one, mem, 2048, 2)
        """, "boolean operands need a scratch created with is_boolean_left / is_boolean_right"),
    "cooperative modulus switch in a block of a size the reference does not take either (torus.cuh:460-463)": ("""
        a, b = gpu.CudaVec(101, st), gpu.CudaVec(101, st)
        lib.cuda_centered_modulus_switch_cooperative_64_async(S, 0, b.ptr, a.ptr, 100, 12, 64, 4)
        """, "supported sizes are 128 and 512 threads per block"),
    "cooperative modulus switch in place": ("""
        a = gpu.CudaVec(101, st)
        lib.cuda_centered_modulus_switch_cooperative_64_async(S, 0, a.ptr, a.ptr, 100, 12, 64, 2)
        """, "Output and input pointers must be different"),
    "radix layer on a multi-bit key whose grouping factor does not divide n": ("""
        s = ffi.CudaStreamsFFI((C.c_void_p * 1)(S), (C.c_uint32 * 1)(0), 1)
        mem = C.c_void_p()
        lib.scratch_cuda_propagate_single_carry_64_inplace_async(
            s, C.byref(mem), ffi.CudaLweBootstrapKeyParamsFFI(10, 1, 256, 4, 1, 256, 0, 3),
            ffi.CudaLweKeyswitchKeyParamsFFI(256, 10, 4, 4), 4, 4, 4, 0, True, 0)
        """, "grouping factor in 1..4 dividing the LWE dimension"),
}


@pytest.mark.parametrize("name", sorted(ABORTS))
def test_misuse_prints_and_aborts(name):
    snippet, needle = ABORTS[name]
    r = run(snippet)
    assert r.returncode == -signal.SIGABRT, (r.returncode, r.stderr[-400:])
    assert needle in r.stderr, r.stderr[-400:]


def test_host_mirror_asserts_like_the_rust_wrappers():
    r = run("""
        bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(np.zeros(10 * 4 * 256, dtype=np.uint64), 10, 1, 256, 4, 1, st)
        d_in = gpu.CudaLweCiphertextList.new(11, 2, st)      # wrong input dimension (key expects 10)
        d_out = gpu.CudaLweCiphertextList.new(256, 2, st)
        d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(np.zeros(512, dtype=np.uint64), 1, 256, st)
        idx = gpu.CudaVec.from_cpu_async(np.arange(2, dtype=np.uint64), st)
        try:
            gpu.cuda_programmable_bootstrap_lwe_ciphertext(d_in, d_out, d_lut, idx, idx, idx, bsk, st)
        except AssertionError as e:
            print("ASSERT:", e)
        """)
    assert r.returncode == 0, r.stderr[-400:]
    assert "ASSERT: Mismatched input LweDimension" in r.stdout


def test_missing_library_fails_loudly_instead_of_falling_back():
    env = dict(os.environ, TFHE_HIP_BACKEND_LIB="/nonexistent/libtfhe_hip_backend.so")
    r = subprocess.run([sys.executable, "-c", PRELUDE], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr.lower() or "There is no CPU fallback" in r.stderr


KEY_REWRITE = """
n_in, n_out, bl, lv, B = 2048, 12, 4, 4, 5
rng = np.random.default_rng(3)
ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(rng.integers(0, 1 << 64, size=n_in * lv * (n_out + 1), dtype=np.uint64),
                                                     n_in, n_out, bl, lv, st)
d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(rng.integers(0, 1 << 64, size=(B, n_in + 1), dtype=np.uint64), st)
d_out = gpu.CudaLweCiphertextList.new(n_out, B, st)
idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), st)
gpu.cuda_keyswitch_lwe_ciphertext(ksk, d_in, d_out, idx, idx, True, st)      # builds the matrix-core layout
st.synchronize()
print("WARM", flush=True)
%s                                                                            # the key changes BEHIND the library
gpu.cuda_keyswitch_lwe_ciphertext(ksk, d_in, d_out, idx, idx, True, st)
st.synchronize()
print("SURVIVED", flush=True)
"""


def test_key_rewritten_behind_the_library_is_detected_not_served_stale():
    """INTEGRATION.md, keyswitch-key contract: the cached matrix-core layout carries a fingerprint of the key; a key
    rewritten in place without going through the library's entry points makes the next keyswitch trap instead of
    returning results computed from stale planes.  (Host emulation: device memory is host memory, the rewrite is a
    plain memset.)"""
    r = run(KEY_REWRITE % "C.memset(ksk.d_vec.ptr, 0x5A, n_in * lv * (n_out + 1) * 8)")
    assert "WARM" in r.stdout and "SURVIVED" not in r.stdout, (r.stdout, r.stderr[-300:])
    assert r.returncode < 0, r.returncode          # killed by the trap
    # the same rewrite THROUGH the library invalidates the layout and the keyswitch simply rebuilds it
    r = run(KEY_REWRITE % "lib.cuda_memset_async(ksk.d_vec.ptr, 0x5A, n_in * lv * (n_out + 1) * 8, S, G)")
    assert r.returncode == 0 and "SURVIVED" in r.stdout, r.stderr[-300:]


@pytest.mark.gpu
def test_key_rewritten_by_a_raw_hip_call_traps_on_the_gpu():
    snippet = KEY_REWRITE % ("hip = C.CDLL('libamdhip64.so'); hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]; "
                             "assert hip.hipMemset(ksk.d_vec.ptr, 0x5A, n_in * lv * (n_out + 1) * 8) == 0; "
                             "assert hip.hipDeviceSynchronize() == 0")
    env = dict(os.environ)
    env.pop("TFHE_HIP_BACKEND_LIB", None)
    r = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(snippet)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert "WARM" in r.stdout and "SURVIVED" not in r.stdout, (r.stdout, r.stderr[-400:])
    assert r.returncode != 0
