"""Test harness: selects which build of the backend the parity tests drive.

  "hip" — the product library tfhe_rs_amd/lib/libtfhe_hip_backend.so on a real MI355X
          (tests marked `gpu`)
  "emu" — the SAME kernel sources compiled for the host by tests/emu (no GPU needed); this
          checks kernel logic only and is test infrastructure, never a product path.
"""
import os
import subprocess

import numpy as np

import tfhe_rs_amd  # noqa: F401
from tfhe_rs_amd import core_crypto_gpu as gpu
from tfhe_rs_amd import ffi

from . import oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(_HERE, "emu")
EMU_LIB = os.path.join(EMU_DIR, "libtfhe_hip_backend_emu.so")
# a host-emulation build of a tuning variant (tools/build_variants.py --emu), checked by the same tests
EMU_LIB_OVERRIDE = os.environ.get("TFHE_EMU_LIB")

_libs = {}


def build_emu():
    subprocess.check_call(["make", "-C", EMU_DIR, "-j8"], stdout=subprocess.DEVNULL)
    return EMU_LIB


def use_backend(kind):
    if kind not in _libs:
        if kind == "emu":
            build_emu()
            _libs[kind] = ffi.Library(EMU_LIB_OVERRIDE or EMU_LIB)
        else:
            _libs[kind] = ffi.Library()  # product library; ImportError if it was not built
    ffi.set_default_library(_libs[kind])
    return _libs[kind]


class Ctx:
    """Keys uploaded once per (backend, params, engine)."""

    def __init__(self, kind, p, keys, engine="fft64", with_ksk=False):
        self.kind, self.p, self.keys, self.engine = kind, p, keys, engine
        self.lib = use_backend(kind)
        self.streams = gpu.CudaStreams.new_single_gpu(0)
        if p.grouping:
            self.bsk = gpu.CudaLweMultiBitBootstrapKey.from_lwe_multi_bit_bootstrap_key(
                keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.grouping, self.streams)
        else:
            self.bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(
                keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, self.streams,
                ms_noise_reduction=bool(p.ms_type), engine=engine)
        self.ksk = None
        if with_ksk:
            self.ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(
                keys.ksk, p.k * p.N, p.n, p.ks_base_log, p.ks_level, self.streams)

    def pbs(self, cts, luts, lut_indexes=None, in_indexes=None, out_indexes=None, out_count=None,
            num_many_lut=1, lut_stride=0):
        """cts [B][n+1]; luts [L][(k+1)N] -> [out_count][kN+1]"""
        p, st = self.p, self.streams
        use_backend(self.kind)
        cts = np.ascontiguousarray(cts, dtype=np.uint64)
        luts = np.ascontiguousarray(luts, dtype=np.uint64).reshape(-1, (p.k + 1) * p.N)
        B = cts.shape[0] if in_indexes is None else len(in_indexes)
        d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts, st)
        oc = out_count if out_count is not None else B * num_many_lut
        d_out = gpu.CudaLweCiphertextList.new(p.k * p.N, oc, st)
        d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(luts, p.k, p.N, st)
        mk = lambda a, dflt: gpu.CudaVec.from_cpu_async(np.asarray(dflt if a is None else a, dtype=np.uint64), st)
        d_li = mk(lut_indexes, np.zeros(B))
        d_ii = mk(in_indexes, np.arange(B))
        d_oi = mk(out_indexes, np.arange(B))
        view = gpu.CudaLweCiphertextList(d_in.d_vec, B, p.n)
        if p.grouping:
            gpu.cuda_multi_bit_programmable_bootstrap_lwe_ciphertext(view, d_out, d_lut, d_li, d_oi, d_ii,
                                                                     self.bsk, st, num_many_lut=num_many_lut,
                                                                     lut_stride=lut_stride)
        else:
            gpu.cuda_programmable_bootstrap_lwe_ciphertext(view, d_out, d_lut, d_li, d_oi, d_ii, self.bsk, st,
                                                           num_many_lut=num_many_lut, lut_stride=lut_stride)
        return d_out.to_lwe_ciphertext_list(st)

    def keyswitch(self, cts_big, use_gemm=False):
        p, st = self.p, self.streams
        use_backend(self.kind)
        cts_big = np.ascontiguousarray(cts_big, dtype=np.uint64)
        B = cts_big.shape[0]
        d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts_big, st)
        d_out = gpu.CudaLweCiphertextList.new(p.n, B, st)
        idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), st)
        gpu.cuda_keyswitch_lwe_ciphertext(self.ksk, d_in, d_out, idx, idx, True, st, use_gemm_ks=use_gemm)
        return d_out.to_lwe_ciphertext_list(st)


def oracle_bsk(p, keys, engine):
    if engine == "fft64":
        return orc.convert_bsk_fft(keys.bsk, p.n, p.k, p.N, p.pbs_level), orc.ENGINE_FFT
    if engine in ("ntt64", "ntt64_split"):
        return orc.convert_bsk_ntt(keys.bsk, p.n, p.k, p.N, p.pbs_level), orc.ENGINE_NTT
    return keys.bsk, orc.ENGINE_EXACT


def oracle_pbs(p, keys, engine, cts, lut):
    if p.grouping:
        # f64 engine: Fourier-domain key (converted once per key, cached), OpenMP over the LWEs inside the C call
        return orc.pbs_multi_bit(orc.ENGINE_FFT if engine == "fft64" else orc.ENGINE_EXACT, cts, lut, keys.bsk,
                                 p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.grouping)
    bsk, e = oracle_bsk(p, keys, engine)
    return orc.pbs_batch(e, cts, lut, bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.ms_type)


def test_arith(lib, streams, op, values, p0=0, p1=0, out_per=1, in_per=1):
    """Run one hip_test_arith op over `values` (u64) and return the u64 outputs."""
    import ctypes as C
    values = np.ascontiguousarray(values, dtype=np.uint64)
    count = values.size // in_per
    d_in = gpu.CudaVec.from_cpu_async(values, streams)
    d_out = gpu.CudaVec(count * out_per, streams)
    lib.hip_test_arith_async(streams.ptr[0], streams.gpu_indexes[0], op, d_in.ptr, d_out.ptr, count, p0, p1)
    return d_out.copy_to_cpu(streams)


test_arith.__test__ = False
