"""Host-side mirror of `tfhe::core_crypto::gpu` for the PBS hot path, over the C ABI.

Same names, argument meaning and error behaviour (assertions on mismatched dimensions) as
the Rust module it mirrors:
  CudaStreams                 tfhe/src/core_crypto/gpu/mod.rs:33-150
  CudaVec                     tfhe/src/core_crypto/gpu/vec.rs
  CudaLweCiphertextList       tfhe/src/core_crypto/gpu/entities/lwe_ciphertext_list.rs
  CudaGlweCiphertextList      tfhe/src/core_crypto/gpu/entities/glwe_ciphertext_list.rs
  CudaLweBootstrapKey         tfhe/src/core_crypto/gpu/entities/lwe_bootstrap_key.rs:57-104
  CudaLweMultiBitBootstrapKey tfhe/src/core_crypto/gpu/entities/lwe_multi_bit_bootstrap_key.rs
  CudaLweKeyswitchKey         tfhe/src/core_crypto/gpu/entities/lwe_keyswitch_key.rs
  cuda_programmable_bootstrap_lwe_ciphertext           gpu/algorithms/lwe_programmable_bootstrapping.rs:10-136
  cuda_multi_bit_programmable_bootstrap_lwe_ciphertext gpu/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:10-145
  cuda_keyswitch_lwe_ciphertext                        gpu/algorithms/lwe_keyswitch.rs:12-143
  cuda_extract_lwe_samples_from_glwe_ciphertext_list   gpu/algorithms/glwe_sample_extraction.rs:12
The Rust original is the reference's host language; no Rust toolchain exists in this image,
so the mirror is Python (INTEGRATION.md shows the Rust binding a maintainer would add).
numpy arrays are the "CPU containers".
"""
import ctypes as C

import numpy as np

from . import ffi

U64 = np.uint64


def _lib():
    return ffi.default_library()


class CudaStreams:
    """One stream per GPU (mod.rs:33-150)."""

    def __init__(self, gpu_indexes):
        self.gpu_indexes = [int(g) for g in gpu_indexes]
        self.ptr = [_lib().cuda_create_stream_ffi(g) for g in self.gpu_indexes]

    @classmethod
    def new_single_gpu(cls, gpu_index=0):
        return cls([gpu_index])

    @classmethod
    def new_multi_gpu(cls):
        return cls(range(_lib().cuda_get_number_of_gpus()))

    def synchronize(self):
        for s, g in zip(self.ptr, self.gpu_indexes):
            _lib().cuda_synchronize_stream(s, g)

    def synchronize_one(self, i):
        _lib().cuda_synchronize_stream(self.ptr[i], self.gpu_indexes[i])

    def __len__(self):
        return len(self.ptr)

    def destroy(self):
        for s, g in zip(self.ptr, self.gpu_indexes):
            _lib().cuda_destroy_stream(s, g)
        self.ptr = []

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class CudaVec:
    """Device array of one dtype on the GPU of streams[stream_index] (vec.rs)."""

    def __init__(self, length, streams, stream_index=0, dtype=U64):
        self.len = int(length)
        self.dtype = np.dtype(dtype)
        self.gpu_index = streams.gpu_indexes[stream_index]
        nbytes = max(self.len * self.dtype.itemsize, 8)
        self.ptr = _lib().cuda_malloc(nbytes, self.gpu_index)
        _lib().cuda_memset_async(self.ptr, 0, nbytes, streams.ptr[stream_index], self.gpu_index)

    @classmethod
    def from_cpu_async(cls, src, streams, stream_index=0):
        src = np.ascontiguousarray(src)
        v = cls(src.size, streams, stream_index, src.dtype)
        v.copy_from_cpu_async(src, streams, stream_index)
        return v

    def copy_from_cpu_async(self, src, streams, stream_index=0):
        src = np.ascontiguousarray(src, dtype=self.dtype)
        assert src.size <= self.len, "CudaVec: source larger than the device vector"
        _lib().cuda_memcpy_async_to_gpu(self.ptr, src.ctypes.data_as(C.c_void_p), src.nbytes,
                                        streams.ptr[stream_index], self.gpu_index)
        # the source is pageable host memory: keep it alive until the stream is drained
        streams.synchronize_one(stream_index)

    def copy_to_cpu(self, streams, stream_index=0):
        out = np.empty(self.len, dtype=self.dtype)
        _lib().cuda_memcpy_async_to_cpu(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes,
                                        streams.ptr[stream_index], self.gpu_index)
        streams.synchronize_one(stream_index)
        return out

    def drop(self):
        if self.ptr:
            _lib().cuda_drop(self.ptr, self.gpu_index)
            self.ptr = None

    def __del__(self):
        try:
            self.drop()
        except Exception:
            pass


class CudaLweCiphertextList:
    def __init__(self, d_vec, lwe_ciphertext_count, lwe_dimension):
        self.d_vec = d_vec
        self.lwe_ciphertext_count = int(lwe_ciphertext_count)
        self.lwe_dimension = int(lwe_dimension)

    @classmethod
    def new(cls, lwe_dimension, lwe_ciphertext_count, streams, dtype=U64):
        """dtype=np.uint32: the 32-bit ciphertexts a u32 keyswitch key produces (KS32 pattern)."""
        return cls(CudaVec((lwe_dimension + 1) * lwe_ciphertext_count, streams, dtype=dtype), lwe_ciphertext_count,
                   lwe_dimension)

    @classmethod
    def from_lwe_ciphertext_list(cls, h_ct, streams):
        h_ct = np.ascontiguousarray(h_ct, dtype=U64)
        assert h_ct.ndim == 2, "expected [count][lwe_size]"
        return cls(CudaVec.from_cpu_async(h_ct.reshape(-1), streams), h_ct.shape[0], h_ct.shape[1] - 1)

    def to_lwe_ciphertext_list(self, streams):
        return self.d_vec.copy_to_cpu(streams).reshape(self.lwe_ciphertext_count, self.lwe_dimension + 1)


class CudaGlweCiphertextList:
    def __init__(self, d_vec, glwe_ciphertext_count, glwe_dimension, polynomial_size):
        self.d_vec = d_vec
        self.glwe_ciphertext_count = int(glwe_ciphertext_count)
        self.glwe_dimension = int(glwe_dimension)
        self.polynomial_size = int(polynomial_size)

    @classmethod
    def from_glwe_ciphertext_list(cls, h_ct, glwe_dimension, polynomial_size, streams):
        h_ct = np.ascontiguousarray(h_ct, dtype=U64).reshape(-1, (glwe_dimension + 1) * polynomial_size)
        return cls(CudaVec.from_cpu_async(h_ct.reshape(-1), streams), h_ct.shape[0], glwe_dimension,
                   polynomial_size)

    def to_glwe_ciphertext_list(self, streams):
        return self.d_vec.copy_to_cpu(streams).reshape(self.glwe_ciphertext_count, -1)


class CudaLweBootstrapKey:
    """Bootstrap key converted once per GPU of `streams` (lwe_bootstrap_key.rs:57-104,
    gpu/ffi.rs:744-787).  engine 'fft64' is the reference GPU path; 'ntt64' the Goldilocks
    extension (integer arithmetic modulo 2^64 - 2^32 + 1: any parameter set), 'ntt64_split' the same function — same
    bits — on the throughput kernel's f64 transforms with the key in four 16-bit limbs (N = 2048, k = 1, one level;
    a key of four times the bytes; 1.56x the integer kernel's rate on the MI355X); 'exact64' the O(N^2)
    exact-convolution verification engine, 'ref64' the reference-order f64 verification engine (tfhe-fft's dif4
    plan: reproduces the reference's f64 golden vectors)."""

    def __init__(self):
        self.d_vec = None

    @classmethod
    def from_lwe_bootstrap_key(cls, h_bsk, input_lwe_dimension, glwe_dimension, polynomial_size,
                               decomp_base_log, decomp_level_count, streams, ms_noise_reduction=False,
                               engine="fft64"):
        self = cls()
        self.input_lwe_dimension = int(input_lwe_dimension)
        self.glwe_dimension = int(glwe_dimension)
        self.polynomial_size = int(polynomial_size)
        self.decomp_base_log = int(decomp_base_log)
        self.decomp_level_count = int(decomp_level_count)
        self.ms_noise_reduction = bool(ms_noise_reduction)
        self.engine = "ntt64" if engine == "ntt64_split" else engine
        self.engine_impl = "ntt64_int" if engine == "ntt64" else engine
        if engine == "ntt64_split":
            assert _lib().hip_programmable_bootstrap_ntt64_split_supported(
                glwe_dimension, polynomial_size, decomp_level_count, decomp_base_log), \
                "parameter set outside the split-key form of the NTT engine"
        engine = self.engine_impl
        h_bsk = np.ascontiguousarray(h_bsk, dtype=U64)
        elems = (self.input_lwe_dimension * (glwe_dimension + 1) ** 2 * decomp_level_count * polynomial_size)
        assert h_bsk.size == elems, "bootstrap key container has the wrong size"
        # n*(k+1)^2*l*N f64 per GPU — the byte size of the standard key; four limb polynomials per key polynomial for
        # 'ntt64_split'
        self.d_vecs = []
        for i in range(len(streams)):
            d = CudaVec(elems * (4 if engine == "ntt64_split" else 1), streams, i, np.float64)
            conv = {"fft64": _lib().cuda_convert_lwe_programmable_bootstrap_key_64_async,
                    "ntt64_int": _lib().hip_convert_lwe_programmable_bootstrap_key_ntt64_async,
                    "ntt64_split": _lib().hip_convert_lwe_programmable_bootstrap_key_ntt64_split_async,
                    "exact64": _lib().hip_convert_lwe_programmable_bootstrap_key_exact64_async,
                    "ref64": _lib().hip_convert_lwe_programmable_bootstrap_key_ref64_async}[engine]
            conv(streams.ptr[i], streams.gpu_indexes[i], d.ptr, h_bsk.ctypes.data_as(C.c_void_p),
                 self.input_lwe_dimension, glwe_dimension, decomp_level_count, polynomial_size)
            self.d_vecs.append(d)
        streams.synchronize()
        self.d_vec = self.d_vecs[0]
        return self

    @property
    def output_lwe_dimension(self):
        return self.glwe_dimension * self.polynomial_size


class CudaLweMultiBitBootstrapKey:
    @classmethod
    def from_lwe_multi_bit_bootstrap_key(cls, h_bsk, input_lwe_dimension, glwe_dimension, polynomial_size,
                                         decomp_base_log, decomp_level_count, grouping_factor, streams):
        self = cls()
        self.input_lwe_dimension = int(input_lwe_dimension)
        self.glwe_dimension = int(glwe_dimension)
        self.polynomial_size = int(polynomial_size)
        self.decomp_base_log = int(decomp_base_log)
        self.decomp_level_count = int(decomp_level_count)
        self.grouping_factor = int(grouping_factor)
        h_bsk = np.ascontiguousarray(h_bsk, dtype=U64)
        self.d_vecs = []
        for i in range(len(streams)):
            d = CudaVec(h_bsk.size, streams, i, U64)
            _lib().cuda_convert_lwe_multi_bit_programmable_bootstrap_key_64_async(
                streams.ptr[i], streams.gpu_indexes[i], d.ptr, h_bsk.ctypes.data_as(C.c_void_p),
                self.input_lwe_dimension, glwe_dimension, decomp_level_count, polynomial_size, grouping_factor)
            self.d_vecs.append(d)
        streams.synchronize()
        self.d_vec = self.d_vecs[0]
        return self

    @property
    def output_lwe_dimension(self):
        return self.glwe_dimension * self.polynomial_size


class CudaLweKeyswitchKey:
    @classmethod
    def from_lwe_keyswitch_key(cls, h_ksk, input_key_lwe_dimension, output_key_lwe_dimension, decomp_base_log,
                               decomp_level_count, streams):
        self = cls()
        self.input_key_lwe_dimension = int(input_key_lwe_dimension)
        self.output_key_lwe_dimension = int(output_key_lwe_dimension)
        self.decomp_base_log = int(decomp_base_log)
        self.decomp_level_count = int(decomp_level_count)
        # a uint32 array is a KS32 key (gpu/ffi.rs:503-618 dispatches on the key scalar), anything else u64
        h_ksk = np.ascontiguousarray(h_ksk)
        if h_ksk.dtype != np.uint32:
            h_ksk = h_ksk.astype(U64, copy=False)
        self.scalar_bits = 8 * h_ksk.dtype.itemsize
        assert h_ksk.size == input_key_lwe_dimension * decomp_level_count * (output_key_lwe_dimension + 1)
        self.d_vecs = [CudaVec.from_cpu_async(h_ksk, streams, i) for i in range(len(streams))]
        self.d_vec = self.d_vecs[0]
        return self


# ciphertexts of the last split-key bootstrap that went through the integer kernel (cuda_programmable_bootstrap_lwe_ciphertext)
last_split_recomputed = 0


def _trivial_indexes(count, streams):
    return CudaVec.from_cpu_async(np.arange(count, dtype=U64), streams)


def cuda_programmable_bootstrap_lwe_ciphertext(input, output, accumulator, lut_indexes, output_indexes,
                                               input_indexes, bsk, streams, num_many_lut=1, lut_stride=0):
    """gpu/algorithms/lwe_programmable_bootstrapping.rs:10-136 + gpu/ffi.rs:21-92
    (scratch -> launch -> cleanup on streams.ptr[0])."""
    assert input.lwe_dimension == bsk.input_lwe_dimension, (
        f"Mismatched input LweDimension. LweCiphertext input LweDimension {input.lwe_dimension}. "
        f"BootstrapKey input LweDimension {bsk.input_lwe_dimension}.")
    assert output.lwe_dimension == bsk.output_lwe_dimension, (
        f"Mismatched output LweDimension. LweCiphertext output LweDimension {output.lwe_dimension}. "
        f"BootstrapKey output LweDimension {bsk.output_lwe_dimension}.")
    assert accumulator.glwe_dimension == bsk.glwe_dimension, "Mismatched GlweSize"
    assert accumulator.polynomial_size == bsk.polynomial_size, "Mismatched PolynomialSize"
    num_samples = input.lwe_ciphertext_count
    assert output.lwe_ciphertext_count >= num_samples * num_many_lut
    lib = _lib()
    buf = C.c_void_p()
    s, g = streams.ptr[0], streams.gpu_indexes[0]
    lib.scratch_cuda_programmable_bootstrap_64_async(
        s, g, C.byref(buf), bsk.input_lwe_dimension, bsk.glwe_dimension, bsk.polynomial_size,
        bsk.decomp_level_count, num_samples, True, 1 if bsk.ms_noise_reduction else 0)
    launch = {"fft64": lib.cuda_programmable_bootstrap_64_async,
              "ntt64_int": lib.hip_programmable_bootstrap_ntt64_async,
              "ntt64_split": lib.hip_programmable_bootstrap_ntt64_split_async,
              "exact64": lib.hip_programmable_bootstrap_exact64_async,
              "ref64": lib.hip_programmable_bootstrap_ref64_async}[bsk.engine_impl]
    launch(s, g, output.d_vec.ptr, output_indexes.ptr, accumulator.d_vec.ptr, lut_indexes.ptr,
           input.d_vec.ptr, input_indexes.ptr, bsk.d_vec.ptr, buf, bsk.input_lwe_dimension, bsk.glwe_dimension,
           bsk.polynomial_size, bsk.decomp_base_log, bsk.decomp_level_count, num_samples, num_many_lut, lut_stride)
    if bsk.engine_impl == "ntt64_split":
        # the exact products of this engine come out of f64 transforms; a ciphertext whose limb products were not within 1/4
        # of integers is recomputed by the integer kernel on the same stream (never observed on a real parameter set's data):
        # the outputs are exact either way, the count is kept for whoever wants to know
        global last_split_recomputed
        last_split_recomputed = int(lib.hip_programmable_bootstrap_ntt64_split_roundoff_status(s, g, buf))
    lib.cleanup_cuda_programmable_bootstrap_64(s, g, C.byref(buf))


def cuda_keyswitch_programmable_bootstrap_lwe_ciphertext(input, output, accumulator, lut_indexes, output_indexes,
                                                         input_indexes, ksk, bsk, streams, num_many_lut=1,
                                                         lut_stride=0):
    """The shortint atomic pattern (shortint/atomic_pattern/standard.rs:162-199: keyswitch then bootstrap) as ONE
    call of the backend (extension hip_keyswitch_programmable_bootstrap_64_async): `input` holds ciphertexts under
    the big key, the keyswitched list lives in the PBS scratch."""
    assert ksk.input_key_lwe_dimension == input.lwe_dimension, "Mismatched input LweDimension"
    assert ksk.output_key_lwe_dimension == bsk.input_lwe_dimension, "keyswitch and bootstrap keys do not chain"
    assert output.lwe_dimension == bsk.output_lwe_dimension, "Mismatched output LweDimension"
    assert bsk.engine == "fft64" and ksk.scalar_bits == 64
    num_samples = input.lwe_ciphertext_count
    assert output.lwe_ciphertext_count >= num_samples * num_many_lut
    lib = _lib()
    buf = C.c_void_p()
    s, g = streams.ptr[0], streams.gpu_indexes[0]
    lib.hip_scratch_keyswitch_programmable_bootstrap_64_async(
        s, g, C.byref(buf), bsk.input_lwe_dimension, bsk.glwe_dimension, bsk.polynomial_size,
        bsk.decomp_level_count, num_samples, True, 1 if bsk.ms_noise_reduction else 0)
    lib.hip_keyswitch_programmable_bootstrap_64_async(
        s, g, output.d_vec.ptr, output_indexes.ptr, accumulator.d_vec.ptr, lut_indexes.ptr, input.d_vec.ptr,
        input_indexes.ptr, ksk.d_vec.ptr, bsk.d_vec.ptr, buf, bsk.input_lwe_dimension, bsk.glwe_dimension,
        bsk.polynomial_size, ksk.decomp_base_log, ksk.decomp_level_count, bsk.decomp_base_log,
        bsk.decomp_level_count, num_samples, num_many_lut, lut_stride)
    lib.cleanup_cuda_programmable_bootstrap_64(s, g, C.byref(buf))


def cuda_multi_bit_programmable_bootstrap_lwe_ciphertext(input, output, accumulator, lut_indexes, output_indexes,
                                                         input_indexes, multi_bit_bsk, streams, num_many_lut=1,
                                                         lut_stride=0):
    """gpu/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:10-145 + gpu/ffi.rs:208-309"""
    bsk = multi_bit_bsk
    assert input.lwe_dimension == bsk.input_lwe_dimension, "Mismatched input LweDimension"
    assert output.lwe_dimension == bsk.output_lwe_dimension, "Mismatched output LweDimension"
    assert accumulator.glwe_dimension == bsk.glwe_dimension, "Mismatched GlweSize"
    assert accumulator.polynomial_size == bsk.polynomial_size, "Mismatched PolynomialSize"
    num_samples = input.lwe_ciphertext_count
    assert output.lwe_ciphertext_count >= num_samples * num_many_lut
    lib = _lib()
    buf = C.c_void_p()
    s, g = streams.ptr[0], streams.gpu_indexes[0]
    lib.scratch_cuda_multi_bit_programmable_bootstrap_64_async(
        s, g, C.byref(buf), bsk.glwe_dimension, bsk.polynomial_size, bsk.decomp_level_count, num_samples, True)
    lib.cuda_multi_bit_programmable_bootstrap_64_async(
        s, g, output.d_vec.ptr, output_indexes.ptr, accumulator.d_vec.ptr, lut_indexes.ptr, input.d_vec.ptr,
        input_indexes.ptr, bsk.d_vec.ptr, buf, bsk.input_lwe_dimension, bsk.glwe_dimension, bsk.polynomial_size,
        bsk.grouping_factor, bsk.decomp_base_log, bsk.decomp_level_count, num_samples, num_many_lut, lut_stride)
    lib.cleanup_cuda_multi_bit_programmable_bootstrap_64(s, g, C.byref(buf))


def cuda_keyswitch_lwe_ciphertext(ksk, input, output, input_indexes, output_indexes, uses_trivial_indices,
                                  streams, use_gemm_ks=False):
    """gpu/algorithms/lwe_keyswitch.rs:12-143 + gpu/ffi.rs:503-618"""
    assert ksk.input_key_lwe_dimension == input.lwe_dimension, (
        f"Mismatched input LweDimension. LweKeyswitchKey input LweDimension: {ksk.input_key_lwe_dimension}, "
        f"input LweCiphertext LweDimension {input.lwe_dimension}.")
    assert ksk.output_key_lwe_dimension == output.lwe_dimension, (
        f"Mismatched output LweDimension. LweKeyswitchKey output LweDimension: {ksk.output_key_lwe_dimension}, "
        f"output LweCiphertext LweDimension {output.lwe_dimension}.")
    lib = _lib()
    s, g = streams.ptr[0], streams.gpu_indexes[0]
    args = (s, g, output.d_vec.ptr, output_indexes.ptr, input.d_vec.ptr, input_indexes.ptr, ksk.d_vec.ptr,
            ksk.input_key_lwe_dimension, ksk.output_key_lwe_dimension, ksk.decomp_base_log,
            ksk.decomp_level_count, input.lwe_ciphertext_count)
    if ksk.scalar_bits == 32:
        assert output.d_vec.dtype == np.uint32, "a u32 keyswitch key writes u32 ciphertexts"
        if use_gemm_ks:
            lib.cuda_keyswitch_gemm_64_32_async(*args, bool(uses_trivial_indices))
        else:
            lib.cuda_keyswitch_lwe_ciphertext_vector_64_32_async(*args)
    elif use_gemm_ks:
        lib.cuda_keyswitch_gemm_64_64_async(*args, bool(uses_trivial_indices))
    else:
        lib.cuda_keyswitch_lwe_ciphertext_vector_64_64_async(*args)


def cuda_extract_lwe_samples_from_glwe_ciphertext_list(input_glwe_list, output_lwe_list, vec_nth, lwe_per_glwe,
                                                       streams):
    """gpu/algorithms/glwe_sample_extraction.rs:12 + gpu/ffi.rs:838-883"""
    nth = np.ascontiguousarray(vec_nth, dtype=np.uint32)
    d_nth = CudaVec.from_cpu_async(nth, streams)
    _lib().cuda_glwe_sample_extract_64_async(
        streams.ptr[0], streams.gpu_indexes[0], output_lwe_list.d_vec.ptr, input_glwe_list.d_vec.ptr, d_nth.ptr,
        nth.size, lwe_per_glwe, input_glwe_list.polynomial_size, input_glwe_list.glwe_dimension,
        input_glwe_list.polynomial_size)
    streams.synchronize()


def cuda_modulus_switch_ciphertext(output_vec, input_vec, lwe_dimension, log_modulus, centered, streams):
    """gpu/ffi.rs:885-898 (plain) / cuda_centered_modulus_switch_64_async (centered, one LWE)."""
    s, g = streams.ptr[0], streams.gpu_indexes[0]
    if centered:
        _lib().cuda_centered_modulus_switch_64_async(s, g, output_vec.ptr, input_vec.ptr, lwe_dimension, log_modulus)
    else:
        _lib().cuda_modulus_switch_64_async(s, g, output_vec.ptr, input_vec.ptr, lwe_dimension + 1, log_modulus)


def cuda_centered_modulus_switch_cooperative(output_vec, input_vec, lwe_dimension, log_modulus, block_dim, streams):
    """cuda_centered_modulus_switch_cooperative_64_async as modulus_switch.rs:405-418 calls it: one LWE, the body
    correction reduced in a block of shape ``block_dim`` = (x, y) (128 or 512 threads; other sizes abort)."""
    _lib().cuda_centered_modulus_switch_cooperative_64_async(streams.ptr[0], streams.gpu_indexes[0], output_vec.ptr,
                                                             input_vec.ptr, lwe_dimension, log_modulus,
                                                             block_dim[0], block_dim[1])


def forward_fft16x4x16_async(streams, input_vec, output_vec, polynomial_size, total_polynomials):
    """gpu/ffi.rs:1054-1085: compressed real polynomials (f64 bit patterns in the CudaVecs) -> spectra in natural frequency
    order; polynomial_size 2048 only (the library aborts otherwise, as the reference)."""
    _lib().cuda_forward_fft16x4x16_async(streams.ptr[0], streams.gpu_indexes[0], input_vec.ptr, output_vec.ptr,
                                         polynomial_size, total_polynomials)


def cuda_modulus_switch_multi_bit_ciphertext(streams, lwe_array_out, lwe_array_in, log_modulus, polynomial_size,
                                             grouping_factor):
    """gpu/ffi.rs:914-936: the multi-bit modulus switch as its own launch (the reference's noise tests); `lwe_array_in`
    is a CudaVec of len() words read as len() // grouping_factor groups, 2^grouping_factor degrees written per group."""
    _lib().cuda_modulus_switch_multi_bit_64_async(streams.ptr[0], streams.gpu_indexes[0], lwe_array_out.ptr,
                                                  lwe_array_in.ptr, lwe_array_in.len, log_modulus, polynomial_size,
                                                  grouping_factor)
    streams.synchronize()


def programmable_bootstrap_multi_bit_noise_tests(streams, lwe_array_out, output_indexes, test_vector, test_vector_indexes,
                                                 lwe_array_in, input_indexes, bootstrapping_key, lwe_dimension,
                                                 glwe_dimension, polynomial_size, base_log, level, grouping_factor,
                                                 num_samples):
    """gpu/ffi.rs:322-397: multi-bit PBS on an input that already carries the output of the multi-bit modulus switch
    behind the ciphertext ([lwe | degrees]); CudaVec arguments, one ciphertext, N = 2048."""
    assert polynomial_size == 2048, (
        f"programmable_bootstrap_multi_bit_noise_tests only supports polynomial size 2048, got {polynomial_size}")
    lib = _lib()
    buf = C.c_void_p()
    s, g = streams.ptr[0], streams.gpu_indexes[0]
    lib.scratch_cuda_multi_bit_programmable_bootstrap_noise_tests_64_async(s, g, C.byref(buf), glwe_dimension,
                                                                           polynomial_size, level, num_samples, True)
    lib.cuda_multi_bit_programmable_bootstrap_noise_tests_64_async(
        s, g, lwe_array_out.ptr, output_indexes.ptr, test_vector.ptr, test_vector_indexes.ptr, lwe_array_in.ptr,
        input_indexes.ptr, bootstrapping_key.ptr, buf, lwe_dimension, glwe_dimension, polynomial_size, grouping_factor,
        base_log, level, num_samples, 1, 0)
    lib.cleanup_cuda_multi_bit_programmable_bootstrap_noise_tests_64(s, g, C.byref(buf))


def get_number_of_gpus():
    return _lib().cuda_get_number_of_gpus()


def is_cuda_available():
    return bool(_lib().cuda_is_available())
