"""Raw ctypes binding of libtfhe_hip_backend.so — the Python twin of the Rust FFI the
reference generates with bindgen (backends/tfhe-cuda-backend/src/bindings.rs and
backends/tfhe-cuda-common/src/cuda_bind.rs).  One entry per symbol declared in
include/tfhe_hip_backend.h; plain pointers and sizes only.

There is NO CPU fallback here: if the HIP library is missing the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libtfhe_hip_backend.so")

_v, _u32, _u64, _i8pp, _b = C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p), C.c_bool



# FFI structs of the radix-integer layer (include/tfhe_hip_backend.h, "radix integers")
class CudaStreamsFFI(C.Structure):
    _fields_ = [("streams", C.POINTER(C.c_void_p)), ("gpu_indexes", C.POINTER(C.c_uint32)), ("gpu_count", C.c_uint32)]


class CudaRadixCiphertextFFI(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("degrees", C.POINTER(C.c_uint64)), ("noise_levels", C.POINTER(C.c_uint64)),
                ("num_radix_blocks", C.c_uint32), ("max_num_radix_blocks", C.c_uint32), ("lwe_dimension", C.c_uint32)]


class CudaLweBootstrapKeyParamsFFI(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("input_lwe_dimension", "glwe_dimension", "polynomial_size", "base_log",
                                          "level_count", "big_lwe_dimension", "pbs_type", "grouping_factor")]


class CudaLweKeyswitchKeyParamsFFI(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("input_lwe_dimension", "output_lwe_dimension", "base_log", "level_count")]


_S, _BK, _KK = CudaStreamsFFI, CudaLweBootstrapKeyParamsFFI, CudaLweKeyswitchKeyParamsFFI
_R = C.POINTER(CudaRadixCiphertextFFI)

# symbol -> (restype, argtypes); mirrors include/tfhe_hip_backend.h line by line
SIGNATURES = {
    # device runtime
    "cuda_create_stream_ffi": (_v, [_u32]),
    "cuda_destroy_stream": (None, [_v, _u32]),
    "cuda_synchronize_stream": (None, [_v, _u32]),
    "cuda_is_available": (_u32, []),
    "cuda_malloc": (_v, [_u64, _u32]),
    "cuda_malloc_async": (_v, [_u64, _v, _u32]),
    "cuda_check_valid_malloc": (_b, [_u64, _u32]),
    "cuda_device_total_memory": (_u64, [_u32]),
    "cuda_memcpy_async_to_gpu": (None, [_v, _v, _u64, _v, _u32]),
    "cuda_memcpy_async_gpu_to_gpu": (None, [_v, _v, _u64, _v, _u32]),
    "cuda_memcpy_gpu_to_gpu": (None, [_v, _v, _u64, _u32]),
    "cuda_memcpy_async_to_cpu": (None, [_v, _v, _u64, _v, _u32]),
    "cuda_memset_async": (None, [_v, _u64, _u64, _v, _u32]),
    "cuda_get_number_of_gpus": (C.c_int, []),
    "cuda_get_number_of_sms": (C.c_int, []),
    "cuda_synchronize_device": (None, [_u32]),
    "cuda_drop": (None, [_v, _u32]),
    # classic PBS
    "cuda_convert_lwe_programmable_bootstrap_key_64_async": (None, [_v, _u32, _v, _v, _u32, _u32, _u32, _u32]),
    "scratch_cuda_programmable_bootstrap_64_async":
        (_u64, [_v, _u32, _i8pp, _u32, _u32, _u32, _u32, _u32, _b, _u32]),
    "cuda_programmable_bootstrap_64_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32]),
    "cleanup_cuda_programmable_bootstrap_64": (None, [_v, _u32, _i8pp]),
    "hip_scratch_keyswitch_programmable_bootstrap_64_async":
        (_u64, [_v, _u32, _i8pp, _u32, _u32, _u32, _u32, _u32, _b, _u32]),
    "hip_keyswitch_programmable_bootstrap_chain_64_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32,
                _u32, _u32]),
    "hip_keyswitch_programmable_bootstrap_64_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32,
                _u32]),
    # multi-bit PBS
    "has_support_to_cuda_programmable_bootstrap_cg_multi_bit": (_b, [_u32, _u32, _u32, _u32, _u32]),
    "cuda_convert_lwe_multi_bit_programmable_bootstrap_key_64_async":
        (None, [_v, _u32, _v, _v, _u32, _u32, _u32, _u32, _u32]),
    "scratch_cuda_multi_bit_programmable_bootstrap_64_async": (_u64, [_v, _u32, _i8pp, _u32, _u32, _u32, _u32, _b]),
    "cuda_multi_bit_programmable_bootstrap_64_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32]),
    "cleanup_cuda_multi_bit_programmable_bootstrap_64": (None, [_v, _u32, _i8pp]),
    "scratch_cuda_multi_bit_programmable_bootstrap_noise_tests_64_async": (_u64, [_v, _u32, _i8pp, _u32, _u32, _u32, _u32, _b]),
    "cuda_multi_bit_programmable_bootstrap_noise_tests_64_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32]),
    "cleanup_cuda_multi_bit_programmable_bootstrap_noise_tests_64": (None, [_v, _u32, _i8pp]),
    # keyswitch
    "cuda_keyswitch_lwe_ciphertext_vector_64_64_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32]),
    "cuda_keyswitch_gemm_64_64_async": (None, [_v, _u32, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _b]),
    "cuda_keyswitch_lwe_ciphertext_vector_64_32_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32]),
    "cuda_keyswitch_gemm_64_32_async": (None, [_v, _u32, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _b]),
    "cuda_closest_representable_64_async": (None, [_v, _u32, _v, _v, _u32, _u32]),
    # ciphertext helpers
    "cuda_convert_lwe_ciphertext_vector_to_gpu_64_async": (None, [_v, _u32, _v, _v, _u32, _u32]),
    "cuda_convert_lwe_ciphertext_vector_to_cpu_64_async": (None, [_v, _u32, _v, _v, _u32, _u32]),
    "cuda_glwe_sample_extract_64_async": (None, [_v, _u32, _v, _v, _v, _u32, _u32, _u32, _u32, _u32]),
    "cuda_modulus_switch_inplace_64_async": (None, [_v, _u32, _v, _u32, _u32]),
    "cuda_modulus_switch_64_async": (None, [_v, _u32, _v, _v, _u32, _u32]),
    "cuda_centered_modulus_switch_64_async": (None, [_v, _u32, _v, _v, _u32, _u32]),
    "cuda_centered_modulus_switch_cooperative_64_async": (None, [_v, _u32, _v, _v, _u32, _u32, _u32, _u32]),
    "cuda_modulus_switch_multi_bit_64_async": (None, [_v, _u32, _v, _v, _u32, _u32, _u32, _u32]),
    "cuda_fourier_polynomial_mul_async": (None, [_v, _u32, _v, _v, _v, _u32, _u32]),
    "cuda_fourier_polynomial_mul_fft16x4x16_async": (None, [_v, _u32, _v, _v, _v, _u32, _u32]),
    "cuda_forward_fft_classic_async": (None, [_v, _u32, _v, _v, _u32, _u32]),
    "cuda_forward_fft16x4x16_async": (None, [_v, _u32, _v, _v, _u32, _u32]),
    "cuda_backward_fft16x4x16_async": (None, [_v, _u32, _v, _v, _u32, _u32]),
    "cuda_fft16x4x16_is_supported_async": (C.c_bool, [_u32]),
    # extensions
    "hip_convert_lwe_programmable_bootstrap_key_ntt64_async": (None, [_v, _u32, _v, _v, _u32, _u32, _u32, _u32]),
    "hip_programmable_bootstrap_ntt64_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32]),
    "hip_programmable_bootstrap_ntt64_split_supported": (C.c_bool, [_u32, _u32, _u32, _u32]),
    "hip_convert_lwe_programmable_bootstrap_key_ntt64_split_async": (None, [_v, _u32, _v, _v, _u32, _u32, _u32, _u32]),
    "hip_programmable_bootstrap_ntt64_split_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32]),
    "hip_programmable_bootstrap_ntt64_split_roundoff_status": (_u32, [_v, _u32, _v]),
    "hip_convert_lwe_programmable_bootstrap_key_exact64_async": (None, [_v, _u32, _v, _v, _u32, _u32, _u32, _u32]),
    "hip_convert_lwe_programmable_bootstrap_key_ref64_async": (None, [_v, _u32, _v, _v, _u32, _u32, _u32, _u32]),
    "hip_programmable_bootstrap_ref64_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32]),
    "hip_programmable_bootstrap_exact64_async":
        (None, [_v, _u32, _v, _v, _v, _v, _v, _v, _v, _v, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32]),
    "hip_backend_set_fft_kernel": (None, [_u32]),
    "hip_backend_last_pbs_kernel": (_u32, []),
    "hip_backend_trim_allocator": (_u64, [_u32]),
    "hip_backend_allocator_stats": (None, [_u32, C.POINTER(C.c_uint64)]),
    "hip_backend_redzone_checks": (_u64, [_u32]),
    "hip_backend_profile_ranges": (_u64, []),
    "hip_backend_version": (C.c_char_p, []),
    "hip_event_create": (_v, []),
    "hip_event_record": (None, [_v, _v]),
    "hip_event_elapsed_ms": (C.c_float, [_v, _v]),
    "hip_event_destroy": (None, [_v]),
    "hip_test_arith_async": (None, [_v, _u32, _u32, _v, _v, _u32, _u32, _u32]),
    "hip_test_transform_async": (None, [_v, _u32, _u32, _u32, _v, _v]),
    "hip_test_fft_tables_host": (None, [_u32, _v, _v, _v]),
    "hip_test_monomial_table_host": (None, [_u32, _v]),
    "hip_backend_set_keyswitch_kernel": (None, [_u32]),
    "hip_backend_last_keyswitch_path": (_u32, []),
    "hip_backend_set_keyswitch_kparts": (None, [_u32]),
    "hip_integer_set_multi_gpu_threshold": (None, [_u32]),
    "hip_integer_active_gpu_count": (_u32, [_u32, _u32, _u32, _u32]),
    "hip_backend_set_ntt_kernel": (None, [_u32]),
    "hip_backend_set_multibit_latency_groups": (None, [_u32]),
    # radix integers
    "scratch_cuda_apply_univariate_lut_64_async": (_u64, [_S, _i8pp, _v, _BK, _KK, _u32, _u32, _u32, _u64, _b, _u32]),
    "cuda_apply_univariate_lut_64_async": (None, [_S, _R, _R, _v, _i8pp, _i8pp]),
    "cleanup_cuda_apply_univariate_lut_64": (None, [_S, _i8pp]),
    "scratch_cuda_apply_many_univariate_lut_64_async": (_u64, [_S, _i8pp, _v, _BK, _KK, _u32, _u32, _u32, _u32, _u64, _b,
                                                                _u32]),
    "cuda_apply_many_univariate_lut_64_async": (None, [_S, _R, _R, _v, _i8pp, _i8pp, _u32, _u32]),
    "cleanup_cuda_apply_many_univariate_lut_64": (None, [_S, _i8pp]),
    "cuda_add_lwe_ciphertext_vector_inplace_64": (None, [_v, _u32, _R, _R]),
    "scratch_cuda_propagate_single_carry_64_inplace_async":
        (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _u32, _u32, _b, _u32]),
    "scratch_cuda_add_and_propagate_single_carry_64_inplace_async":
        (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _u32, _u32, _b, _u32]),
    "cuda_propagate_single_carry_64_inplace_async": (None, [_S, _R, _R, _R, _v, _i8pp, _i8pp, _u32, _u32]),
    "cuda_add_and_propagate_single_carry_64_inplace_async":
        (None, [_S, _R, _R, _R, _R, _v, _i8pp, _i8pp, _u32, _u32]),
    "cleanup_cuda_propagate_single_carry_64_inplace": (None, [_S, _i8pp]),
    "cleanup_cuda_add_and_propagate_single_carry_64_inplace": (None, [_S, _i8pp]),
    "scratch_cuda_integer_mult_inplace_64_async": (_u64, [_S, _i8pp, _b, _b, _u32, _u32, _BK, _KK, _u32, _b, _u32]),
    "cuda_integer_mult_inplace_64_async": (None, [_S, _R, _b, _R, _b, _i8pp, _i8pp, _v, _u32, _u32]),
    "cleanup_cuda_integer_mult_inplace_64": (None, [_S, _i8pp]),
    "cuda_negate_ciphertext_64": (None, [_S, _R, _R, _u32, _u32, _u32]),
    "cuda_scalar_addition_ciphertext_64_inplace": (None, [_S, _R, _v, _v, _u32, _u32, _u32]),
    "cuda_bitnot_ciphertext_64": (None, [_S, _R, _u32, _u32, _u32]),
    "scratch_cuda_integer_bitop_inplace_64_async": (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _u32, _u32, _b, _u32]),
    "scratch_cuda_integer_scalar_bitop_inplace_64_async": (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _u32, _u32, _b, _u32]),
    "cuda_integer_bitop_inplace_64_async": (None, [_S, _R, _R, _v, _i8pp, _i8pp]),
    "cuda_integer_scalar_bitop_inplace_64_async": (None, [_S, _R, _v, _v, _u32, _v, _i8pp, _i8pp]),
    "cleanup_cuda_integer_bitop_inplace_64": (None, [_S, _i8pp]),
    "cleanup_cuda_integer_scalar_bitop_inplace_64": (None, [_S, _i8pp]),
    "scratch_cuda_sub_and_propagate_single_carry_64_inplace_async":
        (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _u32, _u32, _b, _u32]),
    "cuda_sub_and_propagate_single_carry_64_inplace_async":
        (None, [_S, _R, _R, _R, _R, _v, _i8pp, _i8pp, _u32, _u32]),
    "cleanup_cuda_sub_and_propagate_single_carry_64_inplace": (None, [_S, _i8pp]),
    "scratch_cuda_integer_overflowing_sub_64_inplace_async": (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _u32, _u32, _b, _u32]),
    "cuda_integer_overflowing_sub_64_inplace_async": (None, [_S, _R, _R, _R, _R, _v, _i8pp, _i8pp, _u32, _u32]),
    "cleanup_cuda_integer_overflowing_sub_64_inplace": (None, [_S, _i8pp]),
    "scratch_cuda_full_propagation_64_inplace_async": (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _b, _u32]),
    "cuda_full_propagation_64_inplace_async": (None, [_S, _R, _v, _i8pp, _i8pp, _u32]),
    "cleanup_cuda_full_propagation_64_inplace": (None, [_S, _i8pp]),
    "scratch_cuda_integer_comparison_64_async": (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _u32, _u32, _b, _b, _u32]),
    "cuda_integer_comparison_64_async": (None, [_S, _R, _R, _R, _v, _i8pp, _i8pp]),
    "cleanup_cuda_integer_comparison_64": (None, [_S, _i8pp]),
    "scratch_cuda_integer_scalar_comparison_64_async": (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _u32, _u32, _b, _b, _u32]),
    "cuda_integer_scalar_comparison_64_async": (None, [_S, _R, _R, _v, _v, _v, _i8pp, _i8pp, _u32]),
    "cleanup_cuda_integer_scalar_comparison_64": (None, [_S, _i8pp]),
    "scratch_cuda_cmux_64_async": (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _u32, _b, _u32]),
    "cuda_cmux_64_async": (None, [_S, _R, _R, _R, _R, _v, _i8pp, _i8pp]),
    "cleanup_cuda_cmux_64": (None, [_S, _i8pp]),
    "scratch_cuda_logical_scalar_shift_64_inplace_async": (_u64, [_S, _i8pp, _BK, _KK, _u32, _u32, _u32, _u32, _b, _u32]),
    "cuda_logical_scalar_shift_64_inplace_async": (None, [_S, _R, _u32, _v, _i8pp, _i8pp]),
    "cleanup_cuda_logical_scalar_shift_64_inplace": (None, [_S, _i8pp]),
    "hip_integer_scratch_batch": (None, [_u32]),
    "hip_integer_mult_pbs_count": (_u64, [_v]),
    "hip_integer_propagate_pbs_count": (_u64, [_u32]),
}


class Library:
    """Loaded backend library with typed symbols."""

    def __init__(self, path=None):
        # TFHE_HIP_BACKEND_LIB: alternative build of the same library (kernel experiments)
        self.path = path or os.environ.get("TFHE_HIP_BACKEND_LIB") or DEFAULT_LIB
        if not os.path.exists(self.path):
            raise ImportError(
                f"{self.path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        self.cdll = C.CDLL(self.path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)


_default = None


def default_library():
    global _default
    if _default is None:
        _default = Library()
    return _default


def set_default_library(lib):
    """Used by the test-suite to run the host-emulation build through the same wrappers."""
    global _default
    _default = lib
