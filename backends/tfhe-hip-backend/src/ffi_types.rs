//! Enums and `repr(C)` structs of the boundary — field for field those bindgen emits for the reference
//! (backends/tfhe-cuda-backend/src/bindings.rs: PBS_* at :115-123, CudaLweKeyswitchKeyParamsFFI :129-134,
//! CudaStreamsFFI :330-334, CudaRadixCiphertextFFI :348-355, CudaLweBootstrapKeyParamsFFI :506-515);
//! C side: include/tfhe_hip_backend.h.
use crate::ffi;

pub const PBS_TYPE_MULTI_BIT: PBS_TYPE = 0;
pub const PBS_TYPE_CLASSICAL: PBS_TYPE = 1;
pub type PBS_TYPE = ffi::c_uint;
pub const PBS_VARIANT_DEFAULT: PBS_VARIANT = 0;
pub const PBS_VARIANT_CG: PBS_VARIANT = 1;
pub const PBS_VARIANT_TBC: PBS_VARIANT = 2;
pub type PBS_VARIANT = ffi::c_uint;
pub const PBS_MS_REDUCTION_T_NO_REDUCTION: PBS_MS_REDUCTION_T = 0;
pub const PBS_MS_REDUCTION_T_CENTERED: PBS_MS_REDUCTION_T = 1;
pub type PBS_MS_REDUCTION_T = ffi::c_uint;

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct CudaStreamsFFI {
    pub streams: *const *mut ffi::c_void,
    pub gpu_indexes: *const u32,
    pub gpu_count: u32,
}

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct CudaRadixCiphertextFFI {
    pub ptr: *mut ffi::c_void,
    pub degrees: *mut u64,
    pub noise_levels: *mut u64,
    pub num_radix_blocks: u32,
    pub max_num_radix_blocks: u32,
    pub lwe_dimension: u32,
}

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct CudaLweBootstrapKeyParamsFFI {
    pub input_lwe_dimension: u32,
    pub glwe_dimension: u32,
    pub polynomial_size: u32,
    pub base_log: u32,
    pub level_count: u32,
    pub big_lwe_dimension: u32,
    pub pbs_type: u32,
    pub grouping_factor: u32,
}

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct CudaLweKeyswitchKeyParamsFFI {
    pub input_lwe_dimension: u32,
    pub output_lwe_dimension: u32,
    pub base_log: u32,
    pub level_count: u32,
}

const _: () = {
    ["Size of CudaStreamsFFI"][::std::mem::size_of::<CudaStreamsFFI>() - 24usize];
    ["Size of CudaRadixCiphertextFFI"][::std::mem::size_of::<CudaRadixCiphertextFFI>() - 40usize];
    ["Size of CudaLweBootstrapKeyParamsFFI"][::std::mem::size_of::<CudaLweBootstrapKeyParamsFFI>() - 32usize];
    ["Size of CudaLweKeyswitchKeyParamsFFI"][::std::mem::size_of::<CudaLweKeyswitchKeyParamsFFI>() - 16usize];
};
