//! FFI surface of the MI355X PBS backend, with the module layout of `tfhe-cuda-backend`
//! (backends/tfhe-cuda-backend/src/lib.rs): `bindings` (the bindgen-shaped declarations), `cuda_bind`
//! (the device-runtime functions the reference keeps in tfhe-cuda-common), `ffi` (C scalar aliases).
//!
//! A `tfhe` built with a `gpu-hip` feature that aliases `tfhe_cuda_backend` to this crate
//! (`use tfhe_hip_backend as tfhe_cuda_backend;` in tfhe/src/core_crypto/gpu/mod.rs and
//! tfhe/src/integer/gpu/mod.rs) calls the same symbols with the same prototypes; everything outside the
//! PBS hot path resolves to an abort stub of the library (csrc/link_stubs.hip).
#[allow(warnings)]
pub mod bindings;
#[allow(warnings)]
pub mod cuda_bind;
pub mod ffi;
#[allow(warnings)]
pub mod ffi_types;
pub use cuda_bind::*;

#[cfg(test)]
mod tests {
    use super::bindings::*;
    use super::cuda_bind::*;
    use std::ffi::CStr;

    /// Needs an MI355X: the library has no CPU fallback.
    #[test]
    fn library_links_and_reports_its_target() {
        unsafe {
            assert_eq!(cuda_is_available(), 1);
            let v = CStr::from_ptr(hip_backend_version()).to_str().unwrap();
            assert!(v.contains("gfx950"));
            let stream = cuda_create_stream_ffi(0);
            let mut buf: *mut i8 = std::ptr::null_mut();
            // PARAM_MESSAGE_2_CARRY_2: scratch -> cleanup round trip through the reference's own entry points
            scratch_cuda_programmable_bootstrap_64_async(
                stream, 0, &mut buf, 918, 1, 2048, 1, 16, true, ffi_types_centered());
            cleanup_cuda_programmable_bootstrap_64(stream, 0, &mut buf);
            cuda_destroy_stream(stream, 0);
        }
    }
    fn ffi_types_centered() -> crate::ffi_types::PBS_MS_REDUCTION_T {
        crate::ffi_types::PBS_MS_REDUCTION_T_CENTERED
    }
}
