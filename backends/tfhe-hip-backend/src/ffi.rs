//! C scalar aliases, as backends/tfhe-cuda-backend/src/ffi.rs.
#![allow(warnings)]
pub type c_void = std::ffi::c_void;
pub type c_uint = std::ffi::c_uint;
pub type c_int = std::ffi::c_int;
pub type c_char = std::ffi::c_char;
