//! Links libtfhe_hip_backend.so (built by `make -C tfhe_rs_amd/csrc`, hipcc --offload-arch=gfx950).
//! Counterpart of backends/tfhe-cuda-backend/build.rs:35-137, which runs cmake over cuda/ and bindgen over
//! cuda/include: here the bindings are checked in (src/bindings.rs, src/cuda_bind.rs, generated from
//! include/tfhe_hip_backend.h by tools/gen_rust_bindings.py) so no libclang is needed.
use std::path::PathBuf;
use std::process::Command;

fn main() {
    if std::env::var("DOCS_RS").map(|v| v == "1").unwrap_or(false) {
        return;
    }
    // repository root: TFHE_HIP_BACKEND_ROOT, or two levels up from this crate (backends/tfhe-hip-backend)
    let root = std::env::var("TFHE_HIP_BACKEND_ROOT")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("../.."));
    let csrc = root.join("tfhe_rs_amd/csrc");
    let lib = root.join("tfhe_rs_amd/lib");
    println!("cargo::rerun-if-changed={}", csrc.display());
    println!("cargo::rerun-if-changed={}", root.join("include/tfhe_hip_backend.h").display());
    println!("cargo::rerun-if-env-changed=TFHE_HIP_BACKEND_ROOT");
    if cfg!(feature = "build-native") || !lib.join("libtfhe_hip_backend.so").exists() {
        let status = Command::new("make")
            .arg("-C")
            .arg(&csrc)
            .arg("-j8")
            .arg("ARCH=gfx950")
            .status()
            .expect("failed to run make (hipcc from ROCm >= 7.0 is required)");
        assert!(status.success(), "building libtfhe_hip_backend.so failed");
    }
    println!("cargo:rustc-link-search=native={}", lib.display());
    println!("cargo:rustc-link-lib=dylib=tfhe_hip_backend");
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-lib=stdc++");
    // exported to dependents as DEP_TFHE_HIP_BACKEND_INCLUDE
    println!("cargo:include={}", root.join("include").display());
}
