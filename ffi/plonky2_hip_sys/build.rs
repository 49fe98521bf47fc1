// Links libplonky2_hip.so (+ the HIP runtime and hiprtc it depends on). Replaces the reference's cuda/build.rs:19-44,
// which compiled plonky2_gpu.cu with nvcc for sm_75: here the device code is a shared library built by
// `make -C plonky2_gpu_amd/csrc` (hipcc --offload-arch=gfx950), because it is also what the Python and C++ hosts load.
//
// Search order for the library directory:
//   1. $PLONKY2_HIP_LIB_DIR (a directory holding a prebuilt libplonky2_hip.so),
//   2. <repo>/plonky2_gpu_amd/ — built on the spot with make + hipcc when the file is missing and the
//      `build-from-source` feature is on (hipcc is looked up as $HIPCC, then /opt/rocm/bin/hipcc, then on $PATH).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let repo = manifest.join("../..").canonicalize().expect("repo root");
    let lib_dir = match env::var("PLONKY2_HIP_LIB_DIR") {
        Ok(dir) => PathBuf::from(dir),
        Err(_) => repo.join("plonky2_gpu_amd"),
    };
    let so = lib_dir.join("libplonky2_hip.so");
    if !so.exists() {
        if env::var("CARGO_FEATURE_BUILD_FROM_SOURCE").is_err() {
            panic!("{} not found and the build-from-source feature is off; set PLONKY2_HIP_LIB_DIR", so.display());
        }
        let hipcc = env::var("HIPCC").unwrap_or_else(|_| {
            if PathBuf::from("/opt/rocm/bin/hipcc").exists() { "/opt/rocm/bin/hipcc".into() } else { "hipcc".into() }
        });
        let status = Command::new("make")
            .arg("-C")
            .arg(repo.join("plonky2_gpu_amd/csrc"))
            .arg(format!("HIPCC={hipcc}"))
            .arg("-j4")
            .status()
            .expect("running make (is it installed?)");
        assert!(status.success(), "building libplonky2_hip.so failed: hipcc for gfx950 is needed (ROCm >= 7)");
    }
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=plonky2_hip");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-lib=dylib=hiprtc");
    // let the test / prover binaries find the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{rocm}/lib");
    println!("cargo:rerun-if-env-changed=PLONKY2_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=HIPCC");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    println!("cargo:rerun-if-changed={}", repo.join("plonky2_gpu_amd/csrc").display());
    println!("cargo:rerun-if-changed={}", repo.join("include/plonky2_hip.h").display());
    // The reference's build script prints the same line (cuda/build.rs:38). A build-script cfg applies to THIS crate only:
    // nothing in plonky2's own sources is switched by it (they contain no cfg(feature = "cuda"); the GPU entry points
    // from_values_with_gpu / my_prove are always compiled), so it is kept only for parity with the crate it replaces.
    println!("cargo:rustc-cfg=feature=\"cuda\"");
}
