// build.rs of the shim crate (or of bevy_firework itself, behind the `hip_backend` feature): link libfirework_hip.so.
// UNVERIFIED SOURCE (no Rust toolchain in the image this backend was built in).
fn main() {
    if std::env::var("CARGO_FEATURE_HIP_BACKEND").is_ok() {
        // the directory that holds libfirework_hip.so (bevy_firework_amd/csrc after `make`)
        let dir = std::env::var("FIREWORK_HIP_LIB_DIR").expect("set FIREWORK_HIP_LIB_DIR to the directory of libfirework_hip.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=firework_hip");
        println!("cargo:rerun-if-env-changed=FIREWORK_HIP_LIB_DIR");
    }
}
