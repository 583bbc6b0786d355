// Links librtx_amd.so.  RTX_AMD_LIB_DIR names the directory that holds it (the package directory
// raytracing_rust_amd/ of the backend's repository); the HIP runtime it depends on is found through the library's own
// DT_NEEDED entries (/opt/rocm/lib must be on the loader's path).
fn main() {
    let dir = std::env::var("RTX_AMD_LIB_DIR").unwrap_or_else(|_| "../../../raytracing_rust_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtx_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RTX_AMD_LIB_DIR");
}
