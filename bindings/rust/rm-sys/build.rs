// Tells cargo where librm_hip.so lives: RM_HIP_LIB_DIR, or raymarching_amd/ of this repository.
fn main() {
    let dir = std::env::var("RM_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../../raymarching_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rm_hip");
    println!("cargo:rerun-if-env-changed=RM_HIP_LIB_DIR");
}
