// Points the linker at the in-tree libzeekstd_amd.so (built by `make -C zeekstd_amd/csrc`, gfx950 only).
fn main() {
    let root = std::path::Path::new(env!("CARGO_MANIFEST_DIR")).join("..").join("zeekstd_amd");
    let dir = std::env::var("ZEEKSTD_AMD_LIB_DIR").map(std::path::PathBuf::from).unwrap_or(root);
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=zeekstd_amd");
    println!("cargo:rerun-if-env-changed=ZEEKSTD_AMD_LIB_DIR");
}
