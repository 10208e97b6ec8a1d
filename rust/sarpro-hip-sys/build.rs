// Link against libsarpro_hip.so.  SARPRO_HIP_LIB_DIR points at the directory that holds it
// (default: ../../sarpro_amd relative to this crate).
fn main() {
    let dir = std::env::var("SARPRO_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{here}/../../sarpro_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=sarpro_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=SARPRO_HIP_LIB_DIR");
}
