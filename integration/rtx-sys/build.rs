// Links librtx.so.  RTX_LIB_DIR = directory that holds it (default: ../../raytracing_amd
// relative to this crate, where `make -C raytracing_amd/csrc` puts it).
fn main() {
    let dir = std::env::var("RTX_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{here}/../../raytracing_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rtx");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RTX_LIB_DIR");
}
