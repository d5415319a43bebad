// Tells cargo where libtopo_hip.so lives: TOPO_HIP_LIB_DIR, or the in-tree build directory of this repository
// (`make -C topo-renderer_amd/csrc` puts it in topo-renderer_amd/).  Source only: never run in the image this repository is
// built in (no cargo there).
fn main() {
    let dir = std::env::var("TOPO_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../topo-renderer_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=topo_hip");
    println!("cargo:rerun-if-env-changed=TOPO_HIP_LIB_DIR");
}
