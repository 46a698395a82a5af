// build.rs — link libperceive_hip.so (the MI355X-native library behind Model / Searcher).
// PERCEIVE_HIP_LIB_DIR points at the directory that holds it (the repository's perceive_amd/ after
// `make -C perceive_amd/csrc`); the run-time loader finds it through the rpath set here.
// NOT COMPILED in this repository's build image (no Rust toolchain): see README.md.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("PERCEIVE_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../perceive_amd"));
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=perceive_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=PERCEIVE_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
