// Links libaligner_hip.so (built by `python -m aligner_amd.build`).  ALIGNER_HIP_DIR: where it lives.
fn main() {
    if let Ok(dir) = std::env::var("ALIGNER_HIP_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=aligner_hip");
    println!("cargo:rerun-if-env-changed=ALIGNER_HIP_DIR");
}
