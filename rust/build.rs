// build.rs (feature = "gpu"): link the MI355X batch engine.
// libdecaf377_amd.so comes from `make lib` at the root of the engine's repository (hipcc, --offload-arch=gfx950; options
// FB_BITS, DCB_K, CHECK_INVARIANTS -- see its Makefile); DECAF377_AMD_LIB_DIR points at decaf377_amd/lib.
fn main() {
    if std::env::var("CARGO_FEATURE_GPU").is_ok() {
        if let Ok(dir) = std::env::var("DECAF377_AMD_LIB_DIR") {
            println!("cargo:rustc-link-search=native={dir}");
        }
        println!("cargo:rustc-link-lib=dylib=decaf377_amd");
    }
}
