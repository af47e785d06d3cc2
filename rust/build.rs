// build.rs (feature = "gpu"): link the MI355X batch engine.
fn main() {
    if std::env::var("CARGO_FEATURE_GPU").is_ok() {
        if let Ok(dir) = std::env::var("DECAF377_AMD_LIB_DIR") {
            println!("cargo:rustc-link-search=native={dir}");
        }
        println!("cargo:rustc-link-lib=dylib=decaf377_amd");
    }
}
