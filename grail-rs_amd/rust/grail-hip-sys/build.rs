fn main() {
    // libgrail_hip.so is built by `make -C grail-rs_amd` (hipcc, gfx950)
    let dir = std::env::var("GRAIL_HIP_LIB_DIR").unwrap_or_else(|_| "../../lib".into());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=grail_hip");
    println!("cargo:rerun-if-env-changed=GRAIL_HIP_LIB_DIR");
}
