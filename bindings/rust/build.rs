// Links against tla_rust_amd/_build/libtlamc.so (build it with `python -m tla_rust_amd.build`).
fn main() {
    let dir = std::env::var("TLAMC_LIB_DIR").unwrap_or_else(|_| "../../tla_rust_amd/_build".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=tlamc");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
