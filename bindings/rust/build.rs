// Links the prebuilt library (`make -C zlib_amd/csrc`, i.e. hipcc --offload-arch=gfx950).
// NOT compiled in the image this repository was developed in (it has no rustc/cargo): written as source for a
// machine that has both.
fn main() {
    let dir = std::env::var("ZLIB_AMD_LIB_DIR").expect("set ZLIB_AMD_LIB_DIR to the directory of libzlib_amd.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zlib_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ZLIB_AMD_LIB_DIR");
}
