// Links librt_amd.so.  RT_AMD_LIB_DIR names the directory that holds it (raytracer_amd/csrc of the backend's
// repository after `make`); the library needs the HIP runtime of a ROCm installation at run time.
fn main() {
    if let Ok(dir) = std::env::var("RT_AMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rerun-if-env-changed=RT_AMD_LIB_DIR");
}
