// Links libvello_hip.so (built by `make -C vello_amd/csrc`, or python -c 'import __graft_entry__ as g; g.build()').
// VELLO_HIP_LIB_DIR overrides the in-tree location.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("VELLO_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../vello_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=vello_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=VELLO_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/vello_hip.h");
}
