// Links libptau_hip.so.  PTAU_HIP_LIB_DIR = the directory that holds it (snark-setup-operator_amd/ of this repository after
// `python -c 'import __graft_entry__ as g; g.build()'`, or wherever the operator's packaging installs it).
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=PTAU_HIP_LIB_DIR");
    let dir = env::var("PTAU_HIP_LIB_DIR").unwrap_or_else(|_| "../snark-setup-operator_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=ptau_hip");
    // the library is self-contained apart from the HIP runtime it was linked against
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
