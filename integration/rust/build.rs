// build.rs for the reference crate: compiles the HIP render-loop library with
// hipcc and links it.  Set RT_HIP_SRC to the checkout of this repository
// (the directory that holds include/rt_hip.h); ROCM_PATH defaults to /opt/rocm.
//
// Cargo.toml additions:
//     [package]
//     build = "build.rs"
// No new crate dependencies: the binding below is a plain `extern "C"` block.

use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let src = PathBuf::from(env::var("RT_HIP_SRC").expect("RT_HIP_SRC = path of the rt_hip checkout"));
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let csrc = src.join("raytracing_amd").join("csrc");
    let lib = out.join("librt_hip.so");

    let sources = ["rt_kernels.hip", "rt_api.cpp", "rt_scene_build.cpp", "rt_host.cpp"];
    let mut cmd = Command::new(format!("{}/bin/hipcc", rocm));
    cmd.args(&[
        "-shared", "-fPIC", "-O3", "-std=c++17", "--offload-arch=gfx950",
        // the reference's arithmetic: binary32, no contraction, IEEE div/sqrt
        "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
        "-fhip-fp32-correctly-rounded-divide-sqrt",
    ]);
    cmd.arg(format!("-I{}", src.join("include").display()));
    for s in sources.iter() {
        let p = csrc.join(s);
        println!("cargo:rerun-if-changed={}", p.display());
        cmd.arg(p);
    }
    cmd.arg("-o").arg(&lib);
    let status = cmd.status().expect("hipcc not found");
    assert!(status.success(), "hipcc failed");

    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    println!("cargo:rerun-if-env-changed=RT_HIP_SRC");
}
