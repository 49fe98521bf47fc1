"""The Rust FFI crate (ffi/plonky2_hip_sys) cannot be compiled in this image (no cargo/rustc). These checks keep it
honest mechanically: its generated half equals what the generator makes of the header today; every `pub fn` in an
extern "C" block is a symbol the shared library exports; the seven reference functions take the header's argument
counts; the link directives name the libraries the .so needs."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "ffi", "plonky2_hip_sys")


def _extern_fns(path):
    src = open(path).read()
    fns = {}
    for block in re.findall(r'extern "C" \{(.*?)\n\}', src, flags=re.S):
        for name, args in re.findall(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->[^;]*)?;", block, flags=re.S):
            args = re.sub(r"///.*", "", args)
            fns[name] = len([a for a in args.split(",") if ":" in a])
    return fns


def _header_fns():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "plonky2_hip.h")).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"^\s*(?:GlError|void|int|const char)\s*\*?\s*(\w+)\s*\(([^;]*)\);", text, flags=re.M):
        args = " ".join(m.group(2).split())
        out[m.group(1)] = 0 if args == "void" else len(args.split(","))
    return out


def test_generated_bindings_are_current():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ffi_rs.py"), "--check"]).returncode == 0, \
        "ffi/plonky2_hip_sys/src/bindings.rs is stale: run python tools/gen_ffi_rs.py"


def test_every_extern_is_exported_and_matches_the_header():
    import plonky2_gpu_amd as pg

    lib = pg.load()
    hdr = _header_fns()
    fns = {}
    fns.update(_extern_fns(os.path.join(CRATE, "src", "lib.rs")))
    fns.update(_extern_fns(os.path.join(CRATE, "src", "bindings.rs")))
    fns.pop("free", None)  # libc
    assert len(fns) >= 50
    for name, nargs in fns.items():
        assert hasattr(lib, name), f"{name} is declared in the crate but not exported by libplonky2_hip.so"
        assert hdr[name] == nargs, f"{name}: {nargs} arguments in Rust, {hdr[name]} in include/plonky2_hip.h"
    assert set(hdr) == set(fns), sorted(set(hdr) ^ set(fns))
    # the reference's names (cuda/src/lib.rs:58-145) are the hand-written ones
    ref = _extern_fns(os.path.join(CRATE, "src", "lib.rs"))
    assert {"init", "ifft", "build_merkle_tree", "merkle_tree_from_values", "merkle_tree_from_coeffs", "compute_quotient_polys",
            "cudaGetErrorString"} <= set(ref)


def test_build_script_links_what_the_library_needs():
    build = open(os.path.join(CRATE, "build.rs")).read()
    for lib in ("plonky2_hip", "amdhip64", "hiprtc"):
        assert f"rustc-link-lib=dylib={lib}" in build
    so = os.path.join(ROOT, "plonky2_gpu_amd", "libplonky2_hip.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "libamdhip64" in needed and "libhiprtc" in needed
    cargo = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert 'name = "plonky2_cuda"' in cargo  # `use plonky2_cuda::...` in the reference's sources keeps compiling
