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
    for m in re.finditer(r"^\s*(?:GlError|void|int|uint64_t|const char)\s*\*?\s*(\w+)\s*\(([^;]*)\);", text, flags=re.M):
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


# ---------------------------------------------------------------------------------------------------------------------
# ffi/rustacuda_hip (rustacuda's API subset on the HIP runtime) and ffi/patches/*.patch
SHIM = os.path.join(ROOT, "ffi", "rustacuda_hip")
REFERENCE = "/root/reference"


def test_rustacuda_shim_binds_existing_hip_runtime_symbols():
    """every function the shim declares extern "C" is exported by libamdhip64 with that name"""
    fns = _extern_fns_any(os.path.join(SHIM, "src", "lib.rs"))
    assert {"hipStreamCreateWithFlags", "hipStreamSynchronize", "hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree", "hipMemcpy",
            "hipMemcpyAsync"} <= set(fns)
    hip = "/opt/rocm/lib/libamdhip64.so"
    if not os.path.exists(hip):
        import pytest

        pytest.skip("no ROCm runtime in this image")
    exported = subprocess.run(["nm", "-D", "--defined-only", hip], capture_output=True, text=True).stdout
    for name in fns:
        assert re.search(rf"\b{name}\b", exported), f"{name} is not exported by libamdhip64"


def _extern_fns_any(path):
    src = open(path).read()
    out = {}
    for block in re.findall(r'extern "C" \{(.*?)\n\}', src, flags=re.S):
        for name in re.findall(r"\bfn (\w+)\s*\(", block):
            out[name] = True
    return out


# what sideprotocol/plonky2-gpu's Rust imports from rustacuda (fri/oracle.rs:37-38, 55, 67; plonk/prover.rs:37-39;
# fri/prover.rs:6; hash/merkle_tree.rs:9; field/src/goldilocks_field.rs:8) and calls on those types
RUSTACUDA_NAMES_USED = ["pub mod prelude", "pub mod memory", "pub mod stream", "pub struct Stream", "pub struct DeviceBuffer", "pub struct DeviceSlice",
                        "pub trait AsyncCopyDestination", "pub trait CopyDestination", "pub unsafe fn cuda_malloc_locked", "pub unsafe fn cuda_free_locked",
                        "pub use rustacuda_core::{DeviceCopy", "pub struct Context", "fn synchronize", "fn split_at_mut", "fn as_mut_ptr", "fn as_ptr",
                        "unsafe fn async_copy_from", "unsafe fn async_copy_to", "fn len(", "#[repr(transparent)]"]


def test_rustacuda_shim_defines_what_the_reference_uses():
    src = open(os.path.join(SHIM, "src", "lib.rs")).read()
    for name in RUSTACUDA_NAMES_USED:
        assert name in src, name
    # the reference's own import lines, where the tree is mounted: every imported item is defined by the shim
    if os.path.isdir(REFERENCE):
        for rel in ("plonky2/src/fri/oracle.rs", "plonky2/src/plonk/prover.rs", "plonky2/src/fri/prover.rs", "plonky2/src/hash/merkle_tree.rs",
                    "field/src/goldilocks_field.rs"):
            for line in open(os.path.join(REFERENCE, rel)):
                m = re.match(r"\s*use rustacuda::([\w:]+)(?:::\{([^}]*)\}|::(\*))?;", line)
                if not m:
                    continue
                items = [i.strip() for i in (m.group(2) or "").split(",") if i.strip()] or ([] if m.group(3) else [m.group(1).split("::")[-1]])
                for item in items:
                    assert re.search(rf"\b(struct|trait|fn|mod|use)\b[^\n]*\b{item}\b", src), f"{rel}: rustacuda item {item} is not in the shim"
    cargo = open(os.path.join(SHIM, "Cargo.toml")).read()
    assert 'name = "rustacuda"' in cargo  # `use rustacuda::...` keeps compiling: the dependency NAME stays


def test_patches_apply_to_the_reference_tree():
    """ffi/patches/*.patch are diff hunks against sideprotocol/plonky2-gpu; where the tree is mounted they must apply cleanly
    (dry run: the mount is read-only), and they must be hunks, not files."""
    import shutil

    import pytest

    patches = sorted(f for f in os.listdir(os.path.join(ROOT, "ffi", "patches")) if f.endswith(".patch"))
    assert patches == ["plonky2-hip-dumps.patch", "plonky2-hip.patch"]
    for f in patches:
        text = open(os.path.join(ROOT, "ffi", "patches", f)).read()
        assert text.count("\n@@ ") >= 2 and "\n--- a/" in text and "\n+++ b/" in text
        context = sum(1 for line in text.split("\n") if line.startswith(" "))
        assert context <= 60, "a patch carries hunks with three lines of context, never a file"
    if not os.path.isdir(REFERENCE) or not shutil.which("patch"):
        pytest.skip("the reference tree is not mounted here")
    for f in patches:
        r = subprocess.run(["patch", "--dry-run", "-p1", "-d", REFERENCE, "-i", os.path.join(ROOT, "ffi", "patches", f)], capture_output=True, text=True)
        assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr
