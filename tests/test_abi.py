"""CPU-side checks of the boundary: the C-ABI library builds/loads without a GPU and exports
every symbol include/plonky2_hip.h declares; no compute is launched here."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    if not os.path.exists(os.path.join(ROOT, "plonky2_gpu_amd", "libplonky2_hip.so")):
        g.build()
    import plonky2_gpu_amd as pg

    return pg.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "plonky2_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:GlError|void|int|uint64_t|const char)\s*\*?\s*(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_symbols_are_exported(lib):
    from plonky2_gpu_amd import _lib

    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/plonky2_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    # the reference's FFI names (cuda/src/lib.rs:58-145) are all present
    for n in ["init", "ifft", "build_merkle_tree", "merkle_tree_from_values", "merkle_tree_from_coeffs",
              "compute_quotient_polys", "cudaGetErrorString"]:
        assert n in names


def test_only_the_header_is_exported(lib):
    """Built with -fvisibility=hidden: the dynamic symbol table holds the header's functions and nothing else of ours
    (no C++ internals, no helper that happens to be non-static)."""
    import subprocess

    so = os.path.join(ROOT, "plonky2_gpu_amd", "libplonky2_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TtWwBbDdVv"}
    exported = {n for n in exported if not n.startswith(("__hip_", "_init", "_fini", "__bss", "_edata", "_end", "__odr", "__cxa"))}
    extra = sorted(exported - set(declared_symbols()))
    assert not extra, f"exported but not declared in include/plonky2_hip.h: {extra[:20]}"


def test_version_and_error_strings(lib):
    assert lib.gl_version().startswith(b"plonky2_hip")
    assert b"invalid" in lib.cudaGetErrorString(-1)
    assert b"unsupported" in lib.cudaGetErrorString(-2)


def test_product_never_imports_oracle():
    """③: only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "plonky2_gpu_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                # ("oracles" is also FRI's own word for committed polynomial batches, fri/oracle.rs)
                for needle in ("import oracle", "from oracle", "oracle/", "oracle.", "gl_oracle", "pyref", "_ref import", "fri_ref", "plonk_ref",
                               "libplonky2_ref", "ref_gpu", "ref_harness", "#include \"plonky2_gpu_impl"):
                    assert needle not in src.replace("oracle.rs", ""), (needle, os.path.join(dirpath, f))
    # nor do the built libraries know the checkers' names (the reference's kernels of oracle/_ref included)
    for so in ("libplonky2_hip.so", "libplonky2_hip_debug.so"):
        path = os.path.join(ROOT, "plonky2_gpu_amd", so)
        if os.path.exists(path):
            blob = open(path, "rb").read()
            for needle in (b"libplonky2_ref", b"libgl_oracle", b"ref_compute_quotient"):
                assert needle not in blob, (needle, so)


def test_reference_kernel_library_exports_its_entry_points():
    """oracle/_ref/libplonky2_ref.so (the reference's own kernels for gfx950, oracle/ref_harness.hip): every entry point the GPU
    tests call resolves. Built only where /root/reference is mounted; absent -> skipped with the reason."""
    from oracle import ref_gpu

    if not ref_gpu.build():
        pytest.skip(ref_gpu.why_absent())
    lib = ref_gpu.lib()
    for name in ref_gpu.SIGNATURES:
        assert getattr(lib, name) is not None
    assert b"no error" in lib.ref_error_string(0).lower() or lib.ref_error_string(0)


def _imports_of_oracle(src):
    return [ln.strip() for ln in src.splitlines() if ln.strip().startswith(("from oracle", "import oracle"))]


def test_oracle_is_used_only_where_it_may_be():
    """③ beyond the package: in bench.py only the cpu_baseline leg imports oracle/; nothing under tools/
    does (scripts that need the checker live under tests/); smoke() may."""
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # the cpu_baseline leg is a family of functions (the NTT, the commit, the prover's commitments, and the reference's own
    # kernels of oracle/_ref timed as a stated baseline): all named cpu_baseline*, all called from the one place in main() that
    # builds the baseline fields after the timed regions
    rest, legs = "", 0
    for piece in ("\n" + bench).split("\ndef "):
        if piece.startswith("cpu_baseline"):
            legs += 1
            assert _imports_of_oracle(piece), "a cpu_baseline leg times the oracle"
        else:
            rest += piece
    assert legs >= 1
    assert not _imports_of_oracle(rest), "bench.py uses oracle/ outside its cpu_baseline legs"
    assert "ref_gpu" not in rest and "libplonky2_ref" not in rest
    for dirpath, dirs, files in os.walk(os.path.join(ROOT, "tools")):
        dirs[:] = [d for d in dirs if not d.startswith("jitcache")]
        for f in files:
            if f.endswith(".py"):
                hits = _imports_of_oracle(open(os.path.join(dirpath, f)).read())
                assert not hits, (os.path.join(dirpath, f), hits)


def test_no_gpu_means_loud_failure(lib):
    import plonky2_gpu_amd as pg

    if lib.gl_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        pg.Context(0)


def test_the_product_library_has_no_ab_knobs():
    """The A/B switches of the kernels (csrc/knobs.h) are compiled into the diagnostic build only: the product library does not
    even contain their names, the diagnostic one does."""
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    product = open(os.path.join(root, "plonky2_gpu_amd", "libplonky2_hip.so"), "rb").read()
    for knob in (b"PLONKY2_NTT_DIRECT", b"PLONKY2_NTT_KERNEL", b"PLONKY2_NTT_WIDE", b"PLONKY2_NTT_XCD", b"PLONKY2_NTT_WG_PER_CU", b"PLONKY2_NTT_CHUNK_COLS",
                 b"PLONKY2_TRANSPOSE", b"PLONKY2_COMMIT_PIPELINE", b"PLONKY2_COMMIT_CHUNK", b"PLONKY2_FUSED_LEAVES", b"PLONKY2_POSEIDON", b"PLONKY2_DROP_STREAM2_WAIT",
                 b"PLONKY2_HIP_JIT_FUSE", b"PLONKY2_HIP_JIT_PEEPHOLE", b"PLONKY2_HIP_JIT_PREFETCH", b"PLONKY2_HIP_JIT_WAVES", b"PLONKY2_HIP_JIT_UNITS"):
        assert knob not in product, knob
    for setting in (b"PLONKY2_HIP_KERNEL_CACHE", b"PLONKY2_HIP_JIT_FORK"):  # operational settings, not knobs
        assert setting in product, setting
    debug = os.path.join(root, "plonky2_gpu_amd", "libplonky2_hip_debug.so")
    if os.path.exists(debug):
        assert b"PLONKY2_NTT_DIRECT" in open(debug, "rb").read() and b"PLONKY2_HIP_JIT_FUSE_GATES" in open(debug, "rb").read()
