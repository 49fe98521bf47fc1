"""The C oracle (gl_oracle.c and the prover above the commit, prove_oracle.c) under AddressSanitizer and UndefinedBehaviorSanitizer (CPU only; the GPU pool has no sanitizer runs): the checker
everything is compared with must not itself read out of bounds or depend on undefined arithmetic."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan")
    lib = str(tmp_path / "libgl_oracle_san.so")
    subprocess.check_call([gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-fopenmp", "-shared", "-fPIC", os.path.join(ROOT, "oracle", "gl_oracle.c"), os.path.join(ROOT, "oracle", "prove_oracle.c"), "-o", lib])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "oracle_sanitizer_child.py"), lib], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "asan/ubsan run ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
