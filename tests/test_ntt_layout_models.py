"""The index / twiddle / LDS-bank models of the NTT passes (pure Python restatements of what the HIP kernels' lanes do,
checked against the definition of the transform): each script exits non-zero on a wrong value or a bank conflict."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("script", ["ntt_model.py", "ntt_direct_model.py", "ntt_direct_row_model.py", "ntt_direct_inplace_model.py"])
def test_model(script):
    r = subprocess.run([sys.executable, os.path.join(HERE, script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
