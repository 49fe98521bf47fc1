"""Shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np
import pytest

P = 0xFFFFFFFF00000001


@pytest.fixture(scope="session")
def gpu():
    import plonky2_gpu_amd as pg

    lib = pg.load()
    if lib.gl_device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (the product has no CPU fallback)")
    ctx = pg.Context(0)
    yield ctx
    ctx.close()


def bitrev_perm(log_n):
    n = 1 << log_n
    idx = np.arange(n, dtype=np.uint64)
    out = np.zeros(n, dtype=np.uint64)
    for b in range(log_n):
        out |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(log_n - 1 - b)
    return out.astype(np.int64)
