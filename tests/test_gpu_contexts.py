"""Contexts on one device since round 6 (csrc/capi.hip CtxState): each owns its workspace, event pair and hashing stream, so that
several run at the same time; the reference's memory contract as an option (caller-provided workspace and staging buffer,
fri/oracle.rs:94-106: the caller sizes every device buffer up front); a public-inputs hash per context; contexts the caller
builds from its own streams (the reference's CudaInnerContext, fri/oracle.rs:43-47)."""
import ctypes
import os
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def gpu():
    import plonky2_gpu_amd as pg

    ctx = pg.Context(0)
    yield ctx
    ctx.close()


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    return hip


def _free_bytes(hip):
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


@pytest.mark.gpu
def test_a_second_context_is_not_queued_behind_the_first(gpu, oracle):
    """Up to round 5 every call took a per-device lock and made its stream wait for whatever the OTHER context had queued
    (capi.hip DeviceCall): a small transform on a second context finished when the first context's commit did. Now it
    finishes while that commit is still running — and both results are what they are alone."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    other = pg.Context(0)
    try:
        cols, log_n = 135, 18
        n, n_ext = 1 << log_n, 1 << (log_n + 3)
        vals = oracle.random_field((cols, n), seed=601)
        small = oracle.random_field((2, 1 << 12), seed=602)
        expect_small = oracle.canon(oracle.fft_batch(small))
        d_vals = pg.DeviceBuffer.from_host(gpu, vals)
        d_work, d_lde = pg.DeviceBuffer(gpu, cols * n), pg.DeviceBuffer(gpu, cols * n_ext)
        d_dig, d_cap = pg.DeviceBuffer(gpu, 8 * (n_ext - 16)), pg.DeviceBuffer(gpu, 64)

        def commit():
            _lib.call("gl_memcpy_d2d", d_work.ptr, d_vals.ptr, cols * n * 8, gpu.ptr)
            _lib.call("gl_commit_from_values", d_work.ptr, cols, log_n, 3, 4, 0, 7, d_lde.ptr, None, d_dig.ptr, d_cap.ptr, gpu.ptr)

        commit()
        gpu.synchronize()
        cap_alone = d_cap.download().copy()
        d_small = pg.DeviceBuffer(other, small.size)  # allocated up front: hipMalloc / hipFree wait for the whole device

        def small_transform():  # upload, transform, download on the OTHER context: synchronises that one only
            d_small.upload(small)
            _lib.call("gl_ntt_batch", d_small.ptr, 2, 12, 1 << 12, 0, 0, other.ptr)
            return d_small.download().reshape(small.shape)

        assert (small_transform() == expect_small).all()  # warm-up of the second context
        ratios = []
        for _ in range(5):
            gpu.synchronize()
            t0 = time.perf_counter()
            commit()  # asynchronous: returns when the launches are queued
            t_queued = time.perf_counter()
            got = small_transform()
            t_small = time.perf_counter()
            gpu.synchronize()
            t_commit = time.perf_counter()
            assert (got == expect_small).all()
            assert (d_cap.download() == cap_alone).all()
            ratios.append((t_small - t_queued) / (t_commit - t0))
        # the commit takes ~8 ms here, the small transform a fraction of a millisecond; queued behind the commit it would take as long
        assert min(ratios) < 0.5, ratios
        for b in (d_vals, d_work, d_lde, d_dig, d_cap, d_small):
            b.free()
    finally:
        other.close()


@pytest.mark.gpu
def test_proofs_of_one_circuit_handle_on_two_contexts_at_once(gpu):
    """One gl_circuit handle, two host threads, each proving on its own context: the handle keeps a buffer pool per context and
    its compiled gate kernel's launches take turns (gate_jit.hip GateKernel::launch_mu); every proof equals the one made alone."""
    import plonky2_gpu_amd as pg
    from plonk_instance import make_circuit

    circuit, wires, pis = make_circuit(8, seed=611, two_groups=True)
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))
    other = pg.Context(0)
    try:
        flat = np.ascontiguousarray(np.array(wires, dtype=np.uint64).reshape(-1))
        bufs = {gpu: pg.DeviceBuffer.from_host(gpu, flat), other: pg.DeviceBuffer.from_host(other, flat)}
        expect = nc.prove_bytes(bufs[gpu], pis)
        got = {gpu: [], other: []}

        def work(ctx):
            for _ in range(12):
                got[ctx].append(nc.prove_bytes(bufs[ctx], pis, ctx=ctx))

        threads = [threading.Thread(target=work, args=(c,)) for c in (gpu, other)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for c in (gpu, other):
            assert len(got[c]) == 12 and all(g == expect for g in got[c])
        for b in bufs.values():
            b.free()
    finally:
        nc.close()
        other.close()


@pytest.mark.gpu
def test_caller_provided_workspace(gpu, oracle):
    """gl_ctx_set_workspace: the natural-order transforms of a context stage through the CALLER'S buffer (gl_workspace_bytes()
    of it), the library's own 512 MiB go back to the device, and NULL brings a library-owned workspace back."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    lib = _lib.load()
    need = lib.gl_workspace_bytes()
    assert need == 1 << 29
    hip = _hip()
    ctx = pg.Context(0)
    try:
        x = oracle.random_field((3, 1 << 16), seed=621)
        expect = oracle.canon(oracle.fft_batch(x))
        assert (pg.fft_with_options(ctx, x) == expect).all()
        ctx.synchronize()
        mine = pg.DeviceBuffer(ctx, need // 8)
        before = _free_bytes(hip)
        _lib.call("gl_ctx_set_workspace", ctx.ptr, mine.ptr, need)
        assert _free_bytes(hip) >= before + need - (8 << 20), "the library's own workspace was not given back"
        _lib.call("gl_memset_zero", mine.ptr, 3 * (1 << 16) * 8, ctx.ptr)
        assert (pg.fft_with_options(ctx, x) == expect).all()
        assert (pg.ifft_with_options(ctx, expect) == oracle.canon(x)).all()
        assert mine.download(0, 1 << 16).any(), "the transform did not stage through the caller's workspace"
        with pytest.raises(_lib.Plonky2HipError):
            _lib.call("gl_ctx_set_workspace", ctx.ptr, mine.ptr, need - 8)  # too small
        _lib.call("gl_ctx_set_workspace", ctx.ptr, None, 0)
        mine.free()
        assert (pg.fft_with_options(ctx, x) == expect).all()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_context_built_by_the_caller_and_released(oracle):
    """The reference's host builds {stream, stream2} itself (CudaInnerContext, fri/oracle.rs:43-47, through rustacuda): the library
    attaches its state on the first call that sees the pair and gl_ctx_release gives the 512 MiB back before the caller destroys
    its streams."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    hip = _hip()
    assert hip.hipSetDevice(0) == 0
    streams = (ctypes.c_void_p * 2)()
    for i in range(2):
        assert hip.hipStreamCreateWithFlags(ctypes.byref(streams, i * ctypes.sizeof(ctypes.c_void_p)), 1) == 0  # hipStreamNonBlocking

    class Raw:  # what DeviceBuffer and the operator mirror need of a context
        ptr = ctypes.addressof(streams)
        device = 0

        def synchronize(self):
            _lib.call("gl_ctx_synchronize", self.ptr)

    ctx = Raw()
    before = _free_bytes(hip)
    x = oracle.random_field((2, 1 << 14), seed=631)
    assert (pg.fft_with_options(ctx, x) == oracle.canon(oracle.fft_batch(x))).all()
    assert _free_bytes(hip) <= before - (1 << 29) + (8 << 20), "no workspace was attached to the caller's context"
    vals = oracle.random_field((9, 1 << 8), seed=632)
    batch = pg.PolynomialBatch.from_values(ctx, vals, 3, False, 2)
    exp = oracle.commit_from_values(vals, 3, 2, threads=2)
    assert (batch.merkle_tree.cap == oracle.canon(exp["cap"])).all()
    del batch
    _lib.load().gl_ctx_release(ctx.ptr)
    assert _free_bytes(hip) >= before - (8 << 20), "gl_ctx_release did not give the context's state back"
    _lib.load().gl_ctx_release(ctx.ptr)  # a second release is a no-op
    for i in range(2):
        assert hip.hipStreamDestroy(ctypes.c_void_p(streams[i])) == 0


@pytest.mark.gpu
def test_reference_quotient_with_the_callers_staging_and_a_hash_per_context(gpu):
    """compute_quotient_polys (the reference symbol) under the reference's memory contract: the caller hands in the staging buffer
    (gl_reference_quotient_staging_bytes), the library allocates nothing; and two contexts proving two different proofs of the
    circuit each set their own public-inputs hash (gl_reference_set_public_inputs_hash_ctx) without racing on a process-wide one."""
    import random

    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib
    from test_reference_quotient import oracle_quotient, random_instance, run_symbol

    lib = _lib.load()
    log_len = 4
    inst = random_instance(log_len, seed=641)
    _lib.call("gl_reference_quotient_release")
    _lib.call("gl_reference_quotient_prepare", gpu.ptr)
    need = lib.gl_reference_quotient_staging_bytes(log_len)
    assert need == 8 * (234 + 88 + 20) * ((1 << log_len) << 3)
    stage = pg.DeviceBuffer(gpu, need // 8)
    _lib.call("gl_memset_zero", stage.ptr, need, gpu.ptr)
    gpu.synchronize()
    other = pg.Context(0)
    try:
        _lib.call("gl_reference_quotient_set_staging", stage.ptr, need)
        h1 = [random.Random(7).randrange(P) for _ in range(4)]
        h2 = [random.Random(8).randrange(P) for _ in range(4)]
        _lib.call("gl_reference_set_public_inputs_hash_ctx", np.array(h1, dtype=np.uint64), gpu.ptr)
        _lib.call("gl_reference_set_public_inputs_hash_ctx", np.array(h2, dtype=np.uint64), other.ptr)
        got = {}

        def work(ctx, key):
            for _ in range(4):
                out, bufs = run_symbol(ctx, inst)
                got.setdefault(key, []).append(out)
                for b in bufs.values():
                    b.free()

        threads = [threading.Thread(target=work, args=(gpu, 1)), threading.Thread(target=work, args=(other, 2))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        gpu.synchronize()
        other.synchronize()
        assert stage.download().any(), "the caller's staging buffer was not used"  # (it was zeroed above)
        e1 = np.array(oracle_quotient(inst, h1), dtype=np.uint64)
        e2 = np.array(oracle_quotient(inst, h2), dtype=np.uint64)
        assert (e1 != e2).any()
        assert all((g == e1).all() for g in got[1]) and all((g == e2).all() for g in got[2])
        # a buffer too small for the call is not used (the rows are read in place): same result
        _lib.call("gl_reference_quotient_set_staging", stage.ptr, need - 16)
        out, bufs = run_symbol(gpu, inst)
        assert (out == e1).all()
    finally:
        _lib.call("gl_reference_quotient_set_staging", None, 0)
        _lib.call("gl_reference_set_public_inputs_hash_ctx", None, gpu.ptr)
        other.close()
        stage.free()


@pytest.mark.gpu
def test_prove_many_keeps_several_proofs_in_flight_and_every_proof_is_the_one_made_alone(gpu):
    """gl_prove_many: seven witnesses of one circuit, three contexts — worker w proves witnesses w, w + 3, .. on its context, the workers
    share the circuit handle — and every proof equals gl_prove's for that witness alone; different public inputs per proof; a bad
    argument is refused and leaves nothing behind."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib
    from plonk_instance import make_circuit

    circuit, wires, pis = make_circuit(8, seed=651, two_groups=True)
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))
    others = [pg.Context(0), pg.Context(0)]
    try:
        base = np.array(wires, dtype=np.uint64)
        witnesses, alone = [], []
        for i in range(7):
            # the same satisfying witness every time (the transcript, and with it every byte of the proof, still differs through the
            # public inputs' hash only if the circuit reads them; here the proofs are equal and that is fine: what is compared is each
            # proof against the one made alone from the same buffer)
            buf = pg.DeviceBuffer.from_host(gpu, np.ascontiguousarray(base.reshape(-1)))
            witnesses.append(buf)
            alone.append(nc.prove_bytes(buf, pis))
        got = nc.prove_many(witnesses, [pis] * 7, [gpu] + others)
        assert got == alone
        assert nc.prove_many(witnesses[:2], [pis] * 2, [gpu]) == alone[:2]   # one in flight: plain gl_prove in a loop
        assert nc.prove_many([], [], [gpu]) == []
        with pytest.raises(_lib.Plonky2HipError):
            nc.prove_many(witnesses, [pis] * 7, [gpu, gpu])   # two workers on one context
        for b in witnesses:
            b.free()
    finally:
        nc.close()
        for c in others:
            c.close()
