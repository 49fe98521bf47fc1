"""Device FRI opening pipeline vs oracle/fri_ref.py (itself pinned by prove -> verify): the proof
object is identical element for element, and the oracle's verifier accepts it."""
import numpy as np
import pytest

from fri_instance import make_fri_instance, transcript_before_fri
from gpu_util import gpu  # noqa: F401
from oracle import fri_ref, pyref

pytestmark = pytest.mark.gpu
P = pyref.P


def device_oracles(gpu, oracles, params):
    import plonky2_gpu_amd as pg

    out = []
    for o in oracles:
        b = pg.PolynomialBatch.from_values(gpu, np.array(o["values"], dtype=np.uint64), params["rate_bits"], False, params["cap_height"])
        assert b.merkle_tree.cap.tolist() == o["cap"]
        out.append(b)
    return out


def device_transcript(gpu, oracles, openings):
    import plonky2_gpu_amd as pg

    ch = pg.Challenger(gpu)
    for o in oracles:
        ch.observe_cap(o["cap"])
    for vals in openings:
        ch.observe_extension_elements(vals)
    return ch


def test_challenger_matches_oracle(gpu):
    import plonky2_gpu_amd as pg

    a, b = pg.Challenger(gpu), fri_ref.Challenger()
    for c in (a, b):
        c.observe_elements([1, 2, 3])
    assert a.get_n_challenges(3) == b.get_n_challenges(3)
    for c in (a, b):
        c.observe_elements(list(range(20)))
        c.observe_extension_elements([(5, 6)])
    assert a.get_extension_challenge() == b.get_extension_challenge()
    assert a.get_n_challenges(9) == b.get_n_challenges(9)


@pytest.mark.parametrize("degree_bits,arity_bits,polys", [(4, (2, 1), (3, 2)), (5, (3,), (3, 2)), (4, (1, 1, 1), (2, 1)), (3, (), (3, 2)),
                                                          (10, (4, 3), (5, 3))])
def test_prove_openings_equals_oracle_and_verifies(gpu, degree_bits, arity_bits, polys):
    import plonky2_gpu_amd as pg

    oracles, instance, params, openings = make_fri_instance(degree_bits=degree_bits, arity_bits=arity_bits, polys_per_oracle=polys,
                                                            seed=degree_bits + len(arity_bits))
    batches = device_oracles(gpu, oracles, params)
    # openings computed on the device equal the oracle's
    for (pt, sel), exp in zip(instance["batches"], openings):
        for (oi, pi), e in zip(sel, exp):
            got = batches[oi].eval_polynomials_ext2([pt])[0, pi]
            assert (int(got[0]), int(got[1])) == e
    proof = pg.prove_openings(gpu, instance, batches, device_transcript(gpu, oracles, openings), params)
    exp = fri_ref.prove_openings(instance, oracles, transcript_before_fri(oracles, openings), params)
    assert proof["pow_witness"] == exp["pow_witness"]
    assert proof["final_poly"] == exp["final_poly"]
    assert proof["commit_phase_merkle_caps"] == exp["commit_phase_merkle_caps"]
    assert proof["query_round_proofs"] == exp["query_round_proofs"]
    chal = fri_ref.fri_challenges(transcript_before_fri(oracles, openings), proof, degree_bits, params)
    assert fri_ref.verify_fri_proof(instance, openings, chal, [o["cap"] for o in oracles], proof, degree_bits, params)


def test_proof_of_work_is_the_smallest_witness(gpu):
    import ctypes

    from plonky2_gpu_amd import _lib

    state = np.arange(100, 112, dtype=np.uint64)
    for bits in (6, 12):
        min_lz = bits + (64 - P.bit_length())
        w = ctypes.c_uint64()
        _lib.call("gl_fri_proof_of_work", state, 3, min_lz, ctypes.addressof(w), gpu.ptr)
        def ok(c):
            s = [int(v) for v in state]
            s[3] = c
            r = pyref.poseidon(s)[7]
            return 64 - r.bit_length() >= min_lz
        assert ok(w.value)
        lo = max(0, w.value - 300)
        assert not any(ok(c) for c in range(lo, w.value))


def _rand_ext(rng, n):
    return [(rng.randrange(P), rng.randrange(P)) for _ in range(n)]


def _planar(ext):
    return np.array([[e[0] for e in ext], [e[1] for e in ext]], dtype=np.uint64).reshape(-1)


def _unplanar(a):
    a = a.reshape(2, -1)
    return [(int(x), int(y)) for x, y in zip(a[0], a[1])]


@pytest.mark.parametrize("log_n,m", [(0, 1), (3, 1), (4, 5), (9, 3), (12, 37), (15, 2), (18, 3)])
def test_reduce_polys_base(gpu, log_n, m):
    import random

    from plonky2_gpu_amd import _lib
    from plonky2_gpu_amd.device import DeviceBuffer

    rng = random.Random(log_n * 100 + m)
    n = 1 << log_n
    polys = [[rng.randrange(P) for _ in range(n)] for _ in range(m)]
    alpha = (rng.randrange(P), rng.randrange(P))
    d_polys = DeviceBuffer.from_host(gpu, np.array(polys, dtype=np.uint64).reshape(-1))
    ptrs = np.array([d_polys.ptr + 8 * n * j for j in range(m)], dtype=np.uint64)
    d_ptrs = DeviceBuffer.from_host(gpu, ptrs)
    d_out = DeviceBuffer(gpu, 2 * n)
    _lib.call("gl_fri_reduce_polys_base", d_ptrs.ptr, m, n, np.array(alpha, dtype=np.uint64), d_out.ptr, gpu.ptr)
    assert _unplanar(d_out.download()) == fri_ref.reduce_polys_base(polys, alpha)


@pytest.mark.parametrize("log_n", [1, 2, 5, 8, 9, 13, 16, 18, 20])  # 2^18: the ed25519 proof; 2^20: config #3's degree
def test_divide_by_linear_accumulate(gpu, log_n):
    import random

    from plonky2_gpu_amd import _lib
    from plonky2_gpu_amd.device import DeviceBuffer

    rng = random.Random(log_n)
    n = 1 << log_n
    comp1, comp2 = _rand_ext(rng, n), _rand_ext(rng, n)
    z1, z2, scale = _rand_ext(rng, 3)
    d_final = DeviceBuffer(gpu, 2 * n)
    for comp, z, acc in ((comp1, z1, 0), (comp2, z2, 1)):
        d_comp = DeviceBuffer.from_host(gpu, _planar(comp))
        _lib.call("gl_fri_divide_by_linear", d_comp.ptr, n, np.array(z, dtype=np.uint64),
                  np.array(scale, dtype=np.uint64), acc, d_final.ptr, gpu.ptr)
    q1, q2 = fri_ref.divide_by_linear(comp1, z1), fri_ref.divide_by_linear(comp2, z2)
    exp = [(0, 0)] + [fri_ref.ext_add(fri_ref.ext_mul(a, scale), b) for a, b in zip(q1, q2)]
    assert _unplanar(d_final.download()) == exp


@pytest.mark.parametrize("log_len,ab", [(1, 1), (4, 2), (6, 3), (10, 4), (14, 1), (14, 4), (18, 4), (19, 2)])
def test_fold_and_interleave(gpu, log_len, ab):
    import random

    from plonky2_gpu_amd import _lib
    from plonky2_gpu_amd.device import DeviceBuffer

    rng = random.Random(log_len + ab)
    n = 1 << log_len
    coeffs = _rand_ext(rng, n)
    beta = (rng.randrange(P), rng.randrange(P))
    d_c = DeviceBuffer.from_host(gpu, _planar(coeffs))
    d_o = DeviceBuffer(gpu, 2 * (n >> ab))
    _lib.call("gl_fri_fold", d_c.ptr, n, ab, np.array(beta, dtype=np.uint64), d_o.ptr, gpu.ptr)
    assert _unplanar(d_o.download()) == [fri_ref.reduce_with_powers_ext(coeffs[k : k + (1 << ab)], beta) for k in range(0, n, 1 << ab)]
    d_r = DeviceBuffer(gpu, 2 * n)
    _lib.call("gl_ext2_interleave", d_c.ptr, n, d_r.ptr, gpu.ptr)
    assert d_r.download().tolist() == fri_ref.flatten(coeffs)


# ---- the device-resident Challenger and the entry points that read its outputs from device memory (round 6) ----

def _step(pg, gpu, d_ch, srcs, n_out, flags=0):
    """gl_challenger_step; srcs = [(DeviceBuffer or address, count, planar_len)]; returns the n_out challenges (or the 4-word hash)"""
    import ctypes

    import numpy as np
    from plonky2_gpu_amd import _lib

    arr = (_lib.GlObserveSrc * max(1, len(srcs)))()
    for i, (p, count, planar) in enumerate(srcs):
        arr[i] = _lib.GlObserveSrc(p if isinstance(p, int) else p.ptr, count, planar)
    words = 4 if flags & 2 else n_out
    d_out = pg.DeviceBuffer(gpu, max(words, 1))
    _lib.call("gl_challenger_step", d_ch.ptr, ctypes.addressof(arr), len(srcs), n_out, d_out.ptr if words else None, flags, gpu.ptr)
    out = d_out.download()[:words].tolist() if words else []
    d_out.free()
    return [int(v) for v in out]


@pytest.mark.gpu
def test_device_challenger_equals_the_reference_transcript(gpu):
    """gl_challenger_step against the oracle's Challenger (iop/challenger.rs) over a script that crosses every state the buffers can
    be in: observations that end inside a block, exactly on a block, spanning several; challenges drawn from a fresh duplexing, from
    what an earlier draw left, after a partial observation; several sources per step, a planar extension source, non-canonical inputs;
    and hash_n_to_hash_no_pad of lengths 0..17 (hashing.rs:81-108: the short last chunk leaves the old words)."""
    import random

    import numpy as np

    import plonky2_gpu_amd as pg
    from oracle import fri_ref, pyref

    rnd = random.Random(77)
    P = pyref.P
    ref = fri_ref.Challenger()
    d_ch = pg.DeviceBuffer(gpu, 32)
    first = True
    script = [([3], 0), ([5], 2), ([], 7), ([8], 1), ([16, 1], 0), ([], 9), ([7, 9, 3], 3), ([64], 4), ([1], 1), ([1], 1), ([0], 8), ([23], 0), ([], 1)]
    for counts, n_out in script:
        srcs, keep = [], []
        for cnt in counts:
            vals = [rnd.randrange(1 << 64) if rnd.random() < 0.2 else rnd.randrange(P) for _ in range(cnt)]  # some >= p: observed mod p
            buf = pg.DeviceBuffer.from_host(gpu, np.array(vals or [0], dtype=np.uint64))
            keep.append(buf)
            srcs.append((buf, cnt, 0))
            ref.observe_elements(vals)
        got = _step(pg, gpu, d_ch, srcs, n_out, flags=1 if first else 0)
        first = False
        assert got == ref.get_n_challenges(n_out), (counts, n_out)
        for b in keep:
            b.free()
    # a planar extension vector [2][len] is observed interleaved (observe_extension_elements)
    ln = 5
    planes = [rnd.randrange(P) for _ in range(2 * ln)]
    buf = pg.DeviceBuffer.from_host(gpu, np.array(planes, dtype=np.uint64))
    ref.observe_extension_elements([(planes[i], planes[ln + i]) for i in range(ln)])
    assert _step(pg, gpu, d_ch, [(buf, 2 * ln, ln)], 3) == ref.get_n_challenges(3)
    buf.free()
    # the state in device memory is the transcript's: sponge state, input buffer, lengths
    T = d_ch.download()
    assert [int(v) for v in T[:12]] == ref.sponge_state and int(T[28]) == len(ref.input_buffer) and int(T[29]) == len(ref.output_buffer)
    # hash_no_pad on a scratch challenger
    d_scratch = pg.DeviceBuffer(gpu, 32)
    for ln in list(range(0, 18)) + [135]:
        vals = [rnd.randrange(P) for _ in range(ln)]
        buf = pg.DeviceBuffer.from_host(gpu, np.array(vals or [0], dtype=np.uint64))
        assert _step(pg, gpu, d_scratch, [(buf, ln, 0)], 0, flags=3) == pyref.hash_no_pad(vals), ln
        buf.free()
    d_scratch.free()
    d_ch.free()


@pytest.mark.gpu
def test_fold_open_and_proof_of_work_read_the_transcript_from_device_memory(gpu):
    """gl_fri_fold_device = gl_fri_fold with the same beta; gl_merkle_open_batch_device with raw challenges and a shift = gl_merkle_open_batch
    with the reduced indices; gl_fri_proof_of_work_device on a device challenger = gl_fri_proof_of_work on the same duplex state."""
    import ctypes
    import random

    import numpy as np

    import plonky2_gpu_amd as pg
    from oracle import fri_ref, pyref
    from plonky2_gpu_amd import _lib

    rnd = random.Random(78)
    P = pyref.P
    # fold
    ln, ab = 1 << 10, 3
    coeffs = np.array([rnd.randrange(P) for _ in range(2 * ln)], dtype=np.uint64)
    beta = np.array([rnd.randrange(P), rnd.randrange(P)], dtype=np.uint64)
    d_c, d_b = pg.DeviceBuffer.from_host(gpu, coeffs), pg.DeviceBuffer.from_host(gpu, beta)
    d_o1, d_o2 = pg.DeviceBuffer(gpu, 2 * (ln >> ab)), pg.DeviceBuffer(gpu, 2 * (ln >> ab))
    _lib.call("gl_fri_fold", d_c.ptr, ln, ab, beta, d_o1.ptr, gpu.ptr)
    _lib.call("gl_fri_fold_device", d_c.ptr, ln, ab, d_b.ptr, d_o2.ptr, gpu.ptr)
    assert (d_o1.download() == d_o2.download()).all()
    # openings: a tree over 2^9 leaves of 6 elements, cap height 2; queries are raw 64-bit challenges, the tree is the one after a shift of 3
    n_leaves, leaf_len, cap_h, shift, count = 1 << 9, 6, 2, 3, 11
    rows = np.array([rnd.randrange(P) for _ in range(n_leaves * leaf_len)], dtype=np.uint64)
    d_rows = pg.DeviceBuffer.from_host(gpu, rows)
    d_dig, d_cap = pg.DeviceBuffer(gpu, 8 * (n_leaves - (1 << cap_h)) + 4), pg.DeviceBuffer(gpu, 4 << cap_h)
    _lib.call("gl_merkle_tree_from_leaves", d_rows.ptr, leaf_len, n_leaves, cap_h, d_dig.ptr, d_cap.ptr, gpu.ptr)
    raw = np.array([rnd.randrange(1 << 64) for _ in range(count)], dtype=np.uint64)
    reduced = np.array([(int(x) % (n_leaves << shift)) >> shift for x in raw], dtype=np.uint64)
    layers = 9 - cap_h
    h_l, h_s = np.zeros(count * leaf_len, dtype=np.uint64), np.zeros(count * layers * 4, dtype=np.uint64)
    _lib.call("gl_merkle_open_batch", d_rows.ptr, leaf_len, 1, leaf_len, n_leaves, cap_h, d_dig.ptr, reduced, count, h_l, h_s, gpu.ptr)
    d_raw = pg.DeviceBuffer.from_host(gpu, raw)
    d_ol, d_os = pg.DeviceBuffer(gpu, count * leaf_len), pg.DeviceBuffer(gpu, count * layers * 4)
    _lib.call("gl_merkle_open_batch_device", d_rows.ptr, leaf_len, 1, leaf_len, n_leaves, cap_h, d_dig.ptr, d_raw.ptr, count, shift, d_ol.ptr, d_os.ptr, gpu.ptr)
    assert (d_ol.download() == h_l).all() and (d_os.download() == h_s).all()
    # proof of work: a transcript with three elements waiting in its input buffer
    ref = fri_ref.Challenger()
    obs = [rnd.randrange(P) for _ in range(19)]
    ref.observe_elements(obs)
    d_ch = pg.DeviceBuffer(gpu, 32)
    d_obs = pg.DeviceBuffer.from_host(gpu, np.array(obs, dtype=np.uint64))
    assert _step(pg, gpu, d_ch, [(d_obs, len(obs), 0)], 0, flags=1) == []
    state = list(ref.sponge_state)
    for i, x in enumerate(ref.input_buffer):
        state[i] = x
    w_host, w_dev = ctypes.c_uint64(), ctypes.c_uint64()
    _lib.call("gl_fri_proof_of_work", np.array(state, dtype=np.uint64), len(ref.input_buffer), 12, ctypes.byref(w_host), gpu.ptr)
    d_w = pg.DeviceBuffer(gpu, 2)
    _lib.call("gl_fri_proof_of_work_device", d_ch.ptr, 12, d_w.ptr, ctypes.byref(w_dev), gpu.ptr)
    assert w_dev.value == w_host.value and int(d_w.download()[0]) == w_host.value
    ref.observe_element(w_host.value)
    resp = ref.get_challenge()
    assert resp >> (64 - 12) == 0
    assert _step(pg, gpu, d_ch, [(d_w, 1, 0)], 1) == [resp]
    for b in (d_c, d_b, d_o1, d_o2, d_rows, d_dig, d_cap, d_raw, d_ol, d_os, d_ch, d_obs, d_w):
        b.free()


@pytest.mark.gpu
def test_device_transcript_entry_points_refuse_bad_arguments(gpu):
    """The round-6 entry points return GL_E_INVALID (a message, no launch) for null pointers, more than eight sources, a planar source
    longer than its two planes, unknown flags, a tree shape that is not a power of two, a shift that does not fit."""
    import ctypes

    import numpy as np

    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    d = pg.DeviceBuffer(gpu, 64)
    src = (_lib.GlObserveSrc * 9)(*[_lib.GlObserveSrc(d.ptr, 1, 0) for _ in range(9)])
    bad = [
        ("gl_challenger_step", None, ctypes.addressof(src), 1, 0, None, 1, gpu.ptr),                 # no challenger
        ("gl_challenger_step", d.ptr, ctypes.addressof(src), 9, 0, None, 1, gpu.ptr),                # nine sources
        ("gl_challenger_step", d.ptr, ctypes.addressof(src), 1, 2, None, 1, gpu.ptr),                # challenges but no output
        ("gl_challenger_step", d.ptr, ctypes.addressof(src), 1, 0, d.ptr, 4, gpu.ptr),               # unknown flag
        ("gl_challenger_step", d.ptr, None, 1, 0, None, 1, gpu.ptr),                                 # sources announced, none given
        ("gl_fri_fold_device", d.ptr, 16, 2, None, d.ptr, gpu.ptr),
        ("gl_fri_fold_device", d.ptr, 16, 0, d.ptr, d.ptr, gpu.ptr),                                 # arity 0
        ("gl_fri_proof_of_work_device", d.ptr, 12, None, ctypes.addressof(ctypes.c_uint64()), gpu.ptr),
        ("gl_fri_proof_of_work_device", d.ptr, 41, d.ptr, ctypes.addressof(ctypes.c_uint64()), gpu.ptr),
        ("gl_merkle_open_batch_device", d.ptr, 4, 1, 4, 12, 2, d.ptr, d.ptr, 1, 0, d.ptr, d.ptr, gpu.ptr),   # 12 leaves
        ("gl_merkle_open_batch_device", d.ptr, 4, 1, 4, 16, 2, d.ptr, d.ptr, 1, 33, d.ptr, d.ptr, gpu.ptr),  # shift too large
        ("gl_merkle_open_batch_device", d.ptr, 4, 1, 4, 16, 2, d.ptr, None, 1, 0, d.ptr, d.ptr, gpu.ptr),    # no indices
    ]
    planar = (_lib.GlObserveSrc * 1)(_lib.GlObserveSrc(d.ptr, 9, 4))  # 9 elements of a [2][4] vector
    bad.append(("gl_challenger_step", d.ptr, ctypes.addressof(planar), 1, 0, None, 1, gpu.ptr))
    for call in bad:
        with pytest.raises(_lib.Plonky2HipError) as e:
            _lib.call(*call)
        assert e.value.code == _lib.GL_E_INVALID, call[0]
    # and the zero-work forms are accepted: no sources, no challenges; zero queries
    _lib.call("gl_challenger_step", d.ptr, None, 0, 0, None, 1, gpu.ptr)
    _lib.call("gl_merkle_open_batch_device", d.ptr, 4, 1, 4, 16, 2, d.ptr, d.ptr, 0, 0, d.ptr, d.ptr, gpu.ptr)
    gpu.synchronize()
    assert not d.download()[:12].any() and int(d.download()[28]) == 0   # the reset left an empty transcript
    d.free()
