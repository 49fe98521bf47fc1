#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X prover hot path.

Metric (BASELINE.json): NTTs/sec at 2^20 Goldilocks.
Workload at N=1 (BASELINE.json configs[1]): a batch of 64 independent 2^20-point columns
(512 MiB, resident in HBM before the timed region), one step = forward NTT of the batch followed
by the inverse NTT of the batch = 128 transforms, natural order in and out, bit-exact against
field/src/fft.rs (checked every run outside the timed region, without the oracle: the batch must
survive the round trips unchanged, and sampled outputs must equal the DFT definition; the oracle is
used by the cpu_baseline leg only, the parity tests proper are tests/ -m gpu). Each rank owns its own batch on its own GPU: the path shards by column with no
data-path collective, so scaling is weak and torch.distributed is only used for
the barrier and the max-over-ranks of the elapsed time.

Also reported in the same JSON line:
  roofline      HBM roofline of the forward batch transform (the two ntt_pass_kernel launches),
                algorithmic bytes = 16 B x 2^20 x 64 per batch transform,
                duration from HIP events on the launch stream.
  cpu_baseline  the C oracle (a restatement of the reference's fft_classic, "port") on this
                host's cores, bounded sample.
  extra         PolynomialBatch::from_values on BASELINE configs[2] (2^20 rows x 135 columns,
                rate 8, cap 4): ms per commit and Merkle leaves hashed/s (skipped with --no-commit);
                prove() wall-clock on a synthetic circuit of configs[3]'s shape (n = 2^18, 234 wires,
                88 preprocessed polynomials, FRI as standard_recursion_config), one independent proof
                per GPU and step as in configs[4] (skipped with --no-prove).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); what a copy kernel reaches is measured in the run
LOG_N = 20
BATCH = 64
MIN_REGION_S = 0.5  # the timed region lasts at least this long (28 ms windows on devices that differ by 9 %)
WINDOWS = 5         # further windows of the same size after the timed one: median / min / max are reported beside `value`


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--inner", type=int, default=0, help="forward + inverse pairs per step (0 = as many as make the timed region >= 0.5 s)")
    ap.add_argument("--windows", type=int, default=WINDOWS, help="repeats of the timed region reported as value_windows")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-commit", action="store_true", help="skip the configs[2] commit measurement")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-prove", action="store_true", help="skip the configs[3]/[4]-shaped prove() measurement")
    ap.add_argument("--no-reference", action="store_true", help="skip timing the reference's own kernels (oracle/_ref) on this GPU")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="N > 1 ranks on whatever devices there are, over a HOST backend, through the code the RCCL route takes (device tensors over the "
                         "library's pointers, staged through host memory for transport only), with the invariants of the N-rank line checked: see dry_ranks_checks")
    ap.add_argument("--reference-leg-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--prove-degree-bits", type=int, default=18)
    ap.add_argument("--prove-wires", type=int, default=234)
    ap.add_argument("--prove-reps", type=int, default=3)
    ap.add_argument("--prove-in-flight", type=int, default=3, help="host threads, each with its own context and circuit, proving at the same time on one GPU for prove_proofs_per_s (0 = skip)")
    ap.add_argument("--prove-larger", default="19,20", help="further trace sizes (log2 rows) at which one proof of the same shape is timed, rank 0 at N = 1 ('' = none)")
    ap.add_argument("--commit-cols", type=int, default=135)
    ap.add_argument("--commit-log-n", type=int, default=20)
    return ap.parse_args()


def guarded_leg(name, fn, shape=None):
    """A baseline leg, or the reason it could not run (no compiler for the checker, out of host memory, a check of its own that
    fails): a failing baseline leg never costs the GPU measurements of the line."""
    try:
        return fn()
    except BaseException as e:  # noqa: BLE001 (SystemExit included)
        return dict(shape or {}, absent_because=f"{name}: {type(e).__name__}: {e}"[:300])


def cpu_baseline(log_n):
    """The oracle's fft/ifft (port of fft.rs:73-229) on a bounded sample, clocked inside C around the parallel region
    (oracle/gl_oracle.c glo_fft_bench): the root table is built before the clock starts — the reference builds
    fft_root_table once per circuit (circuit_builder.rs:849-851) — every thread is pinned to a core of its own, transforms
    columns it allocated and filled itself (first touch on its own NUMA node) and reads its node's copy of the table, one
    column per task as the reference's rayon split (oracle.rs:720).

    `value` is the rate on ALL the cores this process may use: the host's hardware threads, or — when the container has a
    CPU quota (cgroup cpu.max; the pool's GPU boxes grant 16 CPUs of the host's 256 hardware threads, and threads beyond
    the quota only time-slice it: 256 threads were SLOWER than 32 in round 2) — as many threads as the quota grants."""
    from oracle import oracle as o

    hw = o.hardware_threads()
    quota = o.cpu_quota()
    cores = max(1, min(hw, int(quota))) if quota else hw
    n = 1 << log_n
    cols = 2
    one = o.fft_bench(n, 1, 4)
    per_thread = 2 * 4 / one
    # A quota means a shared host: other tenants keep some cores busy, and the pool's boxes gave 190-380 NTT/s for the same
    # call within one minute. There the threads are left to the scheduler (pinned ones can land on a busy core); on a host of
    # one's own they are pinned, one per core, for NUMA-local columns.
    if quota:
        os.environ["GLO_FFT_BENCH_NO_PIN"] = "1"
    # the all-core figure, repeated until about 10 s of timed work have accumulated (bounded: at most 24 calls)
    timed, calls, per_call = 0.0, 0, []
    while timed < 10.0 and calls < 24:
        dt = o.fft_bench(n, cores, cols, seed=0x706C6F6E6B7932 + 1000 + calls)
        per_call.append(2 * cores * cols / dt)
        timed += dt
        calls += 1
    transforms = 2 * cores * cols * calls
    # for the record: other thread counts, one call each (more threads than the quota grants, fewer than the cores)
    others = {}
    for t in sorted({max(1, cores // 2), min(hw, 2 * cores), hw} - {cores}):
        others[str(t)] = 2 * t * cols / o.fft_bench(n, t, cols, seed=0x706C6F6E6B7932 + t)
    best = max([transforms / timed] + list(others.values()))
    return {
        "value": transforms / timed,
        "unit": "NTT/s",
        "cores": cores,
        "kind": "port",
        "host_hardware_threads": hw,
        "container_cpu_quota": quota,
        "one_thread_NTT_per_s": per_thread,
        "NTT_per_s_per_call_min_median_max": [min(per_call), sorted(per_call)[len(per_call) // 2], max(per_call)],
        "threads_pinned": not quota,
        "NTT_per_s_at_other_thread_counts": others,
        "best_of_all_thread_counts_NTT_per_s": best,
        "parallel_efficiency": transforms / timed / (per_thread * cores),
        "sample": f"{calls} x ({cores} threads x {cols} columns of 2^{log_n}, forward + inverse) = {transforms} transforms in "
                  f"{timed:.2f} s of in-C wall clock on {cores} threads = every CPU the container may use "
                  f"({hw} hardware threads on the host, cgroup quota {quota}); C restatement of fft_classic, root table "
                  f"prebuilt per NUMA node, thread-local columns; one thread alone: {per_thread:.2f} NTT/s",
    }


def cpu_baseline_commit(cols, log_n, rate_bits=3, cap_height=4):
    """configs[2]'s leg on the CPU, WHOLE: the C oracle's commit_from_values (restating PolynomialBatch::from_values, fri/oracle.rs:709-731,
    and MerkleTree::new, hash/merkle_tree.rs:283-319: ifft per column, coset LDE per column, transpose + bit reversal, leaf hashing
    and the cap subtrees, threaded per column / per subtree like the reference's rayon split) of the same shape as the GPU leg —
    135 columns x 2^20 rows — on every core the container may use: half a minute on the pool's 16 CPUs (round 4 ran half the
    rows and projected)."""
    from oracle import oracle as o

    cores = o.usable_threads()
    n = 1 << log_n
    vals = o.random_field((cols, n), seed=0x706C6F6E6B7932 + 7)
    t = time.perf_counter()
    o.commit_from_values(vals, rate_bits, cap_height, threads=cores, want_leaves=True)
    dt = time.perf_counter() - t
    leaves = n << rate_bits
    return {"value": leaves / dt, "unit": "Merkle leaves/s (whole commit)", "cores": cores, "kind": "port", "seconds": dt, "ms": dt * 1e3,
            "poseidon_permutations_per_s": leaves * ((cols + 7) // 8 + 1) / dt,
            "sample": f"one from_values of {cols} columns x 2^{log_n} rows = configs[2] whole, rate {1 << rate_bits}, cap_height {cap_height}, "
                      f"leaf-major matrix included, on {cores} threads; C restatement (oracle/gl_oracle.c glo_commit_from_values)"}


def cpu_baseline_prove(circuit, wires, pis, gpu_proof):
    """configs[3]'s leg on the CPU: a REAL prove() — the C restatement of plonk/prover.rs:40-233 from the full witness on
    (oracle/prove_oracle.c: the three commitments, the permutation argument, all 25 gates' constraints at every LDE point, the
    quotient, the opening set, the FRI opening proof with the smallest proof-of-work witness, the wire format), threaded like
    the reference's rayon loops, on every core the container may use — of the VERY circuit and witness rank 0's GPU leg proved.
    As the checker it also says whether its bytes equal gl_prove's (they must: tests/test_gpu_prove.py asserts it). The
    preprocessing (constants/sigmas commitment, circuit_builder.rs:849-960) is outside the clock on both sides. Until round 5
    this leg timed the three commitments only. The reference's README quotes 45 s for its CPU prover on its authors' 8 cores."""
    from oracle import prove_c

    oc = prove_c.Circuit(circuit)
    tr = {}
    t = time.perf_counter()
    data = oc.prove(wires, pis, trace=tr)
    dt = time.perf_counter() - t
    oc.close()
    return {"value": dt * 1e3, "unit": "ms per proof (whole prove() from the full witness on)", "cores": oc.threads, "kind": "port",
            "stage_ms": {k: round(v * 1e3, 1) for k, v in tr["stage_seconds"].items()},
            "proof_bytes": len(data), "bytes_equal_gl_prove": (data == gpu_proof) if gpu_proof is not None else None,
            "sample": f"one proof of the bench's own circuit and witness (n = 2^{circuit['degree_bits']}, {circuit['num_wires']} wires, "
                      f"{len(circuit['gates'])} gates) on {oc.threads} threads with the C restatement of prove() (oracle/prove_oracle.c)"}


def cpu_baseline_reference_gpu_kernels(pg, _lib, ctx, log_n, batch, commit_cols, commit_log_n):
    """Not a CPU figure but the same kind of leg (a stated baseline beside the product, rank 0 at N = 1, outside every timed
    region): the REFERENCE'S OWN kernels (cuda/plonky2_gpu_impl.cuh compiled unmodified for gfx950, oracle/_ref, see
    oracle/ref_harness.hip) timed on this very MI355X with the launch geometry of cuda/plonky2_gpu.cu. The only same-node
    comparison with the reference this project can have; never the target."""
    from oracle import oracle as o, ref_gpu

    if not ref_gpu.available():
        return {"absent_because": ref_gpu.why_absent()}
    out = {"library": os.path.relpath(ref_gpu.PATH, ROOT), "what": "cuda/plonky2_gpu_impl.cuh, unmodified, hipcc --offload-arch=gfx950; durations by HIP events inside oracle/ref_harness.hip"}
    n = 1 << log_n
    # (1) the headline transform: fft_kernel / ifft_kernel, one 256-thread block per polynomial (plonky2_gpu.cu:81, 131)
    x = pg.DeviceBuffer.from_host(ctx, o.random_field((batch, n), seed=11))
    table = pg.DeviceBuffer.from_host(ctx, o.root_table_concat(n))
    ctx.synchronize()
    f = [ref_gpu.call("ref_fft", x.ptr, batch, n, log_n, table.ptr, 0)[0] for _ in range(3)]
    i = [ref_gpu.call("ref_ifft", x.ptr, batch, n, log_n, table.ptr, ref_gpu.n_inv(log_n))[0] for _ in range(3)]
    out["ntt"] = {"workload": f"{batch} columns x 2^{log_n}, fft_kernel then ifft_kernel", "fft_ms": min(f), "ifft_ms": min(i),
                  "NTT_per_s": 2 * batch / ((min(f) + min(i)) * 1e-3)}
    x.free()
    table.free()
    # (2) the commit of configs[2] = the launches of merkle_tree_from_values (plonky2_gpu.cu:218-433; body disabled by assert(0),
    # its live twin is ifft + merkle_tree_from_coeffs :435-606): ifft, lde, init_lde, mul_shift, fft(r), reverse_index_bits,
    # hash_leaves, reduce_digests, transpose
    cols, cn = commit_cols, 1 << commit_log_n
    rate_bits, h = 3, 4
    n_ext = cn << rate_bits
    pad = cols * n_ext
    if pad >= 1 << 31:
        out["commit"] = {"absent_because": "the reference indexes with int: (poly_num + salt) * n_ext must stay below 2^31"}
        return out
    region = pg.DeviceBuffer(ctx, 2 * pad + 4 * 2 * (n_ext - (1 << h)) + (4 << h))
    seed_block = o.random_field((1 << 22,), seed=12)
    for off in range(0, cols * cn, 1 << 22):  # values: one random block repeated (durations do not depend on the values)
        region.upload(seed_block[:min(1 << 22, cols * cn - off)], off)
    t1 = pg.DeviceBuffer.from_host(ctx, o.root_table_concat(cn))
    t2 = pg.DeviceBuffer.from_host(ctx, o.root_table_concat(n_ext))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_circuit as sc

    pw = np.ones(1, dtype=np.uint64)
    while pw.size < cn:
        pw = np.concatenate([pw, sc.np_mul(pw, np.uint64(pow(7, pw.size, sc.P)))])
    shifts = pg.DeviceBuffer.from_host(ctx, pw[:cn])
    ctx.synchronize()
    ms = {}
    ms["ifft_kernel"] = ref_gpu.call("ref_ifft", region.ptr, cols, cn, commit_log_n, t1.ptr, ref_gpu.n_inv(commit_log_n))[0]
    lde = ref_gpu.call("ref_coset_lde", region.ptr, region.at(pad), cols, cn, commit_log_n, t2.ptr, shifts.ptr, rate_bits, n_ms=4)
    ms.update({"lde_kernel": lde[0], "init_lde_kernel": lde[1], "mul_shift_kernel": lde[2], "fft_kernel": lde[3]})
    ms["reverse_index_bits_kernel"] = ref_gpu.call("ref_reverse_index_bits", region.at(pad), cols, n_ext, commit_log_n + rate_bits)[0]
    mk = ref_gpu.call("ref_merkle_tree", region.at(pad), cols, n_ext, h, n_ms=2)
    ms.update({"hash_leaves_kernel": mk[0], "reduce_digests_kernel": mk[1]})
    ms["transpose_kernel"] = ref_gpu.call("ref_transpose", region.at(pad), region.ptr, cols, n_ext)[0]
    total = sum(ms.values())
    out["commit"] = {"workload": f"configs[2]: {cols} columns x 2^{commit_log_n} rows, rate 8, cap_height {h}: every launch of ifft + merkle_tree_from_coeffs",
                     "kernel_ms": ms, "commit_ms": total, "merkle_leaves_per_s": n_ext / (total * 1e-3)}
    for b in (region, t1, t2, shifts):
        b.free()
    return out


def cpu_baseline_reference_gpu_kernels_in_child(log_n, batch, commit_cols, commit_log_n, timeout_s=420):
    """The leg above in a FRESH CHILD PROCESS with a time limit: the reference's kernels are untrusted code (int indexing,
    32-thread blocks, member-function-pointer stacks, the null stream) — a memory fault or a hang there must cost this leg,
    not the measurements already taken. The child is this script with --reference-leg-child; it prints one JSON
    object; anything else (non-zero exit, timeout, unparsable output) becomes `absent_because`."""
    import subprocess

    from oracle import ref_gpu

    if not ref_gpu.available():
        return {"absent_because": ref_gpu.why_absent()}
    cmd = [sys.executable, os.path.abspath(__file__), "--reference-leg-child", "--log-n", str(log_n), "--batch", str(batch),
           "--commit-cols", str(commit_cols), "--commit-log-n", str(commit_log_n)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"absent_because": f"the child running the reference's kernels did not finish within {timeout_s} s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"absent_because": f"the child running the reference's kernels exited with {r.returncode}: {r.stderr[-300:]}"}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"absent_because": f"unparsable output of the child: {e}"}


def cpu_baseline_reference_leg_child(args):
    """entry of the child process started by cpu_baseline_reference_gpu_kernels_in_child"""
    from oracle import ref_gpu  # noqa: F401  (this function is part of the baseline leg)
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    ctx = pg.Context(0)
    print(json.dumps(cpu_baseline_reference_gpu_kernels(pg, _lib, ctx, args.log_n, args.batch, args.commit_cols, args.commit_log_n)), flush=True)
    ctx.close()


SPLITMIX_SEED = 0x706C6F6E6B7932  # "plonky2" 
P = 0xFFFFFFFF00000001


def splitmix64_field(count, start=0):
    """`count` uniform elements of [0, p): SplitMix64 from SPLITMIX_SEED, draws >= p rejected (the uniform-canonical
    contract of Sample::sample, goldilocks_field.rs:61-70); `start` = index of the first draw of the stream to use."""
    gamma, m1, m2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)
    out = np.empty(count, dtype=np.uint64)
    filled, pos, chunk = 0, int(start), 1 << 22
    with np.errstate(over="ignore"):
        while filled < count:
            k = np.arange(pos + 1, pos + 1 + chunk, dtype=np.uint64)
            z = np.uint64(SPLITMIX_SEED) + k * gamma
            z = (z ^ (z >> np.uint64(30))) * m1
            z = (z ^ (z >> np.uint64(27))) * m2
            z ^= z >> np.uint64(31)
            z = z[z < np.uint64(P)]
            take = min(z.size, count - filled)
            out[filled:filled + take] = z[:take]
            filled += take
            pos += chunk
    return out


def dft_point(x, log_n, k):
    """X[k] = sum_j x[j] * w^(j k) mod p, w = the primitive 2^log_n-th root of unity (field/src/types.rs:268-272;
    fft.rs:242-282 checks fft against exactly this naive evaluation). Vectorised numpy, no oracle."""
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import synth_circuit as sc

    p = sc.P
    wk = pow(pow(1753635133440165772, 1 << (32 - log_n), p), k, p)
    pw = np.ones(1, dtype=np.uint64)  # (w^k)^j for j < n, by doubling
    for b in range(log_n):
        pw = np.concatenate([pw, sc.np_mul(pw, np.uint64(pow(wk, 1 << b, p)))])
    terms = sc.np_mul(np.asarray(x, dtype=np.uint64), pw)
    while terms.size > 1:  # modular sum by halving
        half = terms.size // 2
        terms = sc.np_add(terms[:half], terms[half:])
    return int(terms[0])


PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06_pmc_summary.json")  # tools/pmc_summary.py over the rocprofv3 --pmc passes
# the kernel sources whose counters the summary holds: tools/pmc_summary.py records their sha256 next to the counters
PMC_SOURCES = ["plonky2_gpu_amd/csrc/ntt.hip", "plonky2_gpu_amd/csrc/ntt_direct.hip", "plonky2_gpu_amd/csrc/ntt_kernels.h",
               "plonky2_gpu_amd/csrc/poseidon.h", "plonky2_gpu_amd/csrc/poseidon_limb_constants.h", "plonky2_gpu_amd/csrc/gl_field.h"]


def source_hashes(root=ROOT):
    import hashlib

    out = {}
    for rel in PMC_SOURCES:
        path = os.path.join(root, rel)
        out[rel] = hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None
    return out


def pmc_summary(path=None, root=ROOT):
    """(summary, None) when the committed counter summary was collected on the kernel sources of this tree, (None, reason)
    otherwise: counters are not measured by bench.py (rocprofv3 cannot run inside it), so a summary that is older than the
    kernels it describes must not be quoted."""
    path = path or PMC_SUMMARY
    if not os.path.exists(path):
        return None, f"{os.path.relpath(path, root)} is absent"
    d = json.load(open(path))
    have, now = d.get("source_sha256"), source_hashes(root)
    if not have:
        return None, f"{os.path.relpath(path, root)} records no source hashes"
    stale = sorted(k for k in now if have.get(k) != now[k])
    if stale:
        return None, f"{os.path.relpath(path, root)} was collected on other versions of {', '.join(stale)}: regenerate it (tools/gpu_runs/pmc_passes.sh, tools/pmc_summary.py)"
    return d, None


def _kernel_key(name):
    """'ntt_col_direct_kernel<2,true,false> grid=262144' -> ('ntt_col_direct_kernel', ['2', 'true', 'false'], 262144)"""
    head, _, grid = name.partition(" grid=")
    base, _, args = head.partition("<")
    return base, [a.strip() for a in args.rstrip(">").split(",")] if args else [], int(grid) if grid.isdigit() else 0


def forward_ntt_kernels(d):
    """The two kernels of the forward natural-order batch transform in a counter summary: the natural-order, non-coset column
    pass and the forward (non-inverse) natural row pass, each at the largest grid it was launched with (the 64-column batch;
    smaller grids are the one-column parity check). (col_entry, row_entry, None) or (None, None, why): anything but exactly one
    of each is an error, never a silent partial sum (a renamed template dropped the column pass unnoticed)."""
    col, row = [], []
    for name, e in d.get("kernels", {}).items():
        base, args, grid = _kernel_key(name)
        # <LOGG, NATURAL, COSET[, FINAL]>: the natural-order, non-coset pass that carries the inter-pass twiddle (FINAL, round 5, is the
        # last pass of the two-pass plan for 2^22 and never part of the 2^20 transform)
        if base == "ntt_col_direct_kernel" and len(args) >= 3 and args[1] == "true" and args[2] == "false" and (len(args) < 4 or args[3] == "false"):
            col.append((grid, name, e))
        elif base == "ntt_row_natural_direct_kernel" and args[:1] == ["false"]:
            row.append((grid, name, e))
    picked = []
    for what, found in (("column", col), ("row", row)):
        if not found:
            return None, None, f"the summary holds no forward {what}-pass kernel"
        top = max(g for g, _, _ in found)
        best = [x for x in found if x[0] == top]
        if len(best) != 1:
            return None, None, f"{len(best)} forward {what}-pass kernels at grid {top}: {[n for _, n, _ in best]}"
        picked.append(best[0])
    return picked[0], picked[1], None


def traffic_of_summary(d, batch):
    """HBM-side bytes per forward batch transform of `batch` columns: FETCH_SIZE x 2 (gfx950 note of MI355X_MICROARCH.md) +
    WRITE_SIZE of the column pass and of the row pass, scaled from the columns per launch the passes were collected with."""
    c, r, why = forward_ntt_kernels(d)
    if why:
        return None, why
    total = 0.0
    for _, name, e in (c, r):
        rd, wr = e["derived"].get("read_bytes (FETCH_SIZE KiB x 1024 x 2)"), e["derived"].get("write_bytes (WRITE_SIZE KiB x 1024)")
        if not rd or not wr:
            return None, f"{name} has no FETCH_SIZE / WRITE_SIZE pass"
        total += rd + wr
    return total * (batch / float(d.get("ntt_columns_per_launch", 64))), None


def int_alu_of_summary(d):
    """The binding roof of the passes as a number: executed vector-ALU instructions x 4 cycles (the issue cost of a wave64
    instruction on a SIMD-16) / (1024 SIMDs x the kernels' cycles), both kernels of the forward transform together."""
    c, r, why = forward_ntt_kernels(d)
    if why:
        return None
    insts = sum(e["counters"].get("SQ_INSTS_VALU", 0) for _, _, e in (c, r))
    cycles = sum(e["derived"].get("kernel_cycles", 0) for _, _, e in (c, r))
    if not insts or not cycles:
        return None
    return {"frac": insts * 4.0 / (1024 * cycles), "valu_insts_per_launch_pair": insts, "kernel_cycles_per_launch_pair": cycles,
            "wave_parked_frac": {n.split(" ")[0]: e["derived"].get("wave_parked_frac (s_waitcnt / barrier)") for _, n, e in (c, r)},
            "definition": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE/8), column pass + row pass"}


def pmc_traffic(log_n, batch):
    """(bytes, source note): HBM-side bytes per forward batch transform from the committed counter summary (rocprofv3
    FETCH_SIZE and WRITE_SIZE in separate passes), per launch pair of `batch` columns; (None, why) when the summary is absent,
    stale, or does not hold exactly one column and one row kernel."""
    d, why = pmc_summary()
    if not d:
        return None, why
    if log_n != 20:
        return None, "the counter passes were collected at 2^20"
    total, why = traffic_of_summary(d, batch)
    if total is None:
        return None, why
    return total, os.path.relpath(PMC_SUMMARY, ROOT)


def commit_hash_kernel_counters():
    """Counters of the kernel that dominates a commit, taken on THAT kernel inside the commit (hash_leaves_chunk_kernel at the
    configs[2] grid), not on a stand-alone permutation kernel: instructions per wavefront, the kernel's cycles and duration,
    hence its clock, and the vector-ALU issue estimate (SQ_INSTS_VALU minus the matrix instructions, x 4 cycles, / SIMDs x cycles).
    An ESTIMATE of how busy the issue port is, not a bound: not every vector instruction costs 4 cycles."""
    d, why = pmc_summary()
    if not d:
        return None, why
    found = [(_kernel_key(n)[2], n, e) for n, e in d["kernels"].items() if _kernel_key(n)[0] == "hash_leaves_chunk_kernel"]
    if not found:
        return None, "no hash_leaves_chunk_kernel in the summary"
    grid, name, e = max(found)
    c, dv = e["counters"], e["derived"]
    valu, mfma, cyc = c.get("SQ_INSTS_VALU"), c.get("SQ_INSTS_MFMA", 0.0), dv.get("kernel_cycles")
    out = {"kernel": name, "leaves": grid, "valu_insts_per_wavefront": dv.get("valu_insts_per_wave"), "matrix_insts_per_wavefront": mfma / c["SQ_WAVES"] if c.get("SQ_WAVES") else None,
           "avg_us_per_launch": e.get("avg_us"), "kernel_cycles": cyc,
           "clock_GHz_during_kernel": cyc / (e["avg_us"] * 1e3) if cyc and e.get("avg_us") else None,
           "valu_issue_estimate_frac_of_cycles": (valu - mfma) * 4.0 / (1024 * cyc) if valu and cyc else None,
           "wave_parked_frac": dv.get("wave_parked_frac (s_waitcnt / barrier)"),
           "source": os.path.relpath(PMC_SUMMARY, ROOT)}
    return out, None


def launch_ranks(n, argv=None, script=None, extra_env=None):
    """`bench.py --gpus N` started by hand (no torchrun, WORLD_SIZE unset): this process — which has not loaded the
    HIP library and never will — starts N copies of the script, one per rank, with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set the way torch.distributed.run sets them, waits for all of them and returns the
    worst exit code. Children are fresh processes (no fork of GPU state, no exec from a process that touched the GPU).
    Rank 0's stdout carries the JSON line; the other ranks print nothing on stdout."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = script or os.path.abspath(__file__)
    argv = list(sys.argv[1:] if argv is None else argv)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env))
    codes = []
    try:
        for p in procs:
            codes.append(p.wait())
    finally:
        for p in procs:  # a rank that died leaves the others in a barrier: end exactly the children started here
            if p.poll() is None:
                p.kill()
    bad = [c for c in codes if c != 0]
    return 0 if not bad else (max(bad) if max(bad) > 0 else 1)


def pick_backend(world, ndev):
    """RCCL ("nccl") when every rank has a GPU of its own; gloo when ranks share a device (RCCL refuses two ranks on
    one GPU) or there is one rank. PLONKY2_DIST_BACKEND overrides."""
    forced = os.environ.get("PLONKY2_DIST_BACKEND")
    if forced:
        return forced
    return "nccl" if (world > 1 and ndev >= world) else "gloo"


def main():
    args = parse()
    if args.reference_leg_child:
        return cpu_baseline_reference_leg_child(args)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # nothing below has run yet: no HIP call in this process
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd.dist import ProverGroup
    from plonky2_gpu_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # ORDER MATTERS at N > 1: torch carries its own copy of the HIP runtime, and a process in which libplonky2_hip (the system's
    # runtime) has made the first HIP call can no longer bring torch's up ("No HIP GPUs are available" — found by --dry-ranks in
    # round 6; the other order works, tests/dist_nccl_self.py). So the devices are counted through torch (which does not initialise
    # anything) and the process group — RCCL, or the dry device path — is formed BEFORE the library is loaded.
    if world > 1:
        import torch

        ndev = torch.cuda.device_count()
    else:
        ndev = pg.load().gl_device_count()
    if ndev <= 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # barrier / max-reduce / cap gather only (no data-path collective exists): over RCCL/xGMI when each rank has its
    # own GPU, over gloo when the ranks share one (the one-GPU box)
    dist = ProverGroup(backend="gloo" if args.dry_ranks else pick_backend(world, ndev), device_index=int(os.environ.get("LOCAL_RANK", "0")) % ndev,
                       dry_device_path=args.dry_ranks)
    if pg.load().gl_device_count() != ndev:
        raise SystemExit("bench.py: torch and libplonky2_hip see different numbers of devices")
    ctx = pg.Context(dist.local_rank % ndev)
    log_n, batch = args.log_n, args.batch
    n = 1 << log_n

    # synthetic input, resident in HBM before timing: SplitMix64 from the seed SURVEY 8(d) prescribes, uniform in
    # [0, p) by rejection; every rank takes its own segment of the one stream
    host = splitmix64_field(batch * n, start=dist.rank << 40).reshape(batch, n)
    buf = pg.DeviceBuffer.from_host(ctx, host)

    def pair():
        _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, 0, 0, ctx.ptr)
        _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, 1, 0, ctx.ptr)

    # A step is `inner` forward + inverse transforms of the batch. One pair takes about 1.2 ms, and devices of the pool differ
    # by several per cent on the same binary, so the timed region is kept at >= MIN_REGION_S whatever --steps says: `inner`
    # is set from a short calibration (the same on every rank: max over ranks) and printed in config.
    for _ in range(3):
        pair()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        pair()
    ctx.synchronize()
    t_pair = dist.max((time.perf_counter() - t0) / 8)
    inner = args.inner if args.inner > 0 else max(1, int(np.ceil(1.15 * MIN_REGION_S / (args.steps * t_pair))))  # 15 % margin: the calibration pairs run on a cold clock

    def step():
        for _ in range(inner):
            pair()

    region_events_ms = []   # the same regions between two HIP events on the launch stream (rank-local)

    def timed_window():
        ctx.synchronize()
        dist.barrier()
        e0, e1 = pg.Event(), pg.Event()
        t = time.perf_counter()
        e0.record(ctx)
        for _ in range(args.steps):
            step()
        e1.record(ctx)
        ctx.synchronize()
        dist.barrier()
        region_events_ms.append(e1.elapsed_ms_since(e0))
        return dist.max(time.perf_counter() - t)

    for _ in range(args.warmup):
        step()
    elapsed = timed_window()  # THE timed region: exactly --steps steps between barrier + synchronize on both sides
    # the same region again, WINDOWS times: the spread says how much one window can be trusted
    window_s = [elapsed] + [timed_window() for _ in range(max(0, args.windows))]

    # parity: after an equal number of forward/inverse transforms the batch must equal the input
    back = buf.download().reshape(batch, n)
    if not (back == host).all():
        raise SystemExit("bench: ifft(fft(x)) != x — results invalid")

    # The roofline's kernel = one batch transform = one launch pair (column pass + row pass; the forward and the inverse instance
    # alternate in the timed region and differ by the inverse's index flip in the row pass's stores). Its average launch duration
    # = the timed region between two HIP events on the launch stream / the launch pairs in it: what `rocprofv3 --kernel-trace
    # --stats` of this command reports for the two kernels together (profiles/). Bracketing every pair with its own events is
    # kept beside it: each event is a marker the queue drains for, so those figures run 3-6 % higher.
    launches_in_region = 2 * inner * args.steps
    pair_ms = region_events_ms[0] / launches_in_region
    reps = max(40, args.steps)
    ev = [pg.Event() for _ in range(2 * reps)]
    fwd_ms = []
    for r in range(reps):
        ev[2 * r].record(ctx)
        _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, 0, 0, ctx.ptr)
        ev[2 * r + 1].record(ctx)
        _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, 1, 0, ctx.ptr)
    ctx.synchronize()
    for r in range(reps):
        fwd_ms.append(ev[2 * r + 1].elapsed_ms_since(ev[2 * r]))
    fwd = float(np.median(fwd_ms))
    alg_bytes = 16.0 * n * batch
    achieved = alg_bytes / (pair_ms * 1e-3) / 1e9
    # what a streaming kernel moves on this very device: a 16 B/lane copy of the same 512 MiB, read + write
    scratch = pg.DeviceBuffer(ctx, batch * n)
    cev = [pg.Event() for _ in range(2)]
    copy_ms = []
    for r in range(6):
        cev[0].record(ctx)
        _lib.call("gl_debug_copy", scratch.ptr, buf.ptr, 8 * batch * n, ctx.ptr)
        cev[1].record(ctx)
        ctx.synchronize()
        if r:
            copy_ms.append(cev[1].elapsed_ms_since(cev[0]))
    scratch.free()
    copy_gbs = 2 * 8.0 * batch * n / (min(copy_ms) * 1e-3) / 1e9
    mulmods = (n // 2) * log_n * batch  # (N/2) lg N butterflies per transform, one mulmod + add + sub each
    traffic, traffic_source = pmc_traffic(log_n, batch)
    _d, _ = pmc_summary()
    int_alu = int_alu_of_summary(_d) if (_d and log_n == 20) else None

    out = None
    if dist.rank == 0:
        # Outside the timed region, and without the oracle (which only the cpu_baseline leg may touch): the
        # forward transform of column 0 against the DEFINITION X[k] = sum_j x[j] w^(jk) at a few output
        # indices, evaluated with vectorised numpy field arithmetic. With the round trip above this pins the
        # forward transform itself, not just its invertibility; the full parity tests are tests/test_gpu_ntt.py.
        _lib.call("gl_ntt_batch", buf.ptr, 1, log_n, n, 0, 0, ctx.ptr)
        got = buf.download(0, n)
        _lib.call("gl_ntt_batch", buf.ptr, 1, log_n, n, 1, 0, ctx.ptr)
        ctx.synchronize()
        for k in (1, 5, n // 2 + 3, n - 1):
            if int(got[k]) != dft_point(host[0], log_n, k):
                raise SystemExit(f"bench: forward NTT output {k} differs from the DFT definition")

    extra = {}
    if not args.no_commit and dist.rank == 0:
        extra = bench_commit(pg, _lib, ctx, args.commit_cols, args.commit_log_n)
        hk, why = commit_hash_kernel_counters()
        extra["commit_hash_kernel_counters"] = hk
        if not hk:
            extra["commit_hash_kernel_counters_absent_because"] = why
        elif hk.get("avg_us_per_launch"):
            # the commit is serial in effect (every kernel here fills the register file): leaf hashing vs the whole
            extra["commit_leaf_hashing_frac_of_commit_ms"] = hk["avg_us_per_launch"] * 1e-3 * ((args.commit_cols + 15) // 16) / extra["commit_ms"] \
                if "chunk" in hk["kernel"] else None

    if dist.world > 1 and not args.no_commit:
        sc = bench_sharded_commit(pg, ctx, dist, args.commit_cols, args.commit_log_n)
        if dist.rank == 0:
            extra["sharded_commit"] = sc

    prove_inputs = None
    if not args.no_prove:
        pr, prove_inputs = bench_prove(pg, ctx, dist, args.prove_degree_bits, args.prove_wires, args.prove_reps)
        if dist.rank == 0:
            extra["prove"] = pr

    if not args.no_prove and args.prove_in_flight > 1:
        if dist.world == 1:  # one rank: a failure of this leg (device memory, a second context) costs the leg, not the line
            fl = guarded_leg("prove_in_flight", lambda: bench_prove_in_flight(pg, ctx.device, dist, args.prove_degree_bits, args.prove_wires, args.prove_in_flight,
                                                                               max(4, 2 * args.prove_reps)))
        else:               # several ranks meet at barriers inside the leg: an exception must end the job, not strand the others
            fl = bench_prove_in_flight(pg, ctx.device, dist, args.prove_degree_bits, args.prove_wires, args.prove_in_flight, max(4, 2 * args.prove_reps))
        if dist.rank == 0:
            extra["prove_in_flight"] = fl

    if not args.no_prove and dist.world == 1 and args.prove_larger and args.prove_wires == 234:
        # north_star's 2^20-row traces as whole proofs (and 2^19 between): the same circuit shape, one timed proof each after a warm-up
        larger = {}
        for db in [int(x) for x in args.prove_larger.split(",") if x]:
            r, _ = bench_prove(pg, ctx, dist, db, args.prove_wires, 1)
            larger["2^%d rows" % db] = {k: r[k] for k in ("prove_ms", "proof_bytes", "stage_ms")}
        extra["prove_larger_traces"] = larger

    if dist.rank == 0:
        ntts = 2 * batch * inner * args.steps * dist.world
        rates = sorted(ntts / w for w in window_s)
        out = {
            "metric": "NTTs/sec at 2^20 Goldilocks",
            "value": ntts / elapsed,
            "unit": "NTT/s",
            "n_gpus": dist.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "value_windows": {"count": len(rates), "median": rates[len(rates) // 2], "min": rates[0], "max": rates[-1],
                              "seconds_per_window": round(float(np.median(window_s)), 3),
                              "note": "the timed region (`value`) and WINDOWS repeats of it, NTT/s each"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 (Goldilocks, integer modular)",
            "data": "synthetic",
            "config": {
                "workload": f"configs[1]: {batch} columns x 2^{log_n} points per GPU, {inner} x (forward + inverse NTT of the batch) per step "
                            f"(natural order in/out, bit-exact vs field/src/fft.rs), inputs resident in HBM",
                "batch_columns": batch,
                "log_n": log_n,
                "pairs_per_step": inner,
                "transforms_per_step": 2 * batch * inner,
                "parallelism": f"columns sharded over {dist.world} GPU(s), no collective",
                "ranks": dist.world,
                "devices_visible": ndev,
                "rank_sync_backend": dist.backend or "none (one rank)",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source if traffic is not None else None,
                "traffic_absent_because": None if traffic is not None else traffic_source,
                "traffic_over_algorithmic": traffic / alg_bytes if traffic is not None else None,
                "int_alu_frac": (int_alu or {}).get("frac"),
                "int_alu": int_alu,
                "kernel": "batch NTT = ntt_col_direct_kernel (column pass) + ntt_row_natural_direct_kernel (row pass), one launch pair per "
                          "64-column batch; forward and inverse instances alternate in the timed region",
                "binding_roof": ("HBM-side traffic is %.2f x the algorithmic bytes: the second pass of a two-pass transform re-reads and "
                                 "re-writes every element. " % (traffic / alg_bytes) if traffic is not None else "") +
                                "The passes are bound by vector-ALU issue, not by HBM (int_alu_frac; DESIGN.md 3.1)",
                "algorithmic_bytes_per_launch_pair": alg_bytes,
                "ms": pair_ms,
                "ms_definition": "HIP events around the timed region on the launch stream / launch pairs in it",
                "ms_launch_pairs_timed": launches_in_region,
                "ms_all_windows": [w / launches_in_region for w in region_events_ms],
                "ms_forward_pairs_bracketed_one_by_one": {"median": fwd, "min": float(min(fwd_ms)), "max": float(max(fwd_ms)), "pairs": reps,
                                                          "note": "every event is a marker the queue drains for: 3-6 % above the region's average"},
                "measured_copy_GBps": copy_gbs,
                "frac_of_measured_copy": achieved / copy_gbs,
                "algorithmic_butterflies_per_launch_pair": mulmods,
                "butterflies_per_s": mulmods / (pair_ms * 1e-3),
            },
            # rank 0 at N = 1 only: at N > 1 the host cores belong to the other ranks' transcripts
            "cpu_baseline": None if (args.no_cpu or dist.world > 1) else guarded_leg("cpu_baseline", lambda: cpu_baseline(log_n), {"value": None, "unit": "NTT/s", "cores": None, "kind": "port"}),
            "extra": extra,
        }
        if not (args.no_cpu or dist.world > 1):
            # The other two legs of the metric on this host's cores, and the reference's own kernels on this GPU. Each leg is guarded:
            # a CPU leg that fails (no compiler for the oracle, out of host memory) costs that leg, never the GPU measurements above.
            if not args.no_reference:
                extra["reference_gpu_kernels_on_this_mi355x"] = guarded_leg("reference leg", lambda: cpu_baseline_reference_gpu_kernels_in_child(log_n, batch, args.commit_cols, args.commit_log_n))
            if not args.no_commit:
                extra["commit_cpu_baseline"] = guarded_leg("cpu_baseline_commit", lambda: cpu_baseline_commit(args.commit_cols, args.commit_log_n))
                if "ms" in extra["commit_cpu_baseline"]:
                    extra["commit_speedup_vs_cpu_baseline"] = extra["commit_cpu_baseline"]["ms"] / extra["commit_ms"]
            if not args.no_prove and prove_inputs is not None:
                extra["prove_cpu_baseline"] = guarded_leg("cpu_baseline_prove", lambda: cpu_baseline_prove(*prove_inputs))
                if "value" in extra["prove_cpu_baseline"]:
                    extra["prove_speedup_vs_cpu_baseline"] = extra["prove_cpu_baseline"]["value"] / extra["prove"]["prove_ms"]
        # The WHOLE metric of BASELINE.json ("prove() wall-clock + NTTs/sec", leaves hashed/s as fraction of the HBM roofline) as
        # scalars: the LAST keys of the line (a reader that keeps only the tail of the line still has them), and once more inside
        # `roofline` / `cpu_baseline`, which readers that keep the contract's objects keep whole. Stage names follow the reference's
        # timing labels (plonk/prover.rs:66-233: "to compute wire polynomials / wires commitment / partial products / quotient ...").
        if args.dry_ranks:
            out["dry_ranks"] = dry_ranks_checks(pg, _lib, ctx, args, out, extra)
        pr, fl = extra.get("prove") or {}, extra.get("prove_in_flight") or {}
        head = {
            "ntts_per_s": out["value"],
            "ntt_hbm_frac": out["roofline"]["frac"],
            "prove_ms": pr.get("prove_ms"),
            "prove_proofs_per_s": (1e3 / pr["prove_ms"]) if pr.get("prove_ms") else None,
            "prove_proofs_per_s_in_flight": fl.get("proofs_per_s"),
            "prove_in_flight": fl.get("in_flight"),
            "prove_wires_commitment_ms": (pr.get("stage_ms") or {}).get("wires commitment"),
            "prove_quotient_polys_ms": (pr.get("stage_ms") or {}).get("quotient polys"),
            "commit_ms": extra.get("commit_ms"),
            "merkle_leaves_per_s": extra.get("merkle_leaves_per_s"),
            "commit_hbm_frac": extra.get("commit_hbm_frac"),
            "commit_cpu_baseline_ms": (extra.get("commit_cpu_baseline") or {}).get("ms"),
            "prove_cpu_baseline_ms": (extra.get("prove_cpu_baseline") or {}).get("value"),
        }
        for k in ("prove_ms", "prove_proofs_per_s_in_flight", "commit_ms", "merkle_leaves_per_s", "commit_hbm_frac"):
            out["roofline"][k] = head[k]
        if out["cpu_baseline"]:
            out["cpu_baseline"]["commit_ms"] = head["commit_cpu_baseline_ms"]
            out["cpu_baseline"]["prove_ms"] = head["prove_cpu_baseline_ms"]
        out.update(head)  # after "extra": the line ends with these
        print(json.dumps(out), flush=True)
    buf.free()
    ctx.close()
    dist.barrier()  # rank 0 runs the extra legs; leave the group together
    dist.close()


def bench_prove(pg, ctx, dist, degree_bits, num_wires, reps):
    """prove() (plonk/prover.rs:40-233) per rank on its own synthetic circuit instance of the ed25519
    proof's shape; every rank proves `reps` independent proofs (configs[4]: one proof per GPU, no
    collective). The witness and the preprocessed commitment are resident before the timed region.
    Rank 0's last proof gets oracle-free self-checks outside the timed region; a proof of this very shape
    is verified by the oracle's verifier in tests/test_gpu_prove.py."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_circuit
    from plonky2_gpu_amd.challenger import hash_no_pad

    # At the ed25519 shape the circuit declares the whole 25-gate table of the ed25519 circuit (6 selector groups, 231
    # constraints): the quotient stage evaluates every declared gate at every LDE point, so it costs what the real
    # circuit's does, and the witness (rows of Noop / Constant / PublicInput / Arithmetic{20}) still satisfies it.
    table = "ed25519" if num_wires == 234 else "mini"
    circuit, wires, pis = synth_circuit.make(degree_bits, num_wires=num_wires, num_routed=80, num_constants=8, seed=1 + dist.rank,
                                             gate_table=table)
    synth_circuit.set_public_input_row(wires, hash_no_pad(ctx, pis))
    # gl_circuit_create: preprocessed commitment, circuit digest, run-time compiled gates — once per circuit (the
    # ed25519 gate kernel comes precompiled from build(): plonky2_gpu_amd/kernel_cache/)
    nc = pg.NativeCircuit(ctx, dict(circuit, circuit_digest=None))
    d_wires = pg.DeviceBuffer.from_host(ctx, np.ascontiguousarray(wires))
    nc.prove_bytes(d_wires, pis)  # warm-up: table builds, allocator
    ctx.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        data = nc.prove_bytes(d_wires, pis)  # gl_prove: the whole of prove() in one native call
    ctx.synchronize()
    dist.barrier()
    mine = time.perf_counter() - t0
    elapsed = dist.max(mine)
    # per rank: its own rate, and the first entry of its proof's wires cap (the proof starts with the cap, proof.rs /
    # util/serialization.rs:674-689) gathered over the group's backend — RCCL when every rank has a device of its own
    rates = dist.gather_caps(np.array([[int(reps / mine * 1e6), 0, 0, 0]], dtype=np.uint64))
    caps0 = dist.gather_caps(np.frombuffer(data[:32], dtype="<u8").reshape(1, 4).astype(np.uint64))
    timing = {}
    again = nc.prove_bytes(d_wires, pis, timing)  # one more with per-stage synchronisation for the breakdown
    res = None
    if dist.rank == 0:
        # Self-checks that need no oracle (only the cpu_baseline leg may touch it): the proof parses with the
        # product's reader into the shape the circuit dictates, re-serialises to the same bytes, and is
        # identical from run to run. Its VALIDITY at this very shape is established where the oracle is allowed:
        # tests/test_gpu_prove.py::test_full_size_proof_bytes_equal_the_c_oracle (run with -m gpu).
        parsed = pg.serialization.proof_from_bytes(data, circuit)
        if pg.serialization.proof_to_bytes(parsed) != data or again != data:
            raise SystemExit("bench: the proof does not round-trip through the wire format or is not deterministic")
        res = {
            "workload": f"configs[3] shape, synthetic circuit: n=2^{degree_bits}, {num_wires} wires (80 routed), 88 preprocessed polys, "
                        f"2 challenges, rate 8, cap_height 4, FRI arities {circuit['fri_params']['reduction_arity_bits']}, 28 queries, "
                        f"16 PoW bits; gate list: " + ("the 25 gates / 6 selector groups / 231 constraints of the ed25519 circuit, "
                        "all evaluated at every LDE point as in the real circuit; rows instantiate Noop/Constant/PublicInput/"
                        "Arithmetic(20) (a witness for the other kinds needs the Rust circuit builder)" if table == "ed25519" else
                        "Noop/Constant/PublicInput/Arithmetic") + "; witness + preprocessed commitment resident",
            "gate_table": table,
            "prove_ms": elapsed / reps * 1e3,
            "proofs_per_s_all_gpus": dist.world * reps / elapsed,
            "proofs_per_s_per_rank": [int(r[0, 0]) / 1e6 for r in rates],
            "wires_cap0_per_rank": [[hex(int(x)) for x in c[0]] for c in caps0],
            "proof_bytes": len(data),
            "prover": "gl_prove (native host logic, csrc/prove.hip)",
            "stage_ms": {k: round(v, 3) for k, v in timing.items()},
            "self_checks": "parses, re-serialises identically, deterministic across runs",
            "validity_checked_by": "tests/test_gpu_prove.py::test_full_size_proof_bytes_equal_the_c_oracle",
        }
    d_wires.free()
    nc.close()  # the circuit's working buffers (one proof's worth of HBM) go back before the next leg
    # rank 0 at N = 1 hands its circuit, witness and proof to the CPU leg (cpu_baseline_prove), which proves the same thing
    return res, ((circuit, wires, pis, data) if dist.rank == 0 and dist.world == 1 else None)


def dry_ranks_checks(pg, _lib, ctx, args, out, extra):
    """--dry-ranks, rank 0, after the group's measurements: the N-rank line against what ONE rank computes alone. Raises (non-zero
    exit, no line) when an invariant fails. No RCCL claim is attached: the transport was a host backend; what ran is everything else
    — rank launch and rendezvous, sharding, device tensors over library-owned pointers, the exchange's bookkeeping, the gathers."""
    import hashlib

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_circuit
    from plonky2_gpu_amd.challenger import hash_no_pad
    from plonky2_gpu_amd.dist import shard_range

    W = args.gpus
    checks = {"ranks": out["config"]["ranks"], "transport": out["config"]["rank_sync_backend"]}
    if out["config"]["ranks"] != W or out["n_gpus"] != W:
        raise SystemExit(f"dry-ranks: the line reports {out['config']['ranks']} ranks, {W} were asked for")
    if not args.no_prove:
        pr = extra["prove"]
        if len(pr["wires_cap0_per_rank"]) != W or len(pr["proofs_per_s_per_rank"]) != W:
            raise SystemExit("dry-ranks: the prove leg did not gather one entry per rank")
        table = "ed25519" if args.prove_wires == 234 else "mini"
        for r in range(W):  # rank r proved the circuit of seed 1 + r: the same proof made here, alone, starts with the same cap
            circuit, wires, pis = synth_circuit.make(args.prove_degree_bits, num_wires=args.prove_wires, num_routed=80, num_constants=8, seed=1 + r, gate_table=table)
            synth_circuit.set_public_input_row(wires, hash_no_pad(ctx, pis))
            nc = pg.NativeCircuit(ctx, dict(circuit, circuit_digest=None))
            data = nc.prove_bytes(np.ascontiguousarray(wires), pis)
            nc.close()
            alone = [hex(int(x)) for x in np.frombuffer(data[:32], dtype="<u8")]
            if alone != pr["wires_cap0_per_rank"][r]:
                raise SystemExit(f"dry-ranks: rank {r}'s wires cap {pr['wires_cap0_per_rank'][r]} differs from the one-rank proof's {alone}")
        checks["per_rank_wires_caps_equal_the_one_rank_proofs"] = True
    if not args.no_commit and "sharded_commit" in extra and "exchange" in extra["sharded_commit"]:
        sc, cols, log_n = extra["sharded_commit"], args.commit_cols, args.commit_log_n
        if sc["cap0"] != extra["cap0"] or not sc["deterministic"]:
            raise SystemExit("dry-ranks: the column-sharded commit's cap differs from the one-GPU commit's")
        n_ext = (1 << log_n) << 3
        lo, hi = shard_range(cols, W, 0)
        want = 8 * (hi - lo) * (n_ext // W) * (W - 1)  # 8 * (my columns) * (leaves per rank) * (W - 1)
        if sc["exchange"]["bytes_sent_per_rank"] != want or sc["exchange"]["links_used_per_rank"] != W - 1:
            raise SystemExit(f"dry-ranks: bytes_sent_per_rank {sc['exchange']['bytes_sent_per_rank']} != {want}")
        checks["sharded_commit_cap_equals_the_one_gpu_commit"] = True
        checks["bytes_sent_per_rank_equals_8_cols_leaves_peers"] = want
    checks["line_sha256_prefix"] = hashlib.sha256(json.dumps(extra.get("prove", {}).get("wires_cap0_per_rank", [])).encode()).hexdigest()[:16]
    return checks


def bench_prove_in_flight(pg, device, dist, degree_bits, num_wires, in_flight, reps):
    """Proofs per second of ONE GPU with `in_flight` proofs at a time (the per-GPU factor of configs[4], a THROUGHPUT config):
    that many host threads, each with its own context (streams, workspace: csrc/capi.hip CtxState), circuit handle and witness
    buffer, each proving `reps` proofs back to back. One proof alone leaves the chip idle in its latency-bound phases (the
    transcript's serial sponge, tree layers below 2^16 nodes, openings, host round trips); a second proof fills them. Every proof
    must equal, byte for byte, what the same witness gives on one context alone. All ranks run it at the same time (each on its GPU)."""
    import threading

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_circuit
    from plonky2_gpu_amd.challenger import hash_no_pad

    table = "ed25519" if num_wires == 234 else "mini"
    ctxs = [pg.Context(device) for _ in range(in_flight)]
    circuit, wires, pis = synth_circuit.make(degree_bits, num_wires=num_wires, num_routed=80, num_constants=8, seed=1 + dist.rank, gate_table=table)
    synth_circuit.set_public_input_row(wires, hash_no_pad(ctxs[0], pis))
    wires = np.ascontiguousarray(wires)
    ncs = [pg.NativeCircuit(c, dict(circuit, circuit_digest=None)) for c in ctxs]
    bufs = [pg.DeviceBuffer.from_host(c, wires) for c in ctxs]
    expect = ncs[0].prove_bytes(bufs[0], pis)
    for i in range(in_flight):  # warm-up of every context's pool, and the reference bytes
        if ncs[i].prove_bytes(bufs[i], pis) != expect:
            raise SystemExit("bench: contexts disagree on the proof of one witness")
        ctxs[i].synchronize()
    bad, done_at = [], [0.0] * in_flight

    def work(i):
        for _ in range(reps):
            if ncs[i].prove_bytes(bufs[i], pis) != expect:
                bad.append(i)
        ctxs[i].synchronize()
        done_at[i] = time.perf_counter()

    walls = []
    for attempt in range(2):
        threads = [threading.Thread(target=work, args=(i,)) for i in range(in_flight)]
        dist.barrier()
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        walls.append(dist.max(max(done_at) - t0))
    if bad:
        raise SystemExit("bench: a proof made with %d proofs in flight differs from the proof made alone" % in_flight)
    for b in bufs:
        b.free()
    for nc in ncs:
        nc.close()
    for c in ctxs:
        c.close()
    best = min(walls)
    return {"in_flight": in_flight, "proofs_per_thread": reps, "proofs_per_s": dist.world * in_flight * reps / best,
            "ms_per_proof": best / (in_flight * reps) * 1e3, "every_proof_equals_the_one_made_alone": True,
            "note": "host threads x own context x own circuit handle on each GPU; all GPUs at the same time"}


def bench_sharded_commit(pg, ctx, dist, cols, log_n, rate_bits=3, cap_height=4, iters=2):
    """N > 1 only: ONE commit of configs[2] with its columns sharded over the ranks (second row; plonky2_gpu_amd/dist.py
    ShardedCommitPlan): rank r transforms its slice of the columns, the path's one exchange regroups leaf ranges (every rank sends
    each peer the peer's leaf range of its own columns, point to point: on an xGMI mesh every link carries one pair), each rank
    hashes its leaves and its cap subtrees, the cap is gathered. Reported with the bytes that cross the links, so that the first run
    on a node with a device per rank reads as xGMI GB/s without a code change; with ranks sharing a device (gloo, staged through
    the host) the same figures describe the host path instead."""
    from plonky2_gpu_amd.dist import ShardedCommitPlan

    W = dist.world
    if W & (W - 1) or W > (1 << cap_height):
        return {"absent_because": f"{W} ranks: the sharded commit wants a power of two, at most 2^cap_height"}
    n, n_ext = 1 << log_n, (1 << log_n) << rate_bits
    plan = ShardedCommitPlan(dist, ctx, cols, log_n, rate_bits, cap_height)
    mine = plan.mine
    d_vals = pg.DeviceBuffer(ctx, max(mine, 1) * n)
    d_work = pg.DeviceBuffer(ctx, max(mine, 1) * n)
    for c0 in range(0, cols, 16):  # the very matrix of bench_commit (same chunks of the SplitMix64 stream), my columns of it
        k = min(16, cols - c0)
        lo, hi = max(c0, plan.col_lo), min(c0 + k, plan.col_hi)
        if lo < hi:
            block = splitmix64_field(k * n, start=(1 << 50) + c0 * n * 2).reshape(k, n)
            d_vals.upload(block[lo - c0:hi - c0], (lo - plan.col_lo) * n)
    from plonky2_gpu_amd import _lib

    times, caps = [], []
    for it in range(iters + 1):
        _lib.call("gl_memcpy_d2d", d_work.ptr, d_vals.ptr, max(mine, 1) * n * 8, ctx.ptr)
        ctx.synchronize()
        dist.barrier()
        t = time.perf_counter()
        res = plan.commit(d_work)
        ctx.synchronize()
        dist.barrier()
        dt = dist.max(time.perf_counter() - t)
        if it:
            times.append(dt)
        caps.append(np.asarray(res.cap).tobytes())
    ms = float(np.median(times)) * 1e3
    sent_per_rank = 8 * mine * plan.L * (W - 1)          # my columns, every peer's leaf range
    per_link = 8 * mine * plan.L                          # what one (ordered) pair of ranks moves one way
    out = {"workload": f"configs[2] as ONE commit over {W} ranks: {cols} columns x 2^{log_n} rows, rate 8, cap_height {cap_height}; columns {plan.bounds}",
           "commit_ms": ms, "merkle_leaves_per_s": n_ext / (ms * 1e-3), "deterministic": all(c == caps[0] for c in caps),
           "cap0": [hex(int(x)) for x in np.frombuffer(caps[0], dtype=np.uint64)[:4]],
           "exchange": {"bytes_sent_per_rank": sent_per_rank, "bytes_per_link_one_way": per_link, "links_used_per_rank": W - 1,
                        "bytes_all_ranks": sent_per_rank * W,
                        "GBps_per_rank_if_it_took_the_whole_commit": sent_per_rank / (ms * 1e-3) / 1e9,
                        "GBps_per_link_if_it_took_the_whole_commit": per_link / (ms * 1e-3) / 1e9,
                        "xgmi_link_peak_GBps": 153.0, "note": "the exchange is posted per 16-column chunk under the remaining LDE; these rates divide "
                        "by the WHOLE commit time and are lower bounds of what the links carried",
                        "transport": "RCCL point to point between device buffers" if dist.backend == "nccl" else f"{dist.backend}: staged through host memory (ranks share a device)"}}
    plan.free()
    d_vals.free()
    d_work.free()
    return out


def bench_commit(pg, _lib, ctx, cols, log_n, rate_bits=3, cap_height=4, iters=3):
    """PolynomialBatch::from_values on configs[2]; leaf-major copy included (reference contract)."""
    n = 1 << log_n
    n_ext = n << rate_bits
    d_vals = pg.DeviceBuffer(ctx, cols * n)
    chunk = 16
    for c0 in range(0, cols, chunk):  # [cols][n] uniform field elements from the same SplitMix64 stream, segment 2^50
        k = min(chunk, cols - c0)
        d_vals.upload(splitmix64_field(k * n, start=(1 << 50) + c0 * n * 2), c0 * n)
    d_work = pg.DeviceBuffer(ctx, cols * n)
    d_lde = pg.DeviceBuffer(ctx, cols * n_ext)
    d_leaves = pg.DeviceBuffer(ctx, cols * n_ext)
    d_dig = pg.DeviceBuffer(ctx, 4 * 2 * (n_ext - (1 << cap_height)))
    d_cap = pg.DeviceBuffer(ctx, 4 << cap_height)
    times, caps = [], []
    for it in range(iters + 1):
        _lib.call("gl_memcpy_d2d", d_work.ptr, d_vals.ptr, cols * n * 8, ctx.ptr)
        ctx.synchronize()
        e0, e1 = pg.Event(), pg.Event()
        e0.record(ctx)
        _lib.call("gl_commit_from_values", d_work.ptr, cols, log_n, rate_bits, cap_height, 0, 7, d_lde.ptr, d_leaves.ptr,
                  d_dig.ptr, d_cap.ptr, ctx.ptr)
        e1.record(ctx)
        ctx.synchronize()
        if it:
            times.append(e1.elapsed_ms_since(e0))
        caps.append(d_cap.download().tobytes())
    assert all(c == caps[0] for c in caps), "commit is not deterministic"
    ms = float(np.median(times))
    # the same commit without the leaf-major copy (d_leaves = NULL): what gl_prove uses — the quotient kernel
    # and the batched openings read the column-major LDE directly
    times_nl = []
    for it in range(iters):
        _lib.call("gl_memcpy_d2d", d_work.ptr, d_vals.ptr, cols * n * 8, ctx.ptr)
        ctx.synchronize()
        e0, e1 = pg.Event(), pg.Event()
        e0.record(ctx)
        _lib.call("gl_commit_from_values", d_work.ptr, cols, log_n, rate_bits, cap_height, 0, 7, d_lde.ptr, None, d_dig.ptr, d_cap.ptr,
                  ctx.ptr)
        e1.record(ctx)
        ctx.synchronize()
        times_nl.append(e1.elapsed_ms_since(e0))
        assert d_cap.download().tobytes() == caps[0], "the cap must not depend on the leaf-major copy"
    ms_nl = float(np.median(times_nl))
    # the stages of the same commit one at a time (HIP events around each entry point; inside the commit they overlap little:
    # every kernel here fills the register file): values -> coefficients, coset LDE, leaf hashing + tree over the LDE's columns
    def timed(fn):
        ts = []
        for _ in range(iters):
            ctx.synchronize()
            e0, e1 = pg.Event(), pg.Event()
            e0.record(ctx)
            fn()
            e1.record(ctx)
            ctx.synchronize()
            ts.append(e1.elapsed_ms_since(e0))
        return float(np.median(ts))

    _lib.call("gl_memcpy_d2d", d_work.ptr, d_vals.ptr, cols * n * 8, ctx.ptr)
    stage = {"ifft (values -> coefficients)": timed(lambda: _lib.call("gl_ntt_batch", d_work.ptr, cols, log_n, n, 1, 0, ctx.ptr)),
             "coset LDE (bit-reversed)": timed(lambda: _lib.call("gl_coset_lde_batch", d_work.ptr, d_lde.ptr, cols, log_n, rate_bits, 7, n, n_ext, ctx.ptr)),
             "leaf hashing + tree layers": timed(lambda: _lib.call("gl_merkle_tree_from_columns", d_lde.ptr, cols, n_ext, n_ext, cap_height, d_dig.ptr, d_cap.ptr, ctx.ptr))}
    alg = 8.0 * cols * n + 8.0 * cols * n_ext + 32.0 * (2 * (n_ext - (1 << cap_height)) + (1 << cap_height))
    perms = n_ext * ((cols + 7) // 8) + n_ext - (1 << cap_height)
    for b in (d_vals, d_work, d_lde, d_leaves, d_dig, d_cap):
        b.free()
    return {
        "commit_workload": f"configs[2]: from_values {cols} cols x 2^{log_n} rows, rate 8, cap_height {cap_height}, "
                           f"leaf-major copy included",
        "commit_ms": ms,
        "commit_ms_without_leaf_major_copy": ms_nl,
        "commit_stage_ms_one_at_a_time": {k: round(v, 3) for k, v in stage.items()},
        "commit_stage_ms_sum": round(sum(stage.values()), 3),
        "merkle_leaves_per_s": n_ext / (ms * 1e-3),
        "poseidon_permutations_per_s": perms / (ms * 1e-3),
        "commit_algorithmic_GBps": alg / (ms * 1e-3) / 1e9,
        "commit_hbm_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "cap0": [hex(int(x)) for x in np.frombuffer(caps[0], dtype=np.uint64)[:4]],
    }


if __name__ == "__main__":
    main()
