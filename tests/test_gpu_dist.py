"""SURVEY.md §8(e), second row: ONE commitment whose columns are spread over the ranks, with the path's single
exchange between the LDE and the leaf hashing. Run here with 2 and 4 ranks sharing the one GPU (exchange staged
through host memory over gloo); every rank checks its part against the oracle's commit of the whole matrix
(tests/dist_sharded_commit.py)."""
import os
import subprocess
import sys

import pytest

from gpu_util import gpu  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,shape", [(2, (24, 10, 3, 4)), (4, (135, 8, 3, 4)), (2, (9, 12, 2, 1)), (4, (20, 6, 3, 2))])
def test_column_sharded_commit_equals_the_whole_commit(gpu, world, shape):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + world * 10 + shape[1]), os.path.join(ROOT, "tests", "dist_sharded_commit.py")] + [str(x) for x in shape]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert p.stdout.count(" ok") == world, p.stdout


def test_sharded_commit_argument_errors(gpu):
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd.dist import ProverGroup, sharded_commit_from_values

    g = ProverGroup()  # world 1
    d = pg.DeviceBuffer(gpu, 4 * 16)
    with pytest.raises(ValueError):
        sharded_commit_from_values(g, gpu, d, 1, 5, 5, 4, 3, 2)  # not the slice shard_range assigns
    sc = sharded_commit_from_values(g, gpu, d, 0, 4, 4, 4, 3, 2)  # world 1: the plain commit
    assert sc.leaf_lo == 0 and sc.leaves_per_rank == 128 and sc.cap.shape == (4, 4)


def test_rccl_route_with_one_rank(gpu):
    """backend "nccl" (= RCCL) on the one GPU there is: process group, collectives on device tensors, zero-copy tensors over
    the library's buffers, the sharded commit under that group (tests/dist_nccl_self.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_nccl_self.py")], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "nccl single-rank ok" in p.stdout


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_with_two_ranks_prints_one_line_for_the_whole_job(gpu, launcher):
    """bench.py exactly as the scaling run starts it — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` — and as `python bench.py --gpus N` (which
    starts its N ranks itself before anything touches the GPU). Two ranks share this box's one device (rank synchronisation then
    goes over gloo; with a device per rank it is RCCL): rank 0 prints ONE JSON line, n_gpus = 2, the value is the job's aggregate,
    the commit leg runs on rank 0, the prove leg on every rank, the CPU baseline only at N = 1."""
    import json

    bench = os.path.join(ROOT, "bench.py")
    tail = [bench, "--gpus", "2", "--steps", "2", "--warmup", "1"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(29500 + os.getpid() % 400)] + tail
    else:
        cmd = [sys.executable] + tail
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["metric"] and d["unit"] and d["value"] > 0 and d["config"]["ranks"] == 2
    assert d["cpu_baseline"] is None and d["roofline"]["frac"] > 0
    assert d["extra"]["commit_ms"] > 0 and d["extra"]["prove"]["proofs_per_s_all_gpus"] > 0
    assert len(d["extra"]["prove"]["proofs_per_s_per_rank"]) == 2 and len(d["extra"]["prove"]["wires_cap0_per_rank"]) == 2
    assert d["config"]["pairs_per_step"] >= 1 and d["value_windows"]["count"] >= 1
    # the column-sharded commit of configs[2] over the two ranks, with the bytes its one exchange moves
    sc = d["extra"]["sharded_commit"]
    assert sc["commit_ms"] > 0 and sc["deterministic"] and sc["cap0"] == d["extra"]["cap0"], sc  # same values, same cap as the one-GPU commit
    assert sc["exchange"]["links_used_per_rank"] == 1 and sc["exchange"]["bytes_per_link_one_way"] == 8 * 68 * (1 << 22)


def _device_count():
    import plonky2_gpu_amd as pg

    return pg.load().gl_device_count()


def test_two_devices_exchange_over_rccl(gpu):
    """What a one-GPU box cannot show: the column-sharded commit with each rank on a device of its own and the exchange
    between device buffers over RCCL/xGMI (zero-copy sends from the pack buffer, receives into the leaf block, chunk by chunk
    under the LDE), the cap all-gathered over RCCL. Skips on a box with one GPU; runs the day a multi-GPU box runs the suite."""
    if _device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % _device_count())
    for world, shape in [(2, (135, 16, 3, 4)), (2, (24, 10, 3, 4))] + ([(4, (135, 14, 3, 4))] if _device_count() >= 4 else []):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PLONKY2_DIST_BACKEND="nccl")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(29700 + world * 10 + shape[1]), os.path.join(ROOT, "tests", "dist_sharded_commit.py")] + [str(x) for x in shape]
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
        assert p.stdout.count(" ok") == world, p.stdout


def test_bench_on_two_devices_synchronises_over_rccl(gpu):
    """`bench.py --gpus 2` with a device per rank: rank synchronisation and the cap gather go over RCCL ("nccl"), the line reports
    per-rank proof rates and the gathered caps of the ranks' proofs."""
    import json

    if _device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % _device_count())
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-commit"], env=env,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["rank_sync_backend"] == "nccl" and d["config"]["devices_visible"] >= 2
    pr = d["extra"]["prove"]
    assert len(pr["proofs_per_s_per_rank"]) == 2 and len(pr["wires_cap0_per_rank"]) == 2


@pytest.mark.parametrize("world", [2, 4, 8])
def test_dry_ranks_line_holds_its_invariants(gpu, world):
    """`bench.py --gpus N --dry-ranks`: N ranks on the one device there is, over gloo, through the code a node with
    a device per rank runs except for the backend string — the ranks are started by bench.py itself before anything touches the GPU,
    every rank proves its own circuit, the column-sharded commit builds its send / receive tensors over the library's own device
    pointers (dist.device_tensor) and only the transport goes through host memory — and rank 0 then checks the line against what one
    rank computes alone: ranks == N, every rank's wires cap equals the cap of the same proof made alone, the sharded commit's cap
    equals the one-GPU commit's, bytes_sent_per_rank = 8 x (my columns) x (leaves per rank) x (N - 1). A failed invariant is a
    non-zero exit and no line. Small shapes: the point is the path, not the rate. No RCCL claim is attached to this mode."""
    import json

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PLONKY2_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-ranks", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--log-n", "16", "--windows", "0", "--commit-cols", "40", "--commit-log-n", "12", "--prove-degree-bits", "12", "--prove-reps", "2",
           "--prove-in-flight", "0", "--prove-larger", ""]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    dr = d["dry_ranks"]
    assert d["n_gpus"] == world and dr["ranks"] == world and dr["transport"] == "gloo"
    assert dr["per_rank_wires_caps_equal_the_one_rank_proofs"] is True and dr["sharded_commit_cap_equals_the_one_gpu_commit"] is True
    lo, hi = 0, 40 // world + (1 if 40 % world else 0)
    assert dr["bytes_sent_per_rank_equals_8_cols_leaves_peers"] == 8 * (hi - lo) * ((1 << 15) // world) * (world - 1)
    assert len(set(tuple(c) for c in d["extra"]["prove"]["wires_cap0_per_rank"])) == world  # every rank proved its own circuit
