"""The N>1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous. Each rank commits its
shard of independent units (here with the CPU oracle standing in for the GPU, which this container
lacks); the caps gathered over torch.distributed must equal the single-process result."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["REPO_ROOT"])
import numpy as np
from plonky2_gpu_amd.dist import ProverGroup, shard_range
from oracle import oracle as o

g = ProverGroup(backend="gloo")
n_units = 5  # independent commitments ("proofs"), sharded over the ranks
lo, hi = shard_range(n_units, g.world, g.rank)
caps = []
for u in range(lo, hi):
    vals = o.random_field((3, 16), seed=100 + u)
    caps.append(o.canon(o.commit_from_values(vals, 3, 2)["cap"]))
mine = np.stack(caps) if caps else np.zeros((0, 4, 4), dtype=np.uint64)
# pad to the max shard size so that all_gather sees equal shapes
width = -(-n_units // g.world)
pad = np.zeros((width, 4, 4), dtype=np.uint64); pad[: len(mine)] = mine
g.barrier()
t = g.max(float(g.rank + 1))
allcaps = g.gather_caps(pad.reshape(width * 4, 4))
if g.rank == 0:
    out = []
    for r, c in enumerate(allcaps):
        l, h = shard_range(n_units, g.world, r)
        out.append(c.reshape(width, 4, 4)[: h - l])
    res = np.concatenate(out)
    print("RESULT " + json.dumps({"max": t, "caps": res.tolist(), "units": g.sum(hi - lo)}))
else:
    g.sum(hi - lo)
g.close()
'''


def test_shard_range_is_a_partition():
    from plonky2_gpu_amd.dist import shard_range

    for n in (0, 1, 5, 8, 64, 135):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_gloo_gather_caps(oracle, tmp_path):
    import json

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0]
    res = json.loads(line[len("RESULT "):])
    assert res["max"] == 2.0 and res["units"] == 5.0
    exp = [oracle.canon(oracle.commit_from_values(oracle.random_field((3, 16), seed=100 + u), 3, 2)["cap"]).tolist()
           for u in range(5)]
    assert res["caps"] == exp
