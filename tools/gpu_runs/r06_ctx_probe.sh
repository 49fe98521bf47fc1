cd "$GRAFT_REPO_ROOT"
python tools/experiments/ctx_overlap_probe.py
GPU_MAX_HW_QUEUES=8 python tools/experiments/ctx_overlap_probe.py
GPU_MAX_HW_QUEUES=8 python tools/bench_inflight.py 18 8 1,2,3 0
