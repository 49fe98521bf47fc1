#!/bin/bash
# round 6: the whole GPU suite with durations, then the timeline and in-flight figures of the current build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=25 2>&1 | tail -45 | tee $O/r06_gpu_suite.txt
timeout 300 python tools/bench_prove.py 18 234 3 0 > $O/r06_bench_prove_b.json 2> $O/r06_suite.err; cat $O/r06_bench_prove_b.json
timeout 600 python tools/bench_inflight.py 18 8 1,2,3 0 > $O/r06_inflight_b.json 2>> $O/r06_suite.err; cat $O/r06_inflight_b.json
bash tools/gpu_runs/r06_timeline.sh r06_timeline_b | head -12
