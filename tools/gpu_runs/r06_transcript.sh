#!/bin/bash
# round 6: the device-resident transcript — its unit tests, the proof suites (bytes against the C oracle), then prove timing and timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fri.py tests/test_gpu_prove.py tests/test_gpu_contexts.py -x -q -m gpu 2>&1 | tail -15 | tee $O/r06_transcript_tests.txt
timeout 300 python tools/bench_prove.py 18 234 3 0 > $O/r06_bench_prove_transcript.json 2> $O/r06_transcript.err; cat $O/r06_bench_prove_transcript.json; tail -3 $O/r06_transcript.err
timeout 600 python tools/bench_inflight.py 18 8 1,2,3 0 > $O/r06_inflight_transcript.json 2>> $O/r06_transcript.err; cat $O/r06_inflight_transcript.json
bash tools/gpu_runs/r06_timeline.sh r06_timeline_transcript | tail -60
