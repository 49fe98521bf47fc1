#!/bin/bash
# round 6: contexts run concurrently (csrc/capi.hip CtxState) — the thread tests, then proofs/s with 1, 2, 3 proofs in flight
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "two_host_threads or stream2 or device_memory or recursion_shaped or smoke or small" 2>&1 | tail -5 > gpurun_out/r06_inflight_tests.txt
cat gpurun_out/r06_inflight_tests.txt
timeout 600 python tools/bench_inflight.py 18 8 1,2,3 0 > gpurun_out/r06_inflight.json 2> gpurun_out/r06_inflight.err; tail -3 gpurun_out/r06_inflight.err; cat gpurun_out/r06_inflight.json
timeout 600 python tools/bench_inflight.py 18 8 1,2 1 > gpurun_out/r06_inflight_shared.json 2>> gpurun_out/r06_inflight.err; cat gpurun_out/r06_inflight_shared.json
timeout 300 python tools/bench_prove.py 18 234 3 0 > gpurun_out/r06_bench_prove.json 2>> gpurun_out/r06_inflight.err; cat gpurun_out/r06_bench_prove.json
