#!/bin/bash
# round 5, second run: interleaved-multiplication experiment, the new GPU tests (upstream gate kinds, ABI guard, raw salts), the bench with its new CPU legs
mkdir -p gpurun_out
timeout 300 ./tools/experiments/bin/mul_interleave > gpurun_out/r05_mul_interleave.jsonl 2>&1
python3 -m pytest tests/test_gpu_prove.py -k "recursion or header or blinded" -x -q -m gpu > gpurun_out/r05_new_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_new_gpu_tests.log
( time python3 bench.py ) > gpurun_out/r05_bench_first.json 2> gpurun_out/r05_bench_first.err
cat gpurun_out/r05_mul_interleave.jsonl; tail -n 4 gpurun_out/r05_new_gpu_tests.log; tail -n 5 gpurun_out/r05_bench_first.err
