#!/bin/bash
# round 6: the tests of the concurrent contexts and the new memory-contract entry points, the suites they touch, and one default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_contexts.py tests/test_gpu_prove.py tests/test_reference_quotient.py tests/test_gpu_stream2.py -x -q -m gpu 2>&1 | tail -15 | tee $O/r06_contexts_tests.txt
: # (bench line: see r06_final.sh)
