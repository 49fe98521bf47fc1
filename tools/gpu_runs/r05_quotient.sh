#!/bin/bash
# round 5: the gate kernels with wire loads as buffer loads and ACC as multiply-adds: parity, then the ed25519 quotient timing and the prove stages
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05quotient; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_plonk.py tests/test_reference_quotient.py tests/test_gpu_prove.py tests/test_cpp_prove.py tests/test_reference_dumps.py tests/test_gpu_reference_kernels.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 6 $O/tests.log
timeout 300 python3 tools/bench_quotient_ed25519.py 18 5 1 > $O/quotient.json 2> $O/quotient.err; tail -c 900 $O/quotient.json
timeout 300 python3 tools/bench_prove.py 18 234 5 1 1 > $O/prove.json 2> $O/prove.err; tail -c 1500 $O/prove.json
