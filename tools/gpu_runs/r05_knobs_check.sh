#!/bin/bash
# round 5: after moving the generator's switches to the diagnostic build: the tests that use them, and a short campaign
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05knobs; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_reference_quotient.py tests/test_gpu_plonk.py tests/test_gate_jit_units.py tests/test_abi.py -x -q --durations=4 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 9 $O/tests.log
timeout 900 python3 tests/fuzz_gate_jit.py 24 $(date +%s) > $O/fuzz.log 2>&1; echo "fuzz rc=$?" >> $O/fuzz.log; tail -n 3 $O/fuzz.log; grep -c "second={'PLONKY2_HIP_JIT_FUSE': '0'" $O/fuzz.log
