#!/bin/bash
# the whole GPU suite as the driver runs it, timed, then smoke
mkdir -p gpurun_out
( time python3 -m pytest tests/ -x -q -m gpu --durations=25 ) > gpurun_out/r05_suite.log 2>&1
echo "rc=$?" >> gpurun_out/r05_suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1
echo "rc=$?" >> gpurun_out/r05_smoke.log
tail -n 45 gpurun_out/r05_suite.log; tail -n 3 gpurun_out/r05_smoke.log
