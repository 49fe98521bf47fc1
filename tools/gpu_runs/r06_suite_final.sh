#!/bin/bash
# round 6: the whole GPU suite on the final tree, with durations
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2700 python -m pytest tests -x -q -m gpu --durations=25 2>&1 | tail -40 | tee gpurun_out/r06_gpu_suite_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
