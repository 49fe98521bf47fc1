#!/bin/bash
# round 6: counters + kernel stats of the final kernels (tools/gpu_runs/pmc_passes.sh, TAG=r06), then the bench line on the SAME device
export TAG=r06
bash tools/gpu_runs/pmc_passes.sh > gpurun_out/r06_pmc_passes.log 2>&1
cd "$GRAFT_REPO_ROOT" || exit 1
python3 bench.py > gpurun_out/r06pmc/bench.json 2> gpurun_out/r06pmc/bench.err
tail -n 25 gpurun_out/r06_pmc_passes.log
python3 -c "import json; d=json.loads(open('gpurun_out/r06pmc/bench.json').read()); print(d['value'], d['roofline']['ms'], d['roofline']['frac'], d['roofline']['traffic'], d['extra'].get('commit_ms'), d['extra']['prove']['prove_ms'])"
ls gpurun_out/r06pmc | head -40
