#!/bin/bash
# (a record of a run made BEFORE the generator's switches moved to the diagnostic build, some with knobs of scratch builds that no longer exist:
# to repeat what still applies, export PLONKY2_HIP_LIBRARY=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so)
# round 5: where the fused gate kernels' time goes: the same arithmetic without wire loads / without the alpha table / without both
# (experiment knobs of a scratch build, wrong results), one device
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05where; mkdir -p $O; : > $O/summary.txt
V=$GRAFT_REPO_ROOT/plonky2_gpu_amd/kernel_cache_variants
for rep in 1 2; do
echo "as generated: $(timeout 400 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1 | grep -o '"compiled_ms": [0-9.]*')" | tee -a $O/summary.txt
echo "no wire loads: $(JITX_NOLOAD=1 PLONKY2_HIP_KERNEL_CACHE=$V/noload timeout 400 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1 | grep -o '"compiled_ms": [0-9.]*')" | tee -a $O/summary.txt
echo "no alpha table: $(JITX_NOALPHA=1 PLONKY2_HIP_KERNEL_CACHE=$V/noalpha timeout 400 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1 | grep -o '"compiled_ms": [0-9.]*')" | tee -a $O/summary.txt
echo "neither: $(JITX_NOLOAD=1 JITX_NOALPHA=1 PLONKY2_HIP_KERNEL_CACHE=$V/neither timeout 400 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1 | grep -o '"compiled_ms": [0-9.]*')" | tee -a $O/summary.txt
done
