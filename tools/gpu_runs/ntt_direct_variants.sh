#!/bin/bash
# Diagnostic builds of the direct NTT passes, timed beside the product library (never shipped; their results are wrong by design).
# Build here (no GPU needed):   bash tools/gpu_runs/ntt_direct_variants.sh build
# Run on the GPU box:           gpurun -- 'bash tools/gpu_runs/ntt_direct_variants.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R" || exit 1
V=$R/gpurun_in/ntt_variants
VARIANTS="${VARIANTS:-SAME_LOADS SAME_STORES NO_BARRIER TAIL_FRONT SAME_LOADS+SAME_STORES SAME_LOADS+SAME_STORES+NO_BARRIER}"
if [ "$1" = build ]; then
    make -C plonky2_gpu_amd/csrc > /dev/null || exit 1
    for v in $VARIANTS; do
        mkdir -p $V/$v
        D=""; for f in ${v//+/ }; do D="$D -DDIRECT_DIAG_$f"; done
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 $D -I plonky2_gpu_amd/csrc -c tools/experiments/ntt_direct_diag.hip -o /tmp/ntt_direct_$v.o &&
            /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/ntt_direct_$v.o $(ls plonky2_gpu_amd/csrc/build/*.o | grep -v "ntt_direct\|debug_") -lhiprtc -o $V/$v/libplonky2_hip.so || exit 1
    done
    exit 0
fi
O=gpurun_out/ntt_direct_variants; mkdir -p $O; rm -f $O/ab.jsonl
for rep in 1 2; do
    TAG=product python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
    for v in $VARIANTS; do
        TAG=$v PLONKY2_HIP_LIBRARY=$V/$v/libplonky2_hip.so python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
    done
done
cat $O/ab.jsonl
