#!/bin/bash
# Diagnostic builds of the NTT passes, timed beside the product library (never shipped, results are wrong by design):
#   SKELETON  -DPLONKY2_NTT_SKELETON  every load, LDS exchange, barrier and store, no radix rounds
# Build here (no GPU needed):   bash tools/gpu_runs/ntt_variant_builds.sh build
# Run on the GPU box:           gpurun -- 'bash tools/gpu_runs/ntt_variant_builds.sh'
# -> gpurun_out/ntt_variants/ab.jsonl (profiles/r02_ntt_skeleton_experiment.jsonl is one such file)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R" || exit 1
V=$R/gpurun_in/ntt_variants
if [ "$1" = build ]; then
    make -C plonky2_gpu_amd/csrc > /dev/null || exit 1
    for v in SKELETON; do
        mkdir -p $V/$v
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DPLONKY2_NTT_$v -c plonky2_gpu_amd/csrc/ntt.hip -o /tmp/ntt_$v.o &&
            /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/ntt_$v.o plonky2_gpu_amd/csrc/build/{merkle,plonk,fri,gate_jit,prove,capi}.o -lhiprtc -o $V/$v/libplonky2_hip.so || exit 1
    done
    exit 0
fi
O=gpurun_out/ntt_variants; mkdir -p $O; rm -f $O/ab.jsonl
for rep in 1 2; do
    TAG=product python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
    TAG=skeleton PLONKY2_HIP_LIBRARY=$V/SKELETON/libplonky2_hip.so python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
    TAG=skeleton PLONKY2_NTT_WG_PER_CU=1 PLONKY2_HIP_LIBRARY=$V/SKELETON/libplonky2_hip.so python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
    TAG=product PLONKY2_NTT_WG_PER_CU=1 python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
