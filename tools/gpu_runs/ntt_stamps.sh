#!/bin/bash
# Where a wavefront of the NTT passes spends its cycles (tools/ntt_stamps.py on a -DPLONKY2_NTT_STAMPS build of the library).
# Build here:  bash tools/gpu_runs/ntt_stamps.sh build      Run:  gpurun -- 'bash tools/gpu_runs/ntt_stamps.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R" || exit 1
V=$R/gpurun_in/ntt_variants/STAMPS
if [ "$1" = build ]; then
    make -C plonky2_gpu_amd/csrc > /dev/null || exit 1
    mkdir -p $V
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DPLONKY2_NTT_STAMPS -c plonky2_gpu_amd/csrc/ntt.hip -o /tmp/ntt_stamps.o &&
        /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/ntt_stamps.o plonky2_gpu_amd/csrc/build/{merkle,plonk,fri,gate_jit,prove,capi}.o -lhiprtc -o $V/libplonky2_hip.so
    exit $?
fi
O=gpurun_out/ntt_stamps; mkdir -p $O
python3 tools/ntt_stamps.py $V/libplonky2_hip.so $O/stamps.jsonl > $O/stamps.txt 2>&1
cat $O/stamps.txt
