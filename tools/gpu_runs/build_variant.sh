#!/bin/bash
# build_variant.sh NAME "-DFLAG=1 ..." : the library with ntt.hip / ntt_direct.hip compiled under extra flags ->
# gpurun_in/NAME/plonky2_gpu_amd/libplonky2_hip.so (for tools/gpu_runs/ntt_variants_ab.sh; gpurun_in/ is scratch, not committed)
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/plonky2_gpu_amd/csrc
B=$C/build/variant_$NAME
mkdir -p $B $R/gpurun_in/$NAME/plonky2_gpu_amd
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
make -s -C $C all
for s in ntt ntt_direct; do $CXX $FLAGS -c $C/$s.hip -o $B/$s.o & done; wait
OBJS=$(for o in merkle plonk fri gate_jit gate_emit prove capi; do echo $C/build/$o.o; done)
/opt/rocm/bin/hipcc -shared -fPIC -fvisibility=hidden --offload-arch=gfx950 $B/ntt.o $B/ntt_direct.o $OBJS -lhiprtc -Wl,--version-script=$C/exports.map -o $R/gpurun_in/$NAME/plonky2_gpu_amd/libplonky2_hip.so
echo built $NAME
