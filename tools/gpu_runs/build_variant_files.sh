#!/bin/bash
# build_variant_files.sh NAME "-DFLAG=1 ..." file1 [file2 ..]: the library with the named csrc/*.hip compiled under extra flags ->
# gpurun_in/NAME/plonky2_gpu_amd/libplonky2_hip.so, loaded through PLONKY2_HIP_LIBRARY by the A/B scripts (gpurun_in/ is scratch,
# not committed, and the variant object trees are removed again: nothing of an A/B build ships with the product)
set -e
NAME=$1; FLAGS=$2; shift; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/plonky2_gpu_amd/csrc
B=$(mktemp -d /tmp/variant_$NAME.XXXX)
mkdir -p $R/gpurun_in/$NAME/plonky2_gpu_amd
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
make -s -C $C all
OBJS=""
for o in ntt ntt_direct merkle plonk fri gate_jit gate_emit prove capi; do
  if [[ " $* " == *" $o "* ]]; then (cd $C && $CXX $FLAGS -c $o.hip -o $B/$o.o) & OBJS="$OBJS $B/$o.o"; else OBJS="$OBJS $C/build/$o.o"; fi
done; wait
/opt/rocm/bin/hipcc -shared -fPIC -fvisibility=hidden --offload-arch=gfx950 $OBJS -lhiprtc -Wl,--version-script=$C/exports.map -o $R/gpurun_in/$NAME/plonky2_gpu_amd/libplonky2_hip.so
rm -rf $B
echo built $NAME
