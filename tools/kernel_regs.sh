#!/bin/bash
# kernel_regs.sh FILE.hip [PATTERN] [extra flags]: registers / spills / occupancy of the gfx950 kernels of one source, as the
# compiler reports them (-Rpass-analysis=kernel-resource-usage)
F=$1; PAT=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" -c $F -o /dev/null 2>&1 | grep "remark:" | sed -e 's/ \[-Rpass.*//' | awk '
/Function Name:/ {name=$NF} /TotalSGPRs:/ {s=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {sc=$NF} /SGPRs Spill:/ {ss=$NF} /VGPRs Spill:/ {vs=$NF} /Occupancy/ {o=$NF}
/LDS Size/ {print "vgpr="v, "sgpr="s, "vspill="vs, "sspill="ss, "scratch="sc, "occ="o, name}' | c++filt | grep -E "$PAT" | sed -e 's/plonky2_hip::nttk::(anonymous namespace):://' -e 's/(plonky2_hip.*//' | cut -c1-200
