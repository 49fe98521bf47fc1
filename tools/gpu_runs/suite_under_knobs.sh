#!/bin/bash
# the whole GPU suite once more under the alternative kernel selections (DESIGN / INTEGRATION section 10)
cd "$GRAFT_REPO_ROOT" || exit 1
# the A/B knobs exist in the diagnostic build only (csrc/knobs.h)
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/suite_knobs; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/$tag.log 2>&1; echo "$tag rc=$? $(tail -n 1 $O/$tag.log)"; }
run tile PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_KERNEL=tile
run wave_tiles PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_DIRECT=0
run narrow_nopipe PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_WIDE=0 PLONKY2_COMMIT_PIPELINE=0
run separate_leaves PLONKY2_HIP_LIBRARY=$DBG PLONKY2_FUSED_LEAVES=0
run plain_wg1 PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_XCD=0 PLONKY2_NTT_WG_PER_CU=1
