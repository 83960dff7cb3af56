#!/bin/bash
# SQ counters of the MRI kernels (scripts/r05/mri_bench on one library, 3 reps per op): two --pmc passes (8 SQ slots each)
#   mri_pmc.sh <tag> [lib]
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
TAG=$1
LIB=$GRAFT_REPO_ROOT/${2:-deepinv_amd/libdeepinv_amd.so}
export TMPDIR=/tmp
cd /tmp
for pass in "a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" \
            "c TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; p=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $R/r05_mri_pmc_${TAG}_$p -o pmc --output-format csv -- $GRAFT_REPO_ROOT/scripts/r05/mri_bench $LIB --reps 2 > /dev/null 2>&1; echo "pass $p rc=$?"
done
