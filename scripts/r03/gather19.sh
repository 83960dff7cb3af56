#!/bin/bash
cd $GRAFT_REPO_ROOT
P=scripts/pmc.sh
for lvl in 2 0; do
$P ws_sq1_l$lvl "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" scripts/r03/prof_wsplit.py $lvl 32
$P ws_sq2_l$lvl "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" scripts/r03/prof_wsplit.py $lvl 32
done
cd $GRAFT_REPO_ROOT
for d in gpurun_out/pmc_ws_*; do echo $d; python scripts/pmc_summary.py $d wsplit; done
