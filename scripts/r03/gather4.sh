#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/r03/bench_split2d.py 32 > gpurun_out/r03_split2d_b32_v3.jsonl 2>&1
python scripts/r03/bench_split2d.py 4 > gpurun_out/r03_split2d_b4_v3.jsonl 2>&1
P=scripts/pmc.sh
for L in 1; do
$P s2d3_s1_l$L "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" scripts/r03/prof_split2d.py $L 32 3
$P s2d3_s2_l$L "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" scripts/r03/prof_split2d.py $L 32 3
done
