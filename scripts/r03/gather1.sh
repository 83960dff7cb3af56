#!/bin/bash
# round-3 first GPU call: counter passes on the round-2 kernels (MRI / Radon HBM requests, conv SQ counters per level)
cd $GRAFT_REPO_ROOT
P=scripts/pmc.sh
$P mri_rd "TCC_EA0_RDREQ_sum" scripts/bench_ops.py mri2d mri3d
$P mri_wr "TCC_EA0_WRREQ_sum" scripts/bench_ops.py mri2d mri3d
$P radon_rd "TCC_EA0_RDREQ_sum" scripts/bench_ops.py radon
$P radon_wr "TCC_EA0_WRREQ_sum" scripts/bench_ops.py radon
for L in 0 1 2 3; do
  $P conv_s1_l$L "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" scripts/prof_bf16s.py $L 32 3
done
$P conv_s2_l1 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" scripts/prof_bf16s.py 1 32 3
python scripts/bench_bf16s.py 32 > gpurun_out/r03_bf16s_b32.jsonl 2>&1
python scripts/bench_bf16s.py 4 > gpurun_out/r03_bf16s_b4.jsonl 2>&1
