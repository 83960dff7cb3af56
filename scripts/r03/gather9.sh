#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/prof_bench -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg > $R/r03_bench_prof.json 2> $R/r03_bench_prof.err
rocprofv3 --kernel-trace --stats -d $R/prof_cfg4 -o cfg4 --output-format csv -- python $GRAFT_REPO_ROOT/scripts/r03/prof_cfg4.py > $R/r03_cfg4.json 2>&1
N=2 python $GRAFT_REPO_ROOT/scripts/r03/prof_cfg4.py bf16split > $R/r03_cfg4_bf16split.json 2>&1
cd $GRAFT_REPO_ROOT
P=scripts/pmc.sh
$P radon3 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" scripts/bench_ops.py radon
$P drd3 "TCC_EA0_RDREQ_sum" scripts/bench_ops.py drunet
$P dwr3 "TCC_EA0_WRREQ_sum" scripts/bench_ops.py drunet
$P mrd3 "TCC_EA0_RDREQ_sum" scripts/bench_ops.py mri2d mri3d
$P mwr3 "TCC_EA0_WRREQ_sum" scripts/bench_ops.py mri2d mri3d
$P s2d_final_s1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" scripts/r03/prof_split2d.py 1 32 3
$P s2d_final_s2 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" scripts/r03/prof_split2d.py 1 32 3
