#!/bin/bash
# second hardware run of the F(4x4,3x3) kernel: cout-tile-outermost order at levels 2-3, per-phase cycle stamps, SQ counters at levels 0 and 2
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$PWD/deepinv_amd:$LD_LIBRARY_PATH
R=gpurun_out
mkdir -p $R
timeout 200 scripts/r04/wino4_bench 32 20 > $R/r04_wino4_b32_v2.jsonl 2> $R/r04_wino4_b32_v2.err; echo "b32 rc=$?"
grep -v winograd2 $R/r04_wino4_b32_v2.jsonl | cut -c1-200
timeout 200 scripts/r04/wino4_time 32 5 > $R/r04_wino4_time.jsonl 2> $R/r04_wino4_time.err; echo "time rc=$?"
grep ticks $R/r04_wino4_time.jsonl
P=scripts/r04/pmc_bin.sh
$P w4_sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" scripts/r04/wino4_bench 32 2 5
$P w4_sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" scripts/r04/wino4_bench 32 2 5
python scripts/pmc_summary.py gpurun_out/pmc_w4_sq1 wino
python scripts/pmc_summary.py gpurun_out/pmc_w4_sq2 wino
