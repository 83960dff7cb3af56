#!/bin/bash
# per-GPU batches of the 2- / 4-GPU strong-scaling runs on one GPU, and the SQ counters of the Winograd conv at level 0
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
for b in 16 8; do
timeout 150 python bench.py --batch $b --no-cpu-baseline --no-other-configs --no-fp32-leg > $R/r03_bench_b$b.json 2> $R/r03_bench_b$b.err
cut -c1-230 $R/r03_bench_b$b.json
done
P=scripts/pmc.sh
$P ws_l0 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" scripts/r03/prof_wsplit.py 0 32 > /dev/null
python scripts/pmc_summary.py gpurun_out/pmc_ws_l0 wsplit
