#!/bin/bash
# sixth hardware run: kernel timings after the coalesced staging loads, rocprofv3 kernel statistics of the bench command,
# HBM request counters of the fp32 DRUNet call
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$PWD/deepinv_amd:$LD_LIBRARY_PATH
R=gpurun_out
mkdir -p $R
timeout 100 scripts/r04/wino4_bench 32 20 15 quick > $R/r04_wino4_b32_v5.jsonl 2>&1; echo "b32 rc=$?"; cut -c1-150 $R/r04_wino4_b32_v5.jsonl
scripts/prof.sh r04_bench bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs
find $R/prof_r04_bench -name "*kernel_stats.csv" | head -1 | xargs head -16 | cut -c1-200
scripts/pmc.sh r04_rd "TCC_EA0_RDREQ_sum" scripts/bench_ops.py drunet_fp32
scripts/pmc.sh r04_wr "TCC_EA0_WRREQ_sum" scripts/bench_ops.py drunet_fp32
python scripts/pmc_summary.py $R/pmc_r04_rd conv3x3 | cut -c1-250
python scripts/pmc_summary.py $R/pmc_r04_wr conv3x3 | cut -c1-250
