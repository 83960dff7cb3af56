#!/bin/bash
# rocprofv3 kernel statistics of the bench command at the final code
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
scripts/prof.sh r04_bench3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null
f=$(find $R/prof_r04_bench3 -name "*kernel_stats.csv" | head -1); cp $f $R/r04_bench_kernel_stats.csv; head -8 $f | cut -c1-140
rm -f $R/prof_r04_bench3/*kernel_trace.csv
