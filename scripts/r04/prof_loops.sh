#!/bin/bash
# kernel traces of the loops of BASELINE configs 5 (DiffPIR) and 3 (PnP-HQS): what is there besides the denoiser?
#   -> gpurun_out/r04_cfg5_kernel_stats*.csv, r04_cfg3_kernel_stats.csv (copied to profiles/ by hand)
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 200 python scripts/r04/prof_cfg5.py 2 100 2>&1 | tail -2
scripts/prof.sh r04_cfg5 scripts/r04/prof_cfg5.py 1 20 > /dev/null
f=$(find $R/prof_r04_cfg5 -name "*kernel_stats.csv" | head -1); cp $f $R/r04_cfg5_kernel_stats.csv
timeout 200 python scripts/r04/prof_cfg3.py 1 30 2>&1 | tail -1
scripts/prof.sh r04_cfg3 scripts/r04/prof_cfg3.py 1 6 > /dev/null
f=$(find $R/prof_r04_cfg3 -name "*kernel_stats.csv" | head -1); cp $f $R/r04_cfg3_kernel_stats.csv; head -14 $f | cut -c1-170
