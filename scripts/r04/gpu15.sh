#!/bin/bash
# fifteenth hardware run: kernel trace of BASELINE config 3's PnP-HQS loop (how much of it is the CG prox?)
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 200 python scripts/r04/prof_cfg3.py 1 30 2>&1 | tail -1
scripts/prof.sh r04_cfg3 scripts/r04/prof_cfg3.py 1 6 > /dev/null
f=$(find $R/prof_r04_cfg3 -name "*kernel_stats.csv" | head -1); cp $f $R/r04_cfg3_kernel_stats.csv; head -14 $f | cut -c1-170
