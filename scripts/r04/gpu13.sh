#!/bin/bash
# thirteenth hardware run: kernel trace of BASELINE config 5's DiffPIR loop (what is there besides the denoiser?)
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 200 python scripts/r04/prof_cfg5.py 2 100 2>&1 | tail -2
scripts/prof.sh r04_cfg5 scripts/r04/prof_cfg5.py 1 20
tail -2 $R/prof_r04_cfg5/log.txt
f=$(find $R/prof_r04_cfg5 -name "*kernel_stats.csv" | head -1); cp $f $R/r04_cfg5_kernel_stats_before.csv; head -40 $f | cut -c1-160
