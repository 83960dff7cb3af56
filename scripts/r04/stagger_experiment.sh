#!/bin/bash
# experiment, timing build only: do the workgroups of a launch run their epilogues in lock step and saturate
# HBM in bursts?  Workgroup j of an XCD starts (j % 4) / 4 of a tile late (DINV_W4_STAGGER cycles per tile); per-phase stamps of
# tiles 1-2 and the launch time with and without the stagger
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$PWD/deepinv_amd:$LD_LIBRARY_PATH
R=gpurun_out
mkdir -p $R
: > $R/r04_wino4_stagger.jsonl
for st in 0 64000 0 64000; do
  DINV_W4_STAGGER=$st timeout 100 scripts/r04/wino4_time_d0 32 10 1 quick 2>&1 | grep -E "ticks|\"ms\"" | sed -e "s/^{/{\"stagger\": $st, /" | tee -a $R/r04_wino4_stagger.jsonl | cut -c1-420
done
for st in 0 110000; do
  DINV_W4_STAGGER=$st timeout 100 scripts/r04/wino4_time_d0 32 10 2 quick 2>&1 | grep -E "ticks|\"ms\"" | sed -e "s/^{/{\"stagger\": $st, /" | tee -a $R/r04_wino4_stagger.jsonl | cut -c1-420
done
