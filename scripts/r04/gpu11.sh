#!/bin/bash
# eleventh hardware run: F(4x4) kernel with permuted point slots (same six accumulators retire first in every wave), residual
# requested after the first exchange round, first block of a tile accumulating onto the constant 0
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$PWD/deepinv_amd:$LD_LIBRARY_PATH
R=gpurun_out
mkdir -p $R
timeout 200 scripts/r04/wino4_bench 32 20 15 > $R/r04_wino4_b32_v6.jsonl 2>&1; echo "b32 rc=$?"; grep -v winograd2 $R/r04_wino4_b32_v6.jsonl | cut -c1-200
timeout 100 scripts/r04/wino4_time_d0 32 5 15 quick > $R/r04_wino4_time_v6.jsonl 2>&1; echo "time rc=$?"; grep ticks $R/r04_wino4_time_v6.jsonl
timeout 100 scripts/r04/wino4_bench 4 50 15 quick > $R/r04_wino4_b4_v6.jsonl 2>&1; echo "b4 rc=$?"; grep winograd4 $R/r04_wino4_b4_v6.jsonl | cut -c1-200
