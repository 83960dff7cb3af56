#!/bin/bash
# the F(4x4) kernel per DRUNet level through the C-ABI harness (scripts/r04/wino4_bench.cpp; build lines in its header): correctness
# against F(2x2) and sampled fp64, ms per launch at B = 32 and B = 4, per-phase cycle stamps of the -DDINV_W4_TIMING build
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$PWD/deepinv_amd:$LD_LIBRARY_PATH
R=gpurun_out
mkdir -p $R
timeout 200 scripts/r04/wino4_bench 32 20 15 > $R/r04_wino4_b32_v6.jsonl 2>&1; echo "b32 rc=$?"; grep -v winograd2 $R/r04_wino4_b32_v6.jsonl | cut -c1-200
timeout 100 scripts/r04/wino4_time_d0 32 5 15 quick > $R/r04_wino4_time_v6.jsonl 2>&1; echo "time rc=$?"; grep ticks $R/r04_wino4_time_v6.jsonl
timeout 100 scripts/r04/wino4_bench 4 50 15 quick > $R/r04_wino4_b4_v6.jsonl 2>&1; echo "b4 rc=$?"; grep winograd4 $R/r04_wino4_b4_v6.jsonl | cut -c1-200
