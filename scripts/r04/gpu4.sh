#!/bin/bash
# fourth hardware run: residual loads without spills, tail split along the input channels (two launches), B = 32 and B = 4
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$PWD/deepinv_amd:$LD_LIBRARY_PATH
R=gpurun_out
mkdir -p $R
timeout 200 scripts/r04/wino4_bench 32 20 > $R/r04_wino4_b32_v4.jsonl 2> $R/r04_wino4_b32_v4.err; echo "b32 rc=$?"
grep -v winograd2 $R/r04_wino4_b32_v4.jsonl | cut -c1-200
timeout 100 scripts/r04/wino4_bench 32 20 15 quick nosplit > $R/r04_wino4_b32_v4_nosplit.jsonl 2>&1; echo "nosplit rc=$?"
cut -c1-160 $R/r04_wino4_b32_v4_nosplit.jsonl
timeout 100 scripts/r04/wino4_bench 4 20 > $R/r04_wino4_b4_v4.jsonl 2> $R/r04_wino4_b4_v4.err; echo "b4 rc=$?"
grep -v fp64 $R/r04_wino4_b4_v4.jsonl | cut -c1-160
timeout 200 scripts/r04/wino4_time_d0 32 5 15 quick > $R/r04_wino4_time_v4.jsonl 2> $R/r04_wino4_time_v4.err; echo "time rc=$?"
grep ticks $R/r04_wino4_time_v4.jsonl
