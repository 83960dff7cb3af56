#!/bin/bash
# the F(4x4) kernels per DRUNet level through the C-ABI harness (scripts/r05/wino4_bench.cpp): fp32 MFMA form, bf16 x3 form, F(2x2)
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$PWD/deepinv_amd:$LD_LIBRARY_PATH
R=gpurun_out
mkdir -p $R
T=${1:-1}
timeout 300 scripts/r05/wino4_bench 32 20 15 > $R/r05_wino4_b32_$T.jsonl 2>&1; echo "b32 rc=$?"; grep -v '"winograd2"' $R/r05_wino4_b32_$T.jsonl | cut -c1-260
timeout 200 scripts/r05/wino4_bench 4 50 15 quick > $R/r05_wino4_b4_$T.jsonl 2>&1; echo "b4 rc=$?"; grep winograd4 $R/r05_wino4_b4_$T.jsonl | cut -c1-200
