#!/bin/bash
# first hardware run of the F(4x4,3x3) fp32 kernel: correctness + timing per DRUNet level (B = 32 and 4), then the load-issue microbenchmark
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$PWD/deepinv_amd:$LD_LIBRARY_PATH
R=gpurun_out
mkdir -p $R
timeout 200 scripts/r04/wino4_bench 32 20 > $R/r04_wino4_b32.jsonl 2> $R/r04_wino4_b32.err; echo "b32 rc=$?"
cat $R/r04_wino4_b32.jsonl; tail -3 $R/r04_wino4_b32.err
timeout 100 scripts/r04/wino4_bench 4 20 > $R/r04_wino4_b4.jsonl 2> $R/r04_wino4_b4.err; echo "b4 rc=$?"
grep -v fp64 $R/r04_wino4_b4.jsonl | cut -c1-160
timeout 60 scripts/ubench/vmem_beside_mfma > $R/r04_ubench.jsonl 2> $R/r04_ubench.err; echo "ubench rc=$?"
head -50 $R/r04_ubench.jsonl; tail -3 $R/r04_ubench.err
