#!/bin/bash
# third hardware run: new epilogue (all levels, correctness + timing + phase stamps), diagnostic builds of the main loop at level 2, fixed microbenchmark
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$PWD/deepinv_amd:$LD_LIBRARY_PATH
R=gpurun_out
mkdir -p $R
timeout 200 scripts/r04/wino4_bench 32 20 > $R/r04_wino4_b32_v3.jsonl 2> $R/r04_wino4_b32_v3.err; echo "b32 rc=$?"
grep -v winograd2 $R/r04_wino4_b32_v3.jsonl | cut -c1-200
timeout 200 scripts/r04/wino4_time_d0 32 5 15 quick > $R/r04_wino4_time_v3.jsonl 2> $R/r04_wino4_time_v3.err; echo "time rc=$?"
grep ticks $R/r04_wino4_time_v3.jsonl
for d in 1 2 4 8 16 15 31; do
  echo "diag $d"
  timeout 60 scripts/r04/wino4_time_d$d 32 10 4 quick 2>&1 | grep -E "ticks|\"ms\"" | sed -e "s/^{/{\"diag\": $d, /" | tee -a $R/r04_wino4_diag.jsonl | cut -c1-330
done
timeout 100 scripts/ubench/vmem_beside_mfma > $R/r04_ubench.jsonl 2> $R/r04_ubench.err; echo "ubench rc=$?"
cat $R/r04_ubench.jsonl | cut -c1-200; tail -3 $R/r04_ubench.err
