#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
for b in 4 8 16 32; do python scripts/r03/bench_split2d.py $b; done > $R/r03_tiles.jsonl 2>&1
grep resblock $R/r03_tiles.jsonl
timeout 600 python scripts/bench_train.py > $R/r03_train_step.jsonl 2>&1
grep "^{" $R/r03_train_step.jsonl | cut -c1-200
