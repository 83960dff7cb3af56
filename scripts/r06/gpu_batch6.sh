#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
for b in 32 4; do
    timeout 400 python scripts/r06/loop_lanes_exp.py $b 0 2>&1 | grep -v amdgpu.ids | tee -a $R/r06_loop_lanes_threads_exp.jsonl | cut -c1-300
done
