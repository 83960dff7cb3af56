#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/r03/bench_split2d.py 32 > gpurun_out/r03_split2d_b32_v4.jsonl 2>&1
python scripts/r03/bench_split2d.py 4 > gpurun_out/r03_split2d_b4_v4.jsonl 2>&1
