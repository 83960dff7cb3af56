#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/r03/bench_split2d.py 32 > gpurun_out/r03_split2d_b32.jsonl 2>&1
python scripts/r03/bench_split2d.py 4 > gpurun_out/r03_split2d_b4.jsonl 2>&1
python -m pytest tests/test_drunet_gpu.py -q -k "matches_oracle and not winograd and not bf16" -x 2>&1 | tail -5 > gpurun_out/r03_t1.log
python -m pytest tests/test_named_shapes_gpu.py tests/test_golden_gpu.py -q 2>&1 | tail -15 > gpurun_out/r03_t2.log
python scripts/bench_ops.py drunet drunet4 > gpurun_out/r03_drunet.jsonl 2>&1
