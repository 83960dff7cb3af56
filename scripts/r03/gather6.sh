#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r03_gpu_tests_a.log
python bench.py --steps 3 --warmup 1 > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err
