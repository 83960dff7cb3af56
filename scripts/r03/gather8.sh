#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_mri_gpu.py tests/test_golden_gpu.py tests/test_named_shapes_gpu.py tests/test_trainer_gpu.py tests/test_tomography_gpu.py tests/test_drunet_gpu.py tests/test_loops_gpu.py -q -x 2>&1 | tail -12 > gpurun_out/r03_gpu_tests_b.log
python scripts/bench_ops.py mri2d mri3d radon > gpurun_out/r03_ops_b.jsonl 2>&1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg > gpurun_out/r03_bench_b.json 2> gpurun_out/r03_bench_b.err
