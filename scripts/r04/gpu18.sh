#!/bin/bash
# eighteenth hardware run: Radon kernels with the closed-form lattice positions and the interval march (tests, operator timings, the
# config 3 loop); gradient tolerances at 1e-4
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 900 python -m pytest tests/test_tomography_gpu.py tests/test_named_shapes_gpu.py tests/test_golden_gpu.py tests/test_loops_gpu.py -q -m gpu -s 2>&1 | grep -E "gradient error|passed|failed|Error|error" | head -20
timeout 300 python scripts/bench_ops.py radon 2>&1 | tail -8
timeout 200 python scripts/r04/prof_cfg3.py 1 30 2>&1 | tail -1
