#!/bin/bash
# the tail convolution without packed fp32 ops (csrc/drunet_tail.hip): reproducibility tests, the hunts that found the problem, ms per launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_drunet_gpu.py tests/test_loops_gpu.py -x -q -m gpu -k "tail or lanes" 2>&1 | tail -5
for co in 1 2 3 4; do timeout 300 python scripts/r06/race_hunt8.py $co 2>&1 | grep wsplit | cut -c1-200; done
timeout 600 python scripts/r06/race_hunt7.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
timeout 300 python scripts/r06/time_tail.py product 2>&1 | grep -v amdgpu.ids
timeout 900 python scripts/r06/flake_cfg5.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
