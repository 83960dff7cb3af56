#!/bin/bash
# per-test durations of the GPU suite
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --durations=40 > gpurun_out/r06_gpu_durations.log 2>&1
grep -A45 "slowest" gpurun_out/r06_gpu_durations.log | cut -c1-160; tail -1 gpurun_out/r06_gpu_durations.log
