#!/bin/bash
# fourteenth hardware run: DiffPIR with the host-side schedule and one launch per affine update; DRUNet without the concatenated
# input copy; kernel trace of config 5's loop after the change
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 900 python -m pytest tests/test_elementwise_gpu.py tests/test_golden_gpu.py tests/test_loops_gpu.py tests/test_named_shapes_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 600 python -m pytest tests/test_drunet_gpu.py -q -m gpu -x -k "not winograd4 and not split" 2>&1 | tail -3
timeout 200 python scripts/r04/prof_cfg5.py 2 100 2>&1 | tail -2
scripts/prof.sh r04_cfg5b scripts/r04/prof_cfg5.py 1 20 > /dev/null
f=$(find $R/prof_r04_cfg5b -name "*kernel_stats.csv" | head -1); cp $f $R/r04_cfg5_kernel_stats_after.csv
