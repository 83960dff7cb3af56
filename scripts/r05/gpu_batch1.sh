#!/bin/bash
# round-5 batch 1 on the GPU box: the GPU tests touched so far, the Radon operator rows (fan-beam adjoint rewritten), the bench line
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 900 python -m pytest tests/test_mri_gpu.py tests/test_tomography_gpu.py tests/test_elementwise_gpu.py tests/test_named_shapes_gpu.py \
   "tests/test_drunet_gpu.py" tests/test_loops_gpu.py -q -m gpu -x -k "not cfg3 and not cfg4" > $R/r05_batch1_tests.log 2>&1; echo "tests rc=$?"; tail -5 $R/r05_batch1_tests.log
timeout 300 python scripts/r05/bench_fan.py > $R/r05_fan.jsonl 2>&1; cat $R/r05_fan.jsonl | tail -6
