#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out
timeout 900 python -m pytest tests/test_loops_gpu.py -q -m gpu -x -s -k "lane" > $R/r06_b10_tests.log 2>&1; echo "tests rc=$?"; grep -E "lane calibration|passed|failed" $R/r06_b10_tests.log | cut -c1-300
