#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out
timeout 900 python -m pytest tests/test_named_shapes_gpu.py -q -m gpu -s -k "second or distinct or full_length" > $R/r06_b12_cfg3.log 2>&1; echo "rc=$?"; grep -E "cfg3 full length|cfg3 pair|second draw|passed|failed" $R/r06_b12_cfg3.log | cut -c1-1500
