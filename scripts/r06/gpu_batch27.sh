#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mri_gpu.py -q -m gpu 2>&1 | tail -3
timeout 300 python scripts/r06/bench_mri_lanes.py 2>&1 | grep "lanes\": 1"
