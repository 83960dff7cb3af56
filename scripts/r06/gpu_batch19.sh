#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python scripts/r06/flake_cfg5.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_flake_cfg5.txt
