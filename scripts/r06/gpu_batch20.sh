#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python scripts/r06/race_hunt.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_race_hunt.jsonl
