#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/r06/time_tail.py vgpr_weights 2>&1 | grep -v amdgpu.ids
