#!/bin/bash
# cfg4 training step with the 3x3x3 weight gradient in one call and the activation free list sized by the device's memory
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu -k "3d or cfg4 or backward or unfolded or train" 2>&1 | tail -4
N=5 timeout 300 python scripts/r03/prof_cfg4.py 2>&1 | grep cfg4
N=5 timeout 300 python scripts/r03/prof_cfg4.py 2>&1 | grep cfg4
