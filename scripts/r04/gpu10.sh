#!/bin/bash
# tenth hardware run: packed fp32 VALU beside the fp32 MFMA (scripts/ubench/mfma_coissue.hip), the 3-D pool test after its fix
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 120 scripts/ubench/mfma_coissue > $R/r04_ubench_coissue.txt 2>&1; echo "ubench rc=$?"; cat $R/r04_ubench_coissue.txt
timeout 300 python -m pytest tests/test_drunet_gpu.py -q -m gpu -k "drunet3d" 2>&1 | tail -3
