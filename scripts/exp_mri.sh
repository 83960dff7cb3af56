#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in 4 8 16; do echo "== DINV_ROWS_L=$L"; DINV_ROWS_L=$L python scripts/bench_ops.py mri2d 2>&1 | tail -2 | cut -c1-140; done
echo "== coil loop"; DINV_MRI_COIL_LOOP=1 DINV_ROWS_L=8 python scripts/bench_ops.py mri2d 2>&1 | tail -2 | cut -c1-140
DINV_ROWS_L=8 python -m pytest tests/test_mri_gpu.py -x -q 2>&1 | tail -2
