#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_mri_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -2
for L in 8 16; do echo "== vec4 DINV_ROWS_L=$L"; DINV_ROWS_L=$L python scripts/bench_ops.py mri2d mri3d 2>&1 | tail -4 | cut -c1-150; done
echo "== novec"; DINV_NO_VEC4=1 python scripts/bench_ops.py mri2d 2>&1 | tail -2 | cut -c1-150
