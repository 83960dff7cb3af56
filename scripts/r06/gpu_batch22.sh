#!/bin/bash
# the tail convolution's weight paths (DINV_TAIL_W = 0 scalar loads, 1 vector loads, 2 LDS, 3 lane reads; 4 = scalar loads without
# packed fp32 ops): reproducibility beside a bf16-split launch, and ms per launch
cd $GRAFT_REPO_ROOT
cp deepinv_amd/libdeepinv_amd.so /tmp/lib_product.so
for v in 0 1 2 3 4; do
  cp scripts/r06/variants/lib_$v.so deepinv_amd/libdeepinv_amd.so
  echo "== variant $v"
  timeout 300 python scripts/r06/race_hunt8.py 2 2>&1 | grep -v amdgpu.ids | cut -c1-200
  timeout 300 python scripts/r06/race_hunt8.py 3 2>&1 | grep -v amdgpu.ids | grep wsplit | cut -c1-200
  timeout 300 python scripts/r06/time_tail.py variant_$v 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_product.so deepinv_amd/libdeepinv_amd.so
