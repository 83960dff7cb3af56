#!/bin/bash
cd $GRAFT_REPO_ROOT
cp deepinv_amd/libdeepinv_amd.so /tmp/lib_product.so
for v in 0 1 2; do
  cp scripts/r06/variants/lib_$v.so deepinv_amd/libdeepinv_amd.so
  echo "== variant $v"
  timeout 600 python scripts/r06/race_hunt11.py 2 0 2>&1 | grep -v amdgpu.ids | cut -c1-3000
  timeout 600 python scripts/r06/race_hunt11.py 2 1 2>&1 | grep -v amdgpu.ids | cut -c1-3000
done
cp /tmp/lib_product.so deepinv_amd/libdeepinv_amd.so
