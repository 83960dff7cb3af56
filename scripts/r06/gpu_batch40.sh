#!/bin/bash
# F(4x4) kernel object with and without SLP vectorisation, same box: headline step and the 4-slice step
cd $GRAFT_REPO_ROOT
cp deepinv_amd/libdeepinv_amd.so /tmp/lib_product.so
run() {
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-split-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  B=32', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  timeout 600 python bench.py --batch 4 --steps 4 --warmup 2 --loop-graph --no-cpu-baseline --no-other-configs --no-split-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  B=4 ', d['ms_per_step'])"
}
for rep in 1 2; do
  echo "no SLP (product)"; cp /tmp/lib_product.so deepinv_amd/libdeepinv_amd.so; run
  echo "wino4 with SLP"; cp scripts/r06/variants/lib_w4slp.so deepinv_amd/libdeepinv_amd.so; run
done
cp /tmp/lib_product.so deepinv_amd/libdeepinv_amd.so
