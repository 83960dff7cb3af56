#!/bin/bash
# twenty-fourth hardware run: the three-part-split (six-product) stride-2 down convolution of the fp32 setting: tests, DRUNet forward time
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_drunet_gpu.py -q -m gpu -k "bf16x3 or fp32_precision or default_precision" 2>&1 | tail -2
timeout 200 python scripts/bench_ops.py drunet_fp32 drunet 2>&1 | tail -2 | cut -c1-200
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-split-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['layer_rel_err_vs_fp64'])"
