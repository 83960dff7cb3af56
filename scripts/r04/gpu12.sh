#!/bin/bash
# twelfth hardware run: DRUNet GPU tests (permuted F(4x4) point slots, lane-shift tail kernel), bench line without the CPU leg
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 600 python -m pytest tests/test_drunet_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $R/r04_bench_v2.json 2> $R/r04_bench_v2.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_bench_v2.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','value_bf16split','ms_per_step_bf16split','parity_rel_err_50it','parity_unit_gain_50it')}, d['roofline']['avg_launch_ms'], d['roofline']['frac_executed'])
P
