#!/bin/bash
# validation batch of the round (ran on the final code): the whole GPU suite, smoke, the bench line (loops of configs 3 and 5 in
# both precision settings)
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 1200 python -m pytest tests -q -m gpu > $R/r04_gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $R/r04_gpu_tests.log | tail -2; grep -E "^FAILED|^ERROR" $R/r04_gpu_tests.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 5 --warmup 2 > $R/r04_bench_b32.json 2> $R/r04_bench_b32.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_bench_b32.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','value_bf16split','ms_per_step_bf16split','parity_rel_err_50it')}, d['roofline']['avg_launch_ms'], d['roofline']['frac_executed'])
for v in d['operators']:
    if 'loop' in v['op'] or 'Tomography' in v['op']: print(v['op'], v['ms'], v.get('conv_precision'))
P
