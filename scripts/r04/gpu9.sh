#!/bin/bash
# ninth hardware run: the whole GPU suite, smoke, the bench line at 32 / 16 / 8 / 4 slices
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 900 python -m pytest tests -q -m gpu > $R/r04_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -5 $R/r04_gpu_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for b in 4 8 16; do
timeout 200 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $R/r04_bench_b$b.json 2> $R/r04_bench_b$b.err; echo "bench b$b rc=$?"
python - $b <<'P'
import json,sys
d=json.loads(open('gpurun_out/r04_bench_b%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_bf16split','ms_per_step_bf16split')}, d['roofline']['avg_launch_ms'], d['roofline']['kernels'])
P
done
timeout 400 python bench.py --steps 5 --warmup 2 > $R/r04_bench_b32.json 2> $R/r04_bench_b32.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_bench_b32.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_bf16split','ms_per_step_bf16split','parity_rel_err_50it','parity_unit_gain_50it')}, d['roofline']['avg_launch_ms'], d['roofline']['frac_executed'], d['roofline']['traffic'])
P
