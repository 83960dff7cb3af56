#!/bin/bash
# nineteenth hardware run (final code of the round, after the Radon change): the whole GPU suite, smoke, the bench line at 32 / 16 / 8 / 4 slices, rocprofv3 kernel statistics of the
# bench command (final code of the round)
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 1200 python -m pytest tests -q -m gpu > $R/r04_gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $R/r04_gpu_tests.log | tail -2
timeout 300 python -m pytest tests/test_named_shapes_gpu.py -q -m gpu -s -k cfg4 2>&1 | grep -E "gradient errors|passed|failed"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 500 python bench.py --steps 5 --warmup 2 > $R/r04_bench_b32.json 2> $R/r04_bench_b32.err; echo "bench rc=$?"
for b in 4 8 16; do
timeout 200 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $R/r04_bench_b$b.json 2> $R/r04_bench_b$b.err; echo "bench b$b rc=$?"
done
python - <<'P'
import json
for b in (32,16,8,4):
    d=json.loads(open('gpurun_out/r04_bench_b%d.json'%b).read().strip().splitlines()[-1])
    print(b,{k:d.get(k) for k in ('value','ms_per_step','value_bf16split','ms_per_step_bf16split','parity_rel_err_50it','parity_unit_gain_50it')}, d['roofline']['avg_launch_ms'], d['roofline']['frac_executed'])
P
scripts/prof.sh r04_bench2 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null
f=$(find $R/prof_r04_bench2 -name "*kernel_stats.csv" | head -1); cp $f $R/r04_bench_kernel_stats.csv; head -12 $f | cut -c1-150
