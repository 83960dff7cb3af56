#!/bin/bash
# round-6 batch 4: lanes sweep without the tail split (what --lanes N now means for N > 1), then the GPU suite and the default bench
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
run() { # name, args...
  n=$1; shift
  timeout 600 python bench.py "$@" --steps 4 --warmup 2 --no-split-leg --no-cpu-baseline --no-other-configs > $R/r06_b4_$n.json 2>> $R/r06_b4_bench.err
  python -c "
import json; d=json.loads(open('$R/r06_b4_$n.json').read().strip().splitlines()[-1]); print('$n', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config'].get('batch_lanes'), d['config'].get('loop_graph'), d['roofline']['package_during_timed_steps']['socket_power_w_median'], d['roofline']['package_during_timed_steps']['sclk_mhz_median'])"
}
run b32_l2_eager --batch 32 --lanes 2
run b32_l3_eager --batch 32 --lanes 3
run b32_l3 --batch 32 --lanes 3 --loop-graph
run b32_l4 --batch 32 --lanes 4 --loop-graph
run b16_l3 --batch 16 --lanes 3 --loop-graph
run b16_l4 --batch 16 --lanes 4 --loop-graph
run b8_l3 --batch 8 --lanes 3 --loop-graph
run b8_l4 --batch 8 --lanes 4 --loop-graph
run b4_l2 --batch 4 --lanes 2 --loop-graph
run b4_l2_eager --batch 4 --lanes 2
run b2_l2 --batch 2 --lanes 2 --loop-graph
run b2_l1 --batch 2 --lanes 1 --loop-graph
timeout 1500 python -m pytest tests -q -m gpu > $R/r06_b4_gpu_tests.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $R/r06_b4_gpu_tests.log | tail -2; grep -E "^FAILED|^ERROR" $R/r06_b4_gpu_tests.log | head
timeout 900 python bench.py --steps 3 --warmup 1 > $R/r06_b4_bench.json 2>> $R/r06_b4_bench.err; echo "bench rc=$?"
python - <<P
import json
d=json.loads(open('gpurun_out/r06_b4_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','value_bf16split','ms_per_step_bf16split','parity_rel_err_50it')})
r=d['roofline']; print({k:r.get(k) for k in ('achieved','frac','avg_launch_ms','share_of_step','pmc_stale')}, d['config'])
for v in d['operators']:
    print(v['op'], v['ms'], {k:v[k] for k in v if k.startswith('parity_rel') or k.startswith('frac') or k in ('GBps','A_adjoint_A_calls')})
print(d.get('parity_unit_gain_50it'))
P
tail -3 $R/r06_b4_bench.err
