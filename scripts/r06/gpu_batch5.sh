#!/bin/bash
# round-6 batch 5: the GPU suite, the default bench line, the per-GPU batches of the 2- / 4- / 8-GPU runs with graph replay (as bench.py
# runs them for --gpus > 1), the counter passes of the new operator rows
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
T=${1:-r06_b5}
timeout 1500 python -m pytest tests -q -m gpu -s > $R/${T}_gpu_tests.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $R/${T}_gpu_tests.log | tail -2; grep -E "^FAILED|^ERROR" $R/${T}_gpu_tests.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 5 --warmup 2 > $R/${T}_bench.json 2> $R/${T}_bench.err; echo "bench rc=$?"; tail -3 $R/${T}_bench.err
python - <<P
import json
d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','value_bf16split','ms_per_step_bf16split','parity_rel_err_50it')})
r=d['roofline']; print({k:r.get(k) for k in ('achieved','frac','avg_launch_ms','share_of_step','pmc_stale','traffic','mfma_busy')}, d['config'])
for v in d['operators']:
    print(v['op'], v['ms'], {k:v[k] for k in v if k.startswith('parity_rel') or k.startswith('frac') or k in ('GBps','A_adjoint_A_calls','pmc_over_alg')})
print(d.get('parity_unit_gain_50it'))
P
for b in 16 8 4; do
  timeout 600 python bench.py --batch $b --steps 5 --warmup 2 --loop-graph --no-cpu-baseline --no-other-configs > $R/${T}_bench_batch$b.json 2>> $R/${T}_bench.err
  python -c "
import json; d=json.loads(open('$R/${T}_bench_batch$b.json').read().strip().splitlines()[-1]); print($b, d['ms_per_step'], d.get('ms_per_step_bf16split'), d['roofline']['avg_launch_ms'], d['roofline']['kernels'], d['roofline']['package_during_timed_steps'])"
done
timeout 600 python bench.py --batch 32 --steps 5 --warmup 2 --loop-graph --no-cpu-baseline --no-other-configs > $R/${T}_bench_batch32_graph.json 2>> $R/${T}_bench.err
python -c "
import json; d=json.loads(open('$R/${T}_bench_batch32_graph.json').read().strip().splitlines()[-1]); print(32, 'graph', d['ms_per_step'], d.get('ms_per_step_bf16split'))"
bash scripts/r06/pmc_ops.sh ${2:-unknown} 2>&1 | tail -12
