#!/bin/bash
# validation batch of round 6: the whole GPU suite, smoke, the bench line, the per-GPU batches of the 2- / 4- / 8-GPU runs as bench.py
# runs them for --gpus > 1 (graph replay), the multi-GPU code path on one rank, rocprofv3 kernel stats of a short bench
#   validate.sh [tag] [prof]
cd $GRAFT_REPO_ROOT
R=gpurun_out
T=${1:-r06}
mkdir -p $R
timeout 1500 python -m pytest tests -q -m gpu -s > $R/${T}_gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $R/${T}_gpu_tests.log | tail -2; grep -E "^FAILED|^ERROR" $R/${T}_gpu_tests.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 5 --warmup 2 > $R/${T}_bench.json 2> $R/${T}_bench.err; echo "bench rc=$?"; tail -3 $R/${T}_bench.err
python - <<P
import json
d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','value_bf16split','ms_per_step_bf16split','parity_rel_err_50it')})
r=d['roofline']; print({k:r.get(k) for k in ('achieved','frac','direct_equiv','traffic','mfma_busy','avg_launch_ms','share_of_step','pmc_stale')}, d['config'])
for v in d['operators']:
    print(v['op'], v['ms'], {k:v[k] for k in v if k.startswith('parity_rel') or k.startswith('frac') or k in ('pmc_over_alg','A_adjoint_A_calls')})
P
for b in 16 8 4; do
  timeout 600 python bench.py --batch $b --steps 5 --warmup 2 --loop-graph --no-cpu-baseline --no-other-configs > $R/${T}_bench_batch$b.json 2>> $R/${T}_bench.err
  python -c "
import json; d=json.loads(open('$R/${T}_bench_batch$b.json').read().strip().splitlines()[-1]); print($b, d['ms_per_step'], d.get('ms_per_step_bf16split'), d['config']['batch_lanes'], d['config']['batch_lanes_calibration'])"
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --as-multi --batch 4 --steps 5 --warmup 2 > $R/${T}_bench_as_multi_batch4.json 2>> $R/${T}_bench.err
python -c "
import json; d=json.loads([l for l in open('$R/${T}_bench_as_multi_batch4.json') if l.startswith('{')][-1]); print('as-multi 4', d['ms_per_step'], d.get('ms_per_step_bf16split'), d['config'])"
if [ -n "$2" ]; then
  export TMPDIR=/tmp
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/${T}_bench_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/$R/${T}_bench_prof_line.json 2>/dev/null); echo "prof rc=$?"
  DB=$(find $R/${T}_bench_prof -name "*.db" | head -1)
  python3 scripts/r05/kstats.py $DB > $R/${T}_bench_kernel_stats.txt; head -30 $R/${T}_bench_kernel_stats.txt | cut -c1-200
  python3 scripts/r06/wino4_calls.py $DB >> $R/${T}_bench_kernel_stats.txt; tail -4 $R/${T}_bench_kernel_stats.txt
  rm -rf $R/${T}_bench_prof
fi
