#!/bin/bash
# validation batch of round 5: the whole GPU suite, smoke, the bench line; optionally rocprofv3 kernel stats of a short bench
#   validate.sh [tag] [prof]
cd $GRAFT_REPO_ROOT
R=gpurun_out
T=${1:-r05}
mkdir -p $R
timeout 1500 python -m pytest tests -q -m gpu -s > $R/${T}_gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $R/${T}_gpu_tests.log | tail -2; grep -E "^FAILED|^ERROR" $R/${T}_gpu_tests.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 5 --warmup 2 > $R/${T}_bench.json 2> $R/${T}_bench.err; echo "bench rc=$?"; tail -3 $R/${T}_bench.err
python - <<P
import json
d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','value_bf16split','ms_per_step_bf16split','parity_rel_err_50it')})
r=d['roofline']; print({k:r.get(k) for k in ('achieved','frac','direct_equiv','traffic','mfma_busy','avg_launch_ms','pmc_stale')})
for v in d['operators']:
    print(v['op'], v['ms'], {k:v[k] for k in v if k.startswith('parity_rel') or k.startswith('frac') or k=='pmc_over_alg'})
P
if [ -n "$2" ]; then
  export TMPDIR=/tmp
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/${T}_bench_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1); echo "prof rc=$?"
  python3 scripts/r05/kstats.py $(find $R/${T}_bench_prof -name "*.db" | head -1) > $R/${T}_bench_kernel_stats.txt; head -30 $R/${T}_bench_kernel_stats.txt | cut -c1-200
fi
