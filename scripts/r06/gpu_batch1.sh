#!/bin/bash
# round-6 batch 1 on the GPU box: the new GPU tests (fused BlurFFT, tiled conv2d, generator distributions, bench pre-flight), then the
# whole suite, the bench line with the new operator rows, the 4-slice leg with graph replay, and a kernel trace of the 4-slice loop
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 900 python -m pytest tests/test_blur_gpu.py tests/test_generators_gpu.py tests/test_elementwise_gpu.py tests/test_golden_gpu.py -q -m gpu -s > $R/r06_b1_new_tests.log 2>&1; echo "new tests rc=$?"; tail -4 $R/r06_b1_new_tests.log
timeout 900 python -m pytest tests/test_loops_gpu.py -q -m gpu -x -k "preflight or rccl or graph" -s > $R/r06_b1_preflight.log 2>&1; echo "preflight rc=$?"; tail -4 $R/r06_b1_preflight.log
timeout 1500 python -m pytest tests -q -m gpu > $R/r06_b1_gpu_tests.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $R/r06_b1_gpu_tests.log | tail -2; grep -E "^FAILED|^ERROR" $R/r06_b1_gpu_tests.log | head
timeout 900 python bench.py --steps 3 --warmup 1 > $R/r06_b1_bench.json 2> $R/r06_b1_bench.err; echo "bench rc=$?"; tail -3 $R/r06_b1_bench.err
python - <<P
import json
d=json.loads(open('gpurun_out/r06_b1_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','value_bf16split','ms_per_step_bf16split','parity_rel_err_50it')})
r=d['roofline']; print({k:r.get(k) for k in ('achieved','frac','avg_launch_ms','pmc_stale')})
for v in d['operators']:
    print(v['op'], v['ms'], {k:v[k] for k in v if k.startswith('parity_rel') or k.startswith('frac') or k in ('GBps','A_adjoint_A_calls')})
P
for b in 4 8 16; do
  timeout 600 python bench.py --batch $b --steps 5 --warmup 2 --loop-graph --no-cpu-baseline --no-other-configs > $R/r06_b1_bench_batch$b.json 2>> $R/r06_b1_bench.err
  python -c "
import json; d=json.loads(open('$R/r06_b1_bench_batch$b.json').read().strip().splitlines()[-1]); print($b, d['ms_per_step'], d.get('ms_per_step_bf16split'), d['roofline']['avg_launch_ms'], d['roofline']['kernels'])"
done
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$R/r06_b1_prof4 -o b4 -- python $GRAFT_REPO_ROOT/bench.py --batch 4 --steps 1 --warmup 1 --iters 12 --loop-graph --no-split-leg --no-cpu-baseline --no-other-configs > /dev/null 2>&1); echo "prof rc=$?"
DB=$(find $R/r06_b1_prof4 -name "*.db" | head -1)
python3 scripts/r05/kstats.py $DB > $R/r06_b1_b4_kernel_stats.txt; head -40 $R/r06_b1_b4_kernel_stats.txt | cut -c1-180
python3 scripts/r06/gaps.py $DB > $R/r06_b1_b4_gaps.txt; cat $R/r06_b1_b4_gaps.txt
rm -rf $R/r06_b1_prof4
