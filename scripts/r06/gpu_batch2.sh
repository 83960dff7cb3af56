#!/bin/bash
# round-6 batch 2: the tests that failed / were added since batch 1, batch lanes (two half-batches on two streams) at 4 / 8 / 16 / 32
# slices with and without graph replay, per-level sustained power of the F(4x4) kernel
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 900 python -m pytest tests/test_generators_gpu.py tests/test_loops_gpu.py -q -m gpu -x -k "poly or lanes or preflight or graph" > $R/r06_b2_tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/r06_b2_tests.log
timeout 600 python -m pytest tests/test_named_shapes_gpu.py -q -m gpu -s -k "pair or second" > $R/r06_b2_cfg3.log 2>&1; echo "cfg3 rc=$?"; grep -E "cfg3 pair|second draw|passed|failed" $R/r06_b2_cfg3.log | cut -c1-900
for b in 4 8 16 32; do
  for l in 1 2; do
    timeout 600 python bench.py --batch $b --lanes $l --steps 4 --warmup 2 --loop-graph --no-split-leg --no-cpu-baseline --no-other-configs > $R/r06_b2_bench_b${b}_l${l}.json 2>> $R/r06_b2_bench.err
    python -c "
import json; d=json.loads(open('$R/r06_b2_bench_b${b}_l${l}.json').read().strip().splitlines()[-1]); print('batch',$b,'lanes',$l,'graph', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['package_during_timed_steps'])"
  done
done
timeout 600 python bench.py --batch 32 --lanes 2 --steps 4 --warmup 2 --no-split-leg --no-cpu-baseline --no-other-configs > $R/r06_b2_bench_b32_l2_eager.json 2>> $R/r06_b2_bench.err
python -c "
import json; d=json.loads(open('$R/r06_b2_bench_b32_l2_eager.json').read().strip().splitlines()[-1]); print('batch 32 lanes 2 eager', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['package_during_timed_steps'])"
timeout 600 python bench.py --batch 4 --lanes 4 --steps 4 --warmup 2 --loop-graph --no-split-leg --no-cpu-baseline --no-other-configs > $R/r06_b2_bench_b4_l4.json 2>> $R/r06_b2_bench.err
python -c "
import json; d=json.loads(open('$R/r06_b2_bench_b4_l4.json').read().strip().splitlines()[-1]); print('batch 4 lanes 4 graph', d['ms_per_step'])"
timeout 300 python scripts/r06/wino4_power_levels.py 32 3 > $R/r06_wino4_power_levels_b32.jsonl 2>&1; cat $R/r06_wino4_power_levels_b32.jsonl | cut -c1-400
timeout 300 python scripts/r06/wino4_power_levels.py 4 2 > $R/r06_wino4_power_levels_b4.jsonl 2>&1; cat $R/r06_wino4_power_levels_b4.jsonl | cut -c1-400
tail -5 $R/r06_b2_bench.err
