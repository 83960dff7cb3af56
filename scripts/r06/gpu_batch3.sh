#!/bin/bash
# round-6 batch 3: blur kernels after the rewrite of the tiled inner loop / 512-thread BlurFFT column pass, lanes experiments, loops
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 900 python -m pytest tests/test_blur_gpu.py tests/test_golden_gpu.py tests/test_loops_gpu.py -q -m gpu -x > $R/r06_b3_tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/r06_b3_tests.log
timeout 600 python -m pytest tests/test_named_shapes_gpu.py -q -m gpu -s -k "distinct" > $R/r06_b3_cfg3.log 2>&1; echo "cfg3 rc=$?"; grep -E "cfg3 pair|passed|failed" $R/r06_b3_cfg3.log | cut -c1-1200
timeout 300 python scripts/r06/bench_blur.py > $R/r06_b3_blur.jsonl 2>&1; cat $R/r06_b3_blur.jsonl | cut -c1-300
run() { # name, args...
  n=$1; shift
  timeout 600 python bench.py "$@" --steps 4 --warmup 2 --loop-graph --no-split-leg --no-cpu-baseline --no-other-configs > $R/r06_b3_$n.json 2>> $R/r06_b3_bench.err
  python -c "
import json; d=json.loads(open('$R/r06_b3_$n.json').read().strip().splitlines()[-1]); print('$n', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config'].get('batch_lanes'), d['roofline']['package_during_timed_steps']['socket_power_w_median'], d['roofline']['package_during_timed_steps']['sclk_mhz_median'])"
}
run b4_auto --batch 4
run b4_l2_nosplit --batch 4 --lanes 2 --no-tail-split
run b4_l1_nosplit --batch 4 --lanes 1 --no-tail-split
run b4_l2_bf16x3 --batch 4 --lanes 2 --bf16x3
run b8_l2_nosplit --batch 8 --lanes 2 --no-tail-split
run b8_l2_bf16x3 --batch 8 --lanes 2 --bf16x3
run b8_l4 --batch 8 --lanes 4
run b16_l2_nosplit --batch 16 --lanes 2 --no-tail-split
run b16_l4 --batch 16 --lanes 4
run b32_l2_nosplit --batch 32 --lanes 2 --no-tail-split
tail -3 $R/r06_b3_bench.err
