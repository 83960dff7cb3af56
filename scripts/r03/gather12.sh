#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > $R/r03_gpu_tests_full.log 2>&1
tail -n 3 $R/r03_gpu_tests_full.log
timeout 600 python bench.py > $R/r03_bench_final.json 2> $R/r03_bench_final.err
tail -c 600 $R/r03_bench_final.json
timeout 300 python bench.py --batch 4 --no-cpu-baseline --no-other-configs > $R/r03_bench_b4.json 2> $R/r03_bench_b4.err
cut -c1-300 $R/r03_bench_b4.json
timeout 300 python scripts/bench_ops.py mri2d mri3d radon drunet drunet4 > $R/r03_ops_final.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/prof_bench_final -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg > $R/r03_bench_prof_final.json 2> $R/r03_bench_prof_final.err
