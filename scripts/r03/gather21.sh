#!/bin/bash
# validation of the final round-3 code: whole GPU suite, default bench line, batch-4 bench line
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q > $R/r03_gpu_tests_ws.log 2>&1
tail -n 15 $R/r03_gpu_tests_ws.log
timeout 500 python bench.py > $R/r03_bench_ws.json 2> $R/r03_bench_ws.err
tail -c 400 $R/r03_bench_ws.json
timeout 200 python bench.py --batch 4 --no-cpu-baseline --no-other-configs > $R/r03_bench_ws_b4.json 2> $R/r03_bench_ws_b4.err
cut -c1-260 $R/r03_bench_ws_b4.json
