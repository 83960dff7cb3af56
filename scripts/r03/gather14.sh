#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > $R/r03_gpu_tests_full.log 2>&1
tail -n 3 $R/r03_gpu_tests_full.log
timeout 600 python bench.py > $R/r03_bench_final.json 2> $R/r03_bench_final.err
tail -c 300 $R/r03_bench_final.json
timeout 300 python bench.py --batch 4 --no-cpu-baseline --no-other-configs > $R/r03_bench_b4.json 2> $R/r03_bench_b4.err
cut -c1-260 $R/r03_bench_b4.json
P=scripts/pmc.sh
$P cfg4_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" scripts/r03/prof_cfg4.py
$P cfg4_rd "TCC_EA0_RDREQ_sum" scripts/r03/prof_cfg4.py
$P cfg4_wr "TCC_EA0_WRREQ_sum" scripts/r03/prof_cfg4.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/prof_bench_final -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg > $R/r03_bench_prof_final.json 2> $R/r03_bench_prof_final.err
