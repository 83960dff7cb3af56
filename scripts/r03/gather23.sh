#!/bin/bash
# counter passes (HBM requests of one DRUNet call) and the rocprofv3 kernel statistics of the bench command, final round-3 code
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
P=scripts/pmc.sh
$P wrd "TCC_EA0_RDREQ_sum" scripts/bench_ops.py drunet > /dev/null
$P wwr "TCC_EA0_WRREQ_sum" scripts/bench_ops.py drunet > /dev/null
for d in gpurun_out/pmc_wrd gpurun_out/pmc_wwr; do python scripts/pmc_summary.py $d conv3x3 | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/prof_bench_ws -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-other-configs > $R/r03_bench_prof_ws.json 2> $R/r03_bench_prof_ws.err
ls $R/prof_bench_ws | head; cut -c1-300 $R/r03_bench_prof_ws.json
