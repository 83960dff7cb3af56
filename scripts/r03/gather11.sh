#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_tomography_gpu.py tests/test_named_shapes_gpu.py -m gpu -x -q -k "radon or tomo or Tomo or cfg3" > $R/r03_gpu_tests_d.log 2>&1
tail -n 3 $R/r03_gpu_tests_d.log
cd /tmp && export TMPDIR=/tmp
for v in main v4; do
  rocprofv3 --kernel-trace --stats -d $R/prof_var_$v -o var --output-format csv -- python $GRAFT_REPO_ROOT/scripts/r03/exp/run_variant.py $v > $R/r03_var_$v.jsonl 2>$R/r03_var_$v.err
  grep -h '"op"' $R/r03_var_$v.jsonl | cut -c1-160
done
