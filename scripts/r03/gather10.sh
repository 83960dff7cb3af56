#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_drunet_gpu.py tests/test_named_shapes_gpu.py tests/test_trainer_gpu.py -m gpu -x -q -k "3d or backward or wgrad or train or cfg4 or unfolded" > $R/r03_gpu_tests_c.log 2>&1
tail -3 $R/r03_gpu_tests_c.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/prof_cfg4b -o cfg4 --output-format csv -- python $GRAFT_REPO_ROOT/scripts/r03/prof_cfg4.py > $R/r03_cfg4b.json 2>&1
N=3 python $GRAFT_REPO_ROOT/scripts/r03/prof_cfg4.py > $R/r03_cfg4b_plain.json 2>&1
N=3 python $GRAFT_REPO_ROOT/scripts/r03/prof_cfg4.py bf16split > $R/r03_cfg4b_bf16split.json 2>&1
tail -1 $R/r03_cfg4b_plain.json $R/r03_cfg4b_bf16split.json
timeout 600 python $GRAFT_REPO_ROOT/scripts/bench_train.py > $R/r03_train_step.jsonl 2>&1
tail -n 3 $R/r03_train_step.jsonl | cut -c1-300
