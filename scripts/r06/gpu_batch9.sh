#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 600 python -m pytest tests/test_loops_gpu.py -q -m gpu -x -k "lane or preflight" > $R/r06_b9_tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/r06_b9_tests.log
am() { n=$1; shift
  timeout 600 env "$@" > $R/r06_b9_$n.json 2>> $R/r06_b9.err
  python -c "
import json; d=json.loads([l for l in open('$R/r06_b9_$n.json') if l.startswith('{')][-1]); print('$n', d['ms_per_step'], d['config'].get('batch_lanes'), d['config'].get('loop_graph'), d['config'].get('loop_graph_error'))"
}
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --as-multi --batch 4 --steps 4 --warmup 2 --no-split-leg"
am multi_default X=1 $TR
am multi_l1 X=1 $TR --lanes 1
am multi_hwq4 GPU_MAX_HW_QUEUES=4 $TR
am multi_hwq2 GPU_MAX_HW_QUEUES=2 $TR
am plain_hwq2 GPU_MAX_HW_QUEUES=2 python bench.py --batch 4 --steps 4 --warmup 2 --no-split-leg --loop-graph --no-cpu-baseline --no-other-configs
am plain_hwq1 GPU_MAX_HW_QUEUES=1 python bench.py --batch 4 --steps 4 --warmup 2 --no-split-leg --loop-graph --no-cpu-baseline --no-other-configs
tail -3 $R/r06_b9.err
