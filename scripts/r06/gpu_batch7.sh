#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 600 python -m pytest tests/test_mri_gpu.py tests/test_loops_gpu.py -q -m gpu -x > $R/r06_b7_tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/r06_b7_tests.log
timeout 600 python scripts/r06/bench_mri_lanes.py 2>&1 | grep -v amdgpu.ids | tee $R/r06_mri_lanes.jsonl
# the multi-GPU code path on one rank (RCCL process group alive: its streams beside the lanes)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --as-multi --batch 4 --steps 4 --warmup 2 --no-split-leg > $R/r06_b7_as_multi_b4.json 2>> $R/r06_b7.err
python -c "
import json; d=json.loads([l for l in open('$R/r06_b7_as_multi_b4.json') if l.startswith('{')][-1]); print('as-multi b4', d['ms_per_step'], d['config'])"
