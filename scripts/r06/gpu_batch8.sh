#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
am() { # name, env..., args
  n=$1; shift
  timeout 600 env "$@" > $R/r06_b8_$n.json 2>> $R/r06_b8.err
  python -c "
import json; d=json.loads([l for l in open('$R/r06_b8_$n.json') if l.startswith('{')][-1]); print('$n', d['ms_per_step'], d['config'].get('batch_lanes'), d['config'].get('loop_graph'), d['config'].get('loop_graph_error'))"
}
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --as-multi --batch 4 --steps 4 --warmup 2 --no-split-leg"
am multi_graph_l2 X=1 $TR
am multi_graph_l1 X=1 $TR --lanes 1
am multi_graph_l2_hwq8 GPU_MAX_HW_QUEUES=8 $TR
am plain_graph_l2 X=1 python bench.py --batch 4 --steps 4 --warmup 2 --no-split-leg --loop-graph --no-cpu-baseline --no-other-configs
am plain_graph_l2_hwq8 GPU_MAX_HW_QUEUES=8 python bench.py --batch 4 --steps 4 --warmup 2 --no-split-leg --loop-graph --no-cpu-baseline --no-other-configs
am plain_graph_l2_hwq2 GPU_MAX_HW_QUEUES=2 python bench.py --batch 4 --steps 4 --warmup 2 --no-split-leg --loop-graph --no-cpu-baseline --no-other-configs
python - <<P
import sys, json, torch
sys.path.insert(0, '.')
import bench, deepinv_amd as dinv
dev = torch.device('cuda:0')
for B in (8, 12, 8):
    physics, x, y, maps, mask = bench.make_problem(dinv, B, 0, 320, 320, 8, dev)
    def t_op(fn, n=50):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    print(json.dumps({"batch": B, "A": t_op(lambda: physics.A(x)), "AT": t_op(lambda: physics.A_adjoint(y)), "ATA": t_op(lambda: physics.A_adjoint_A(x))}))
P
tail -3 $R/r06_b8.err
