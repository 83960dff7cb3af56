#!/bin/bash
# fifth hardware run: the Python path - new GPU tests, the cfg2 reference fixture test, the bench line
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
timeout 600 python -m pytest tests/test_drunet_gpu.py -q -x -k "winograd4 or fp32_precision or drunet3d or default_precision" > $R/r04_tests_a.log 2>&1; echo "tests a rc=$?"; tail -4 $R/r04_tests_a.log
timeout 400 python -m pytest tests/test_named_shapes_gpu.py -q -x -k "cfg2" > $R/r04_tests_b.log 2>&1; echo "tests b rc=$?"; tail -6 $R/r04_tests_b.log
timeout 600 python bench.py --steps 5 --warmup 2 > $R/r04_bench_a.json 2> $R/r04_bench_a.err; echo "bench rc=$?"; tail -3 $R/r04_bench_a.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_bench_a.json').read().strip().splitlines()[-1])
ops=d.pop('operators'); cb=d.pop('cpu_baseline',None)
print(json.dumps(d)[:3500])
for o in ops: print(o)
print(cb)
P
