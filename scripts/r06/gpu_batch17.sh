#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out
timeout 900 python -m pytest tests/test_drunet_gpu.py -q -m gpu -x -k "tail or drunet_forward or model" > $R/r06_b17_tests.log 2>&1; echo "tests rc=$?"; tail -2 $R/r06_b17_tests.log
for b in 4 8 32; do
  timeout 600 python bench.py --batch $b --steps 4 --warmup 2 --loop-graph --no-split-leg --no-cpu-baseline --no-other-configs > $R/r06_b17_b$b.json 2>> $R/r06_b17.err
  python -c "
import json; d=json.loads(open('$R/r06_b17_b$b.json').read().strip().splitlines()[-1]); print($b, d['ms_per_step'], d['config']['batch_lanes'], d['config']['batch_lanes_calibration'])"
done
