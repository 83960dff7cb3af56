#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_loops_gpu.py -q -m gpu -k "lanes or preflight" 2>&1 | tail -2
timeout 600 python bench.py --batch 4 --steps 3 --warmup 1 --loop-graph --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['batch_lanes'], d['config']['batch_lanes_calibration'])"
