#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_named_shapes_gpu.py -q -m gpu -s -k "cfg5" --durations=5 > /tmp/t.log 2>&1; grep "cfg5 full\|passed\|failed\|Error\|s call" /tmp/t.log | cut -c1-300
