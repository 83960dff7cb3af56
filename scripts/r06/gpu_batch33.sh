#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_named_shapes_gpu.py -q -m gpu -s -k "cfg2_second" > /tmp/t.log 2>&1; grep "second draw\|passed\|failed\|Error" /tmp/t.log | cut -c1-400
