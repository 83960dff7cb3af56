#!/bin/bash
# the library built without SLP vectorisation and with the hand-written packed form in conv2d_tiled: blur / radon / generator tests,
# the probe again (the product's wsplit as partner), operator timings
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu -k "blur or conv or radon or tomo or downsampl or tail or lanes" 2>&1 | tail -3
timeout 600 python scripts/r06/bench_blur.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
timeout 300 python scripts/r06/time_tail.py noslp 2>&1 | grep "batch\": 32"
