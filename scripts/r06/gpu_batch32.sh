#!/bin/bash
# the rebuilt library of the final tree: smoke, a cross-section of the GPU suite, the short bench
cd $GRAFT_REPO_ROOT
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -q -m gpu -x -k "tail or lanes or mri or blur or radon or golden or generators" 2>&1 | tail -2
timeout 600 python bench.py --steps 2 --warmup 1 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
