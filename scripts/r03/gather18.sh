#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 400 python -m pytest tests/test_drunet_gpu.py -m gpu -q -k "wsplit" > $R/r03_ws_tests.log 2>&1
tail -n 3 $R/r03_ws_tests.log
: > $R/r03_wsplit3.jsonl
timeout 150 python scripts/r03/bench_wsplit.py 32 >> $R/r03_wsplit3.jsonl 2>> $R/r03_wsplit3.err
timeout 150 python scripts/r03/bench_wsplit.py 4 >> $R/r03_wsplit3.jsonl 2>> $R/r03_wsplit3.err
python - <<'PY'
import json
for l in open('gpurun_out/r03_wsplit3.jsonl'):
    d=json.loads(l)
    if 'lvl' in d: print(d['B'], d['lvl'], 'direct', d.get('direct_conv1_ms'), d.get('direct_conv2_ms'), 'ws', d.get('wsplit_conv1_ms'), d.get('wsplit_conv2_ms'), 'zero', d.get('wsplit_conv2_zero_data_ms'), d.get('err_wsplit'), d.get('error'))
    else: print(d)
PY
