#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
: > $R/r03_wsplit7.jsonl
for v in base order2; do timeout 120 python scripts/r03/bench_wsplit.py 32 $v >> $R/r03_wsplit7.jsonl 2>> $R/r03_wsplit7.err; done
python - <<'PY'
import json
for l in open('gpurun_out/r03_wsplit7.jsonl'):
    d=json.loads(l)
    if 'lvl' in d: print(d['variant'], d['lvl'], d.get('wsplit_conv1_ms'), d.get('wsplit_conv2_ms'), d.get('wsplit_conv2_zero_data_ms'), d.get('err_wsplit'), d.get('error'))
    else: print(d)
PY
