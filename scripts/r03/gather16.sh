#!/bin/bash
# first hardware run of the Winograd F(2,3) bf16-split conv: parity tests, timing against the direct kernel, diagnostic builds
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 400 python -m pytest tests/test_drunet_gpu.py -m gpu -q -k "wsplit" > $R/r03_ws_tests.log 2>&1
tail -n 5 $R/r03_ws_tests.log
: > $R/r03_wsplit.jsonl
timeout 150 python scripts/r03/bench_wsplit.py 32 >> $R/r03_wsplit.jsonl 2>> $R/r03_wsplit.err
for v in notransform noexchange nofence; do timeout 120 python scripts/r03/bench_wsplit.py 32 $v >> $R/r03_wsplit.jsonl 2>> $R/r03_wsplit.err; done
timeout 120 python scripts/r03/bench_wsplit.py 4 >> $R/r03_wsplit.jsonl 2>> $R/r03_wsplit.err
cut -c1-420 $R/r03_wsplit.jsonl
tail -n 5 $R/r03_wsplit.err
