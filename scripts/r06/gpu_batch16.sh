#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$R/r06_ov4_prof -o b -- python $GRAFT_REPO_ROOT/bench.py --batch 4 --steps 1 --warmup 1 --iters 12 --loop-graph --no-split-leg --no-cpu-baseline --no-other-configs > /dev/null 2>&1); echo "prof rc=$?"
DB=$(find $R/r06_ov4_prof -name "*.db" | head -1)
python3 scripts/r05/kstats.py $DB "wino4|conv3x3|down2x2|up2x2|tail|pack|rows|cols|lincomb" > $R/r06_b4_lanes_kernel_stats.txt; cat $R/r06_b4_lanes_kernel_stats.txt | cut -c1-190
python3 scripts/r06/wino4_calls.py $DB | tee -a $R/r06_b4_lanes_kernel_stats.txt
python3 - <<P
import sqlite3, sys
c = sqlite3.connect("$DB")
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the last 6 iterations of the timed step: union busy time of ALL kernels vs span
rows = rows[-6 * 140:]
span = (rows[-1][2] - rows[0][1]) / 1e6
union, cs, ce = 0.0, None, None
for _, s, e in rows:
    if ce is None or s > ce:
        if ce is not None: union += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
tot = sum(e - s for _, s, e in rows) / 1e6
print(f"last {len(rows)} kernels: span {span:.2f} ms, device busy (union) {union / 1e6:.2f} ms, sum of durations {tot:.2f} ms, mean concurrency {tot / (union / 1e6):.2f}")
P
rm -rf $R/r06_ov4_prof
