#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$R/r06_cfg5_prof -o c5 -- python $GRAFT_REPO_ROOT/scripts/r06/prof_cfg5_loop.py 2>&1 | grep rep)
DB=$(find $R/r06_cfg5_prof -name "*.db" | head -1)
python3 - <<P
import sqlite3
c = sqlite3.connect("$DB")
rows = c.execute("select start, end, name from kernels order by start").fetchall()
E = max(e for _, e, _ in rows)
for win_ms in (1290.0, 600.0):
    lo = E - int(win_ms * 1e6)
    sel = [(max(s, lo), e) for s, e, _ in rows if e > lo]
    tot = sum(e - s for s, e in sel) / 1e6
    cur_e, union = 0, 0
    for s, e in sel:
        if s > cur_e: union += e - s; cur_e = e
        elif e > cur_e: union += e - cur_e; cur_e = e
    print({"window_ms": win_ms, "kernels": len(sel), "sum_kernel_ms": round(tot, 1), "gpu_busy_union_ms": round(union / 1e6, 1), "busy_fraction": round(union / 1e6 / win_ms, 3)})
P
rm -rf $R/r06_cfg5_prof
