#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$R/r06_ov_prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --iters 12 --no-split-leg --no-cpu-baseline --no-other-configs > /dev/null 2>&1); echo "prof rc=$?"
DB=$(find $R/r06_ov_prof -name "*.db" | head -1)
python3 scripts/r06/overlaps.py $DB "rows_dif|cols64|rows_combine|lincomb|tail3x3|up2x2|down2x2|pack_kernel" | tee $R/r06_overlaps.txt
rm -rf $R/r06_ov_prof
