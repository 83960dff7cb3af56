#!/bin/bash
# cfg4 training step: kernel table at the final code
cd $GRAFT_REPO_ROOT
R=gpurun_out; mkdir -p $R
export TMPDIR=/tmp
(cd /tmp && N=2 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/r06_cfg4_prof -o cfg4 -- python $GRAFT_REPO_ROOT/scripts/r03/prof_cfg4.py 2>/dev/null | grep cfg4)
DB=$(find $R/r06_cfg4_prof -name "*.db" | head -1)
python3 scripts/r05/kstats.py $DB > $R/r06_cfg4_kstats.txt
rm -rf $R/r06_cfg4_prof
