#!/bin/bash
# cfg4 training step: kernel table at the final code, and the cfg3 tests at the re-stated CG-count bound
cd $GRAFT_REPO_ROOT
R=gpurun_out; mkdir -p $R
timeout 900 python -m pytest tests/test_named_shapes_gpu.py -q -m gpu -k "cfg3" 2>&1 | tail -3
N=3 timeout 300 python scripts/r03/prof_cfg4.py 2>&1 | grep cfg4
export TMPDIR=/tmp
(cd /tmp && N=2 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/r06_cfg4_prof -o cfg4 -- python $GRAFT_REPO_ROOT/scripts/r03/prof_cfg4.py 2>/dev/null | grep cfg4)
DB=$(find $R/r06_cfg4_prof -name "*.db" | head -1)
python3 scripts/r05/kstats.py $DB > $R/r06_cfg4_kstats.txt
CSV=$(find $R/r06_cfg4_prof -name "*kernel_stats.csv" | head -1)
cp $CSV $R/r06_cfg4_kernel_stats.csv
head -45 $R/r06_cfg4_kernel_stats.csv | cut -c1-230
rm -rf $R/r06_cfg4_prof
