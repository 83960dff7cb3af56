#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out; export TMPDIR=/tmp
for w in A AT prox; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$R/r06_ds_$w -o ds -- python $GRAFT_REPO_ROOT/scripts/r06/prof_downsampling.py $w > /dev/null 2>&1)
  DB=$(find $R/r06_ds_$w -name "*.db" | head -1)
  echo "== $w"; python3 scripts/r05/kstats.py $DB | grep -v "n=   1 " | cut -c1-60,100-200
  rm -rf $R/r06_ds_$w
done
