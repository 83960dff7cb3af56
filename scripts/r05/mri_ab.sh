#!/bin/bash
# A/B of the MRI passes on the GPU box: scripts/r05/mri_bench over the variant libraries, then a rocprofv3 kernel trace of two of them
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
V=scripts/r05/variants
LIBS="$V/libv0.so $(ls $V/lib*.so | grep -v libv0.so)"
timeout 300 scripts/r05/mri_bench $LIBS --reps 30 > $R/r05_mri_ab_${1:-1}.jsonl 2>&1; echo "bench rc=$?"
cat $R/r05_mri_ab_${1:-1}.jsonl | cut -c1-330
export TMPDIR=/tmp
PROF="${2:-$V/libv0.so $V/libv4.so}"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/r05_mri_prof_${1:-1} -o mri -- $GRAFT_REPO_ROOT/scripts/r05/mri_bench $(for l in $PROF; do echo $GRAFT_REPO_ROOT/$l; done) --reps 10 > /dev/null 2>&1); echo "prof rc=$?"
find $R/r05_mri_prof_${1:-1} -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-200 {} | head -40'
