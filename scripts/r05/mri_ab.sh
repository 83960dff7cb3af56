#!/bin/bash
# A/B of the MRI passes on the GPU box: scripts/r05/mri_bench over the variant libraries (+ the product's), then a rocprofv3 kernel
# trace of the same run.   mri_ab.sh <tag>
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
V=scripts/r05/variants
LIBS="$V/libv0.so $(ls $V/lib*.so | grep -v libv0.so) deepinv_amd/libdeepinv_amd.so"
timeout 300 scripts/r05/mri_bench $LIBS --reps 30 > $R/r05_mri_ab_${1:-1}.jsonl 2>&1; echo "bench rc=$?"
cat $R/r05_mri_ab_${1:-1}.jsonl | cut -c1-330
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/r05_mri_prof_${1:-1} -o mri -- $GRAFT_REPO_ROOT/scripts/r05/mri_bench $GRAFT_REPO_ROOT/deepinv_amd/libdeepinv_amd.so --reps 10 > /dev/null 2>&1); echo "prof rc=$?"
python3 scripts/r05/kstats.py $(find $R/r05_mri_prof_${1:-1} -name "*.db" | head -1) | cut -c1-220
