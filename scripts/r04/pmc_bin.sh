#!/bin/bash
# usage: scripts/r04/pmc_bin.sh <name> "<counters>" <binary-relative-to-repo> [args...]  -> gpurun_out/pmc_<name>/ (counters only with --kernel-trace)
name=$1; shift; ctrs=$1; shift
bin=$GRAFT_REPO_ROOT/$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
mkdir -p $out
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/deepinv_amd:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc $ctrs --kernel-trace -d $out -o $name --output-format csv -- $bin "$@" > $out/log.txt 2>&1
echo "rocprof $name rc=$?"
