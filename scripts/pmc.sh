#!/bin/bash
# usage: scripts/pmc.sh <name> "<counters>" <python-script-relative-to-repo> [args...]  -> gpurun_out/pmc_<name>/
name=$1; shift; ctrs=$1; shift
script=$GRAFT_REPO_ROOT/$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace -d $out -o $name --output-format csv -- python $script "$@" > $out/log.txt 2>&1
echo "rocprof rc=$?"; find $out -name "*.csv" | head
