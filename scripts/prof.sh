#!/bin/bash
# usage: scripts/prof.sh <name> <python-script-relative-to-repo> [args...]
#   -> gpurun_out/prof_<name>/  (rocprofv3 kernel trace + per-kernel stats, csv)
name=$1; shift
script=$GRAFT_REPO_ROOT/$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o $name --output-format csv -- python $script "$@" > $out/log.txt 2>&1
echo "rocprof rc=$?"; find $out -name "*.csv" | head
