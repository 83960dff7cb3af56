#!/bin/bash
# usage: scripts/prof.sh <name> <command...>   -> gpurun_out/prof_<name>/ (kernel trace + stats)
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o $name --output-format csv -- "$@" > $out/log.txt 2>&1
echo "rocprof rc=$?"; ls -R $out | head -20
