#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/r03/mri_ab.py > gpurun_out/r03_mri_ab.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_mri_ab -o mri --output-format csv -- python $GRAFT_REPO_ROOT/scripts/r03/mri_ab.py > /dev/null 2>&1
python -c "import os; print(os.cpu_count(), len(os.sched_getaffinity(0)))" > $GRAFT_REPO_ROOT/gpurun_out/r03_cpuinfo.txt
cat /sys/fs/cgroup/cpu.max >> $GRAFT_REPO_ROOT/gpurun_out/r03_cpuinfo.txt 2>&1
nproc >> $GRAFT_REPO_ROOT/gpurun_out/r03_cpuinfo.txt
