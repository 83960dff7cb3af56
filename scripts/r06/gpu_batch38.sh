#!/bin/bash
# the whole GPU suite and the smoke test at the last commit of the round
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06_gpu_tests_last.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r06_gpu_tests_last.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
