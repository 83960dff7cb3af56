#!/bin/bash
# last batch of round 6: counter passes of the operator rows at the final sources, then the validation batch
cd $GRAFT_REPO_ROOT
bash scripts/r06/pmc_ops.sh ${1:-unknown} 2>&1 | tail -10
bash scripts/r06/pmc_radon.sh ${1:-unknown} 2>&1 | tail -5
bash scripts/r06/validate.sh r06 prof
