#!/bin/bash
# counter passes of the Tomography operator rows at the current radon sources (round 5's recipe: scripts/r05/pmc_refresh.sh, radon part)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
for c in rd:RDREQ wr:WRREQ; do
  d=r05_pmc_radon_${c%%:*}
  (cd /tmp && timeout 400 rocprofv3 --pmc TCC_EA0_${c##*:}_sum --kernel-trace -d $R/$d -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/r05/bench_fan.py > /dev/null 2>&1); echo "$d rc=$?"
done
python3 scripts/r05/merge_pmc.py $R ${1:-unknown} | grep -i "tomo\|radon"
cp profiles/pmc_traffic.json $R/r06_pmc_traffic.json
rm -rf $R/r05_pmc_radon_rd $R/r05_pmc_radon_wr
