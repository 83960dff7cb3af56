#!/bin/bash
# round-6 refresh of the F(4x4) row of profiles/pmc_traffic.json: the launches as they run now (16-slice lanes, no channel split)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
run() { local d=$1 c=$2; shift 2
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace -d $R/$d -o pmc --output-format csv -- "$@" > /dev/null 2>&1); echo "$d rc=$?"
}
run r06_pmc_drunet_rd "TCC_EA0_RDREQ_sum" python $GRAFT_REPO_ROOT/scripts/r06/pmc_drunet_lanes.py
run r06_pmc_drunet_wr "TCC_EA0_WRREQ_sum" python $GRAFT_REPO_ROOT/scripts/r06/pmc_drunet_lanes.py
run r06_pmc_drunet_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" python $GRAFT_REPO_ROOT/scripts/r06/pmc_drunet_lanes.py
python3 scripts/r06/merge_pmc_drunet.py $R ${1:-unknown}
cp profiles/pmc_traffic.json $R/r06_pmc_traffic.json
rm -rf $R/r06_pmc_drunet_*
