#!/bin/bash
# HBM request counters of the fp32 DRUNet call at the final code (roofline.traffic of the bench line)
cd $GRAFT_REPO_ROOT
R=gpurun_out
mkdir -p $R
scripts/pmc.sh r04f_rd "TCC_EA0_RDREQ_sum" scripts/bench_ops.py drunet_fp32 > /dev/null
scripts/pmc.sh r04f_wr "TCC_EA0_WRREQ_sum" scripts/bench_ops.py drunet_fp32 > /dev/null
python scripts/pmc_summary.py $R/pmc_r04f_rd conv3x3_wino4 | cut -c1-200
python scripts/pmc_summary.py $R/pmc_r04f_wr conv3x3_wino4 | cut -c1-200
cp profiles/pmc_traffic.json $R/pmc_traffic_before.json
python scripts/r04/merge_pmc.py $R/pmc_r04f_rd $R/pmc_r04f_wr ${1:-unknown}   # $1 = commit the library was built from
cp profiles/pmc_traffic.json $R/pmc_traffic_after.json
