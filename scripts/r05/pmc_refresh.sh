#!/bin/bash
# Round-5 refresh of profiles/pmc_traffic.json at the CURRENT kernel sources (separate --pmc passes with --kernel-trace only, as the
# guide's HBM section prescribes): HBM requests + MFMA busy of the fp32 DRUNet call, HBM requests of the MRI and Tomography operators.
#   gpurun -- 'bash scripts/r05/pmc_refresh.sh <commit>'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
run() {   # run <dir> "<counters>" <cmd...>
  local d=$1 c=$2; shift 2
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace -d $R/$d -o pmc --output-format csv -- "$@" > /dev/null 2>&1); echo "$d rc=$?"
}
run r05_pmc_drunet_rd "TCC_EA0_RDREQ_sum" python $GRAFT_REPO_ROOT/scripts/bench_ops.py drunet_fp32
run r05_pmc_drunet_wr "TCC_EA0_WRREQ_sum" python $GRAFT_REPO_ROOT/scripts/bench_ops.py drunet_fp32
run r05_pmc_drunet_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" python $GRAFT_REPO_ROOT/scripts/bench_ops.py drunet_fp32
run r05_pmc_mri_rd "TCC_EA0_RDREQ_sum" $GRAFT_REPO_ROOT/scripts/r05/mri_bench $GRAFT_REPO_ROOT/deepinv_amd/libdeepinv_amd.so --reps 2
run r05_pmc_mri_wr "TCC_EA0_WRREQ_sum" $GRAFT_REPO_ROOT/scripts/r05/mri_bench $GRAFT_REPO_ROOT/deepinv_amd/libdeepinv_amd.so --reps 2
run r05_pmc_radon_rd "TCC_EA0_RDREQ_sum" python $GRAFT_REPO_ROOT/scripts/r05/bench_fan.py
run r05_pmc_radon_wr "TCC_EA0_WRREQ_sum" python $GRAFT_REPO_ROOT/scripts/r05/bench_fan.py
python3 scripts/r05/merge_pmc.py $R ${1:-unknown}
cp profiles/pmc_traffic.json $R/r05_pmc_traffic.json
