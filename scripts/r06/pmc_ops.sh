#!/bin/bash
# counter passes of the round-6 operator rows (BlurFFT / Blur / single-coil MRI): separate --pmc passes with --kernel-trace only
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
i=0
for op in "BlurFFT.A" "BlurFFT.A_adjoint" "BlurFFT.prox_l2" "Blur(circular).A" "Blur(circular).A_adjoint" "Blur(valid).A" "Blur(valid).A_adjoint" "MRI.A" "MRI.A_adjoint"; do
  i=$((i+1))
  for c in RDREQ WRREQ; do
    (cd /tmp && timeout 200 rocprofv3 --pmc TCC_EA0_${c}_sum --kernel-trace -d $R/r06_pmc_op${i}_$c -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/r06/pmc_op.py "$op" 10 > /dev/null 2>&1); echo "$op $c rc=$?"
  done
done
python3 scripts/r06/merge_pmc_ops.py $R ${1:-unknown}
cp profiles/pmc_traffic.json $R/r06_pmc_traffic.json
rm -rf $R/r06_pmc_op*
