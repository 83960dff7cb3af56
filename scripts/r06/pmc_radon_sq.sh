#!/bin/bash
# SQ counters of the parallel-beam Radon kernels (where do the cycles go: vector ALU, LDS, waiting?)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
run() { local d=$1 c=$2; shift 2
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace -d $R/$d -o pmc --output-format csv -- "$@" > /dev/null 2>&1); echo "$d rc=$?"
}
run r06_sq_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" python $GRAFT_REPO_ROOT/scripts/r05/bench_fan.py
run r06_sq_b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" python $GRAFT_REPO_ROOT/scripts/r05/bench_fan.py
python3 - <<P
import csv, glob, collections, re
for d in ("r06_sq_a", "r06_sq_b"):
    tot = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
    for f in glob.glob("$R/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
            if "radon" in k:
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU"):
                    n[k] += 1
    for k, v in tot.items():
        print(d, k, "launches", n[k], {c: round(x / max(n[k], 1)) for c, x in v.items()})
P
rm -rf $R/r06_sq_a $R/r06_sq_b
