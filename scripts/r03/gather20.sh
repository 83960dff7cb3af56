#!/bin/bash
cd $GRAFT_REPO_ROOT
P=scripts/pmc.sh
for v in base nostage nostage_noa; do
$P wsv_$v "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" scripts/r03/prof_wsplit.py 2 32 3 $v > /dev/null
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for v in ("base","nostage","nostage_noa"):
    d=f"gpurun_out/pmc_wsv_{v}"
    dur=collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(d+"/*kernel_trace.csv")[0])):
        if "wsplit" in r["Kernel_Name"]: dur[r["Kernel_Name"][40:75]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob(d+"/*counter_collection.csv")[0])):
        if "wsplit" in r["Kernel_Name"]: agg[r["Kernel_Name"][40:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        c={n:sum(x)/len(x) for n,x in agg[k].items()}
        t=sum(dur[k])/len(dur[k])
        print(v,k,"dur_us",round(t/1e3,1),"clk_GHz",round(c["GRBM_GUI_ACTIVE"]/8/t,3),"mfma_busy_frac",round(c["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(c["GRBM_GUI_ACTIVE"]/8),3),{n:round(x/1e6,1) for n,x in c.items()})
PY
