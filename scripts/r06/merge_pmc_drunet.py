"""Merge the round-6 counter passes of the fp32 DRUNet call (scripts/r06/pmc_drunet.sh) into the F(4x4) row of profiles/pmc_traffic.json:
bytes per LAUNCH of a 16-slice lane = TCC_EA0_RDREQ x 128 B + TCC_EA0_WRREQ x 64 B averaged over the 56 x 2 lanes x 3 calls launches (the
guide's gfx950 rule, as scripts/r05/merge_pmc.py), matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
usage: python scripts/r06/merge_pmc_drunet.py <gpurun_out dir> <commit>"""
import csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R, commit = sys.argv[1], sys.argv[2]
SRC = ["deepinv_amd/csrc/drunet_wino4.hip", "deepinv_amd/csrc/drunet_common.hpp"]


def sha(files):
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def collect(d, counter):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(R, d) + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"].startswith(counter) and "conv3x3_wino4_kernel" in r["Kernel_Name"]:
                tot += float(r["Counter_Value"])
                n += 1
    return tot, n


rd, nrd = collect("r06_pmc_drunet_rd", "TCC_EA0_RDREQ")
wr, nwr = collect("r06_pmc_drunet_wr", "TCC_EA0_WRREQ")
mf, _ = collect("r06_pmc_drunet_sq", "SQ_VALU_MFMA_BUSY_CYCLES")
ga, _ = collect("r06_pmc_drunet_sq", "GRBM_GUI_ACTIVE")
if not (nrd and nwr):
    sys.exit("no counters collected")
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
out = json.load(open(path))
row = {"bytes_per_launch": round(rd / nrd * 128 + wr / nwr * 64), "launches_averaged": nrd, "commit": commit,
       "config": {"batch": 32, "height": 320, "width": 320},
       "launch": "a 16-slice batch lane of the 32-slice call (DRUNet.batch_lanes = 2), no channel split of the last round; counter collection "
                 "serialises the kernels: a lane's launch running alone",
       "read_bytes": round(rd / nrd * 128), "write_bytes_tallied": round(wr / nwr * 64),
       "algorithmic_bytes_per_launch": {"read": 325.5e6, "write": 217e6,
                                        "note": "mean over the 56 ResBlock convolutions of one DRUNet(2->2) forward at 16 x 320 x 320 (half of round 5's "
                                                "32-slice launch); the write counter is uncalibrated (MI355X_MICROARCH.md, HBM)"},
       "sources": SRC, "sources_sha16": sha(SRC),
       "method": "rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum (separate passes, --kernel-trace only); bytes = RDREQ x 128 + WRREQ x 64; "
                 "scripts/r06/pmc_drunet.sh on scripts/r06/pmc_drunet_lanes.py"}
if mf and ga:
    row["mfma_busy"] = round(mf / (1024.0 * ga / 8.0), 4)
    row["mfma_busy_method"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), sums over the F(4x4) launches (each running alone)"
out["conv3x3_wino4_kernel_r05_one_sequence"] = out.get("conv3x3_wino4_kernel")
out["conv3x3_wino4_kernel"] = row
json.dump(out, open(path, "w"), indent=1)
print("conv3x3_wino4_kernel", row["bytes_per_launch"], row.get("mfma_busy"), "launches", nrd)
