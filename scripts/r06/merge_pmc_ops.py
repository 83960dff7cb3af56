"""Merge the counter passes of scripts/r06/pmc_ops.sh into profiles/pmc_traffic.json: bytes per call of an operator = (sum over the
product's kernels launched by scripts/r06/pmc_op.py of TCC_EA0_RDREQ x 128 B + TCC_EA0_WRREQ x 64 B) / calls - the rule of the guide's
HBM section for wide coalesced streams (MI355X_MICROARCH.md), as in scripts/r05/merge_pmc.py.  ATen kernels (tensor fills of the
harness) are excluded by name.   usage: python scripts/r06/merge_pmc_ops.py <gpurun_out dir> <commit>"""
import csv, glob, hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R, commit = sys.argv[1], sys.argv[2]
OPS = ["BlurFFT.A", "BlurFFT.A_adjoint", "BlurFFT.prox_l2", "Blur(circular).A", "Blur(circular).A_adjoint", "Blur(valid).A",
       "Blur(valid).A_adjoint", "MRI.A", "MRI.A_adjoint"]
CFG = {"B": "cfg1-shape x32", "M": "single coil 320x320 x32"}
SRC = {"B": ["deepinv_amd/csrc/blur.hip", "deepinv_amd/csrc/fft_static.hpp", "deepinv_amd/csrc/fft_launch.hpp", "deepinv_amd/csrc/fft_core.hpp",
             "deepinv_amd/csrc/elementwise.hip"],
       "M": ["deepinv_amd/csrc/mri.hip", "deepinv_amd/csrc/mri_wave.hpp", "deepinv_amd/csrc/fft_wave.hpp", "deepinv_amd/csrc/fft_static.hpp",
             "deepinv_amd/csrc/fft_launch.hpp", "deepinv_amd/csrc/fft_core.hpp"]}
OURS = re.compile(r"dinv|anonymous namespace|mriw|fft_|blurfft|conv2d_|lincomb|spectrum")


def sha(files):
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def total(d, counter):
    tot, kernels = 0.0, {}
    for f in glob.glob(os.path.join(R, d) + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"].startswith(counter) and OURS.search(r["Kernel_Name"]) and "at::native" not in r["Kernel_Name"]:
                tot += float(r["Counter_Value"])
                short = re.sub(r"<.*", "", re.sub(r"\(anonymous namespace\)::|dinv::|void ", "", r["Kernel_Name"]))
                kernels[short] = kernels.get(short, 0) + 1
    return tot, kernels


path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
out = json.load(open(path))
calls = 10
for i, op in enumerate(OPS, 1):
    rd, kr = total(f"r06_pmc_op{i}_RDREQ", "TCC_EA0_RDREQ")
    wr, _ = total(f"r06_pmc_op{i}_WRREQ", "TCC_EA0_WRREQ")
    if not rd or not wr:
        print("no counters for", op)
        continue
    fam = "M" if op.startswith("MRI") else "B"
    row = {"bytes_per_call": round((rd * 128 + wr * 64) / calls), "read_bytes": round(rd * 128 / calls), "write_bytes_tallied": round(wr * 64 / calls),
           "batch": 32, "calls_averaged": calls, "kernels_per_call": {k: v / calls for k, v in kr.items()}, "commit": commit,
           "sources": SRC[fam], "sources_sha16": sha(SRC[fam]),
           "method": "rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum (separate passes, --kernel-trace only); bytes = RDREQ x 128 + WRREQ x 64; "
                     "scripts/r06/pmc_ops.sh (one operator per process, 10 calls)"}
    out[f"op:{op}@{CFG[fam]}"] = row
    print(op, row["bytes_per_call"], row["kernels_per_call"])
json.dump(out, open(path, "w"), indent=1)
