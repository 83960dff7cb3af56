"""Merge the round-5 rocprofv3 --pmc passes (scripts/r05/pmc_refresh.sh) into profiles/pmc_traffic.json.

bytes = TCC_EA0_RDREQ x 128 B (gfx950 tallies a 128-byte request of a wide coalesced read as one 64-byte unit: MI355X_MICROARCH.md,
HBM section) + TCC_EA0_WRREQ x 64 B, averaged per launch; an operator row is the sum over the kernels of one call.
Every row records the kernel source files it was measured on and their hash (`sources`, `sources_sha16`): bench.py drops a row
whose hash differs from the tree's.   usage: python scripts/r05/merge_pmc.py <gpurun_out dir> <commit>"""
import collections, csv, glob, hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R, commit = sys.argv[1], sys.argv[2]
METHOD = ("rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum (separate passes, --kernel-trace only); bytes = RDREQ x 128 + WRREQ x 64; "
          "scripts/r05/pmc_refresh.sh")


def sha(files):
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def collect(d, counter, key=lambda n: n):
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(R, d) + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"].startswith(counter):
                k = key(r["Kernel_Name"])
                if k:
                    tot[k] += float(r["Counter_Value"])
                    n[k] += 1
    return tot, n


path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
out = json.load(open(path))

# ---- DRUNet fp32 call: the F(4x4) kernel (whole-tile launch + tail-split launch of one call count as one call)
CONV_SRC = ["deepinv_amd/csrc/drunet_wino4.hip", "deepinv_amd/csrc/drunet_common.hpp"]
fam = lambda n: (re.search(r"(\w+_kernel)", n) or [None, None])[1]
rd, nrd = collect("r05_pmc_drunet_rd", "TCC_EA0_RDREQ", fam)
wr, nwr = collect("r05_pmc_drunet_wr", "TCC_EA0_WRREQ", fam)
calls = lambda d, c: collect(d, c, lambda n: "conv3x3_wino4_kernel" if ("wino4_kernel" in n and not re.search(r"wino4_kernel<[^>]*true>\(", n)) else None)[1]["conv3x3_wino4_kernel"]
k = "conv3x3_wino4_kernel"
if rd.get(k) and wr.get(k):
    ncall = calls("r05_pmc_drunet_rd", "TCC_EA0_RDREQ")
    row = {"bytes_per_launch": round(rd[k] / ncall * 128 + wr[k] / max(calls("r05_pmc_drunet_wr", "TCC_EA0_WRREQ"), 1) * 64),
           "launches_averaged": ncall, "commit": commit, "config": {"batch": 32, "height": 320, "width": 320},
           "read_bytes": round(rd[k] / ncall * 128), "write_bytes_tallied": round(wr[k] / max(calls("r05_pmc_drunet_wr", "TCC_EA0_WRREQ"), 1) * 64),
           "algorithmic_bytes_per_launch": {"read": 651e6, "write": 434e6,
                                            "note": "mean over the 56 ResBlock convolutions of one DRUNet(2->2) forward at 32 x 320 x 320; the write "
                                                    "counter is uncalibrated (MI355X_MICROARCH.md, HBM): requests of 64 and 128 bytes are tallied alike"},
           "sources": CONV_SRC, "sources_sha16": sha(CONV_SRC), "method": METHOD + " on scripts/bench_ops.py drunet_fp32"}
    mf, _ = collect("r05_pmc_drunet_sq", "SQ_VALU_MFMA_BUSY_CYCLES", fam)
    ga, _ = collect("r05_pmc_drunet_sq", "GRBM_GUI_ACTIVE", fam)
    if mf.get(k) and ga.get(k):
        # SQ_VALU_MFMA_BUSY_CYCLES: SIMD-cycles with the matrix pipe busy, summed over the chip's 1024 SIMDs;
        # GRBM_GUI_ACTIVE: active cycles summed over the 8 XCDs -> busy fraction = MFMA_BUSY / (1024 x GRBM / 8)
        row["mfma_busy"] = round(mf[k] / (1024.0 * ga[k] / 8.0), 4)
        row["mfma_busy_method"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), sums over the F(4x4) launches of one "
                                   "DRUNet call (a profiled pass: clocks ~5 % below an unprofiled run)")
    out[k] = row
    print(k, row["bytes_per_launch"], row.get("mfma_busy"))


# ---- operators: sum over the kernels of one call
def op_rows(tag, ops, sources, batch):
    # keyed by (kernel name, grid): the harness also runs smaller problems through kernels of the same name - the LARGEST grid
    # of a name is the named configuration
    def collect_g(d, counter):
        tot, n = collections.Counter(), collections.Counter()
        for f in glob.glob(os.path.join(R, d) + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"].startswith(counter):
                    k = (r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0))
                    tot[k] += float(r["Counter_Value"])
                    n[k] += 1
        return tot, n
    rd, nrd = collect_g(f"r05_pmc_{tag}_rd", "TCC_EA0_RDREQ")
    wr, nwr = collect_g(f"r05_pmc_{tag}_wr", "TCC_EA0_WRREQ")
    for op, pats in ops.items():
        total, used = 0.0, []
        for p in pats:
            names = sorted((k for k in rd if re.search(p, k[0])), key=lambda k: -k[1])
            if not names:
                total = None
                break
            n = names[0]
            total += rd[n] / nrd[n] * 128 + wr.get(n, 0.0) / max(nwr.get(n, 1), 1) * 64
            used.append(re.sub(r"\(.*$", "", n[0].replace("(anonymous namespace)::", "").replace("void ", ""))[:90])
        if total is not None:
            out["op:" + op] = {"bytes_per_call": round(total), "batch": batch, "commit": commit, "kernels": used, "sources": sources,
                               "sources_sha16": sha(sources), "method": METHOD}
            print(op, round(total / 1e6, 1), "MB")


MRI_SRC = ["deepinv_amd/csrc/mri.hip", "deepinv_amd/csrc/mri_wave.hpp", "deepinv_amd/csrc/fft_wave.hpp", "deepinv_amd/csrc/fft_static.hpp",
           "deepinv_amd/csrc/fft_launch.hpp", "deepinv_amd/csrc/fft_core.hpp"]
# (cfg2's kernels: the 320 x 320 instantiations; scripts/r05/mri_bench also runs the cfg4 volume and a 4-slice batch, whose kernels
# have other template arguments or grids - per-launch averages are taken per kernel NAME, so only the 5-row instantiations count here)
op_rows("mri", {"MultiCoilMRI.A@cfg2": [r"rows_dif_kernel<.*320.*, 5, false", r"cols64_kernel<5, 0"],
                "MultiCoilMRI.A_adjoint@cfg2": [r"rows_dif_kernel<.*320.*, 5, true", r"cols64_combine_kernel<5"],
                "MultiCoilMRI.A_adjoint_A@cfg2": [r"rows_dif_kernel<.*320.*, 5, false", r"cols64_kernel<5, 2", r"rows_combine_kernel<.*320.*, 5"]},
        MRI_SRC, 32)
RAD_SRC = ["deepinv_amd/csrc/radon_tiled.hip", "deepinv_amd/csrc/radon.hip"]
op_rows("radon", {"Tomography.A@cfg3": [r"radon_pack_image2", r"radon_fwd_tiled_kernel<8, false", r"radon_fwd_tiled_kernel<8, true"],
                  "Tomography.A_adjoint@cfg3": [r"radon_pack_sino2", r"radon_adj_tiled_kernel"],
                  "Tomography(fan_beam).A_adjoint@cfg3-geometry": [r"radon_fan_adj_kernel"]}, RAD_SRC, 8)
json.dump(out, open(path, "w"), indent=1)
