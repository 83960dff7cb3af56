"""Round-3 artefacts for profiles/: PMC summaries (per kernel, per launch) and profiles/pmc_traffic.json from the rocprofv3
--pmc passes kept under gpurun_out/ (HBM bytes = TCC_EA0_RDREQ x 64 B x 2 [gfx950 wide-read correction, MI355X_MICROARCH.md]
+ TCC_EA0_WRREQ x 64 B).  usage: python scripts/r03/make_profiles.py <commit>"""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
commit = sys.argv[1]


def per_kernel(d):
    """{kernel name: {counter: mean per launch}, 'launches': n}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(os.path.join(G, d) + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    return {k: ({c: v / cnt[k][c] for c, v in cs.items()}, max(cnt[k].values())) for k, cs in agg.items()}


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("dinv::", "").replace("void ", "")
    return re.sub(r"\(.*", "", k)[:110]


def write_summary(name, dirs, note):
    rows = collections.defaultdict(dict)
    launches = {}
    for d in dirs:
        for k, (cs, n) in per_kernel(d).items():
            if any(s in k for s in ("at::native", "rocclr", "Cijk")):
                continue
            rows[short(k)].update({c: round(v) for c, v in cs.items()})
            launches[short(k)] = n
    cols = sorted({c for r in rows.values() for c in r})
    with open(os.path.join(P, "pmc", name), "w") as f:
        f.write("# " + note + "\n")
        f.write("kernel,launches," + ",".join(cols) + "\n")
        for k, r in sorted(rows.items()):
            f.write('"%s",%d,' % (k, launches[k]) + ",".join(str(r.get(c, "")) for c in cols) + "\n")
    return rows


os.makedirs(os.path.join(P, "pmc"), exist_ok=True)
conv = write_summary("r03_conv_split2d_sq.csv", ["pmc_s2d_final_s1", "pmc_s2d_final_s2"],
                     "scripts/r03/prof_split2d.py 1 32 3 (level 1: 128 ch, 160x160, B=32): SQ counters of conv3x3_split2d_kernel, mean per launch")
radon = write_summary("r03_radon_sq.csv", ["pmc_radon3"], "scripts/bench_ops.py radon (B=8, 512x512, 720 angles): SQ / LDS counters, mean per launch")
dr = write_summary("r03_drunet_hbm.csv", ["pmc_drd3", "pmc_dwr3"], "scripts/bench_ops.py drunet (one DRUNet call, B=32, 320x320): TCC_EA0 requests per launch")
mr = write_summary("r03_mri_hbm.csv", ["pmc_mrd3", "pmc_mwr3"], "scripts/bench_ops.py mri2d mri3d: TCC_EA0 requests per launch")
rr = write_summary("r03_radon_hbm.csv", ["pmc_radon_rd", "pmc_radon_wr"], "scripts/bench_ops.py radon: TCC_EA0 requests per launch")


def hbm(row):
    return row.get("TCC_EA0_RDREQ_sum", 0) * 128 + row.get("TCC_EA0_WRREQ_sum", 0) * 64


def pick(rows, *subs):
    hit = [k for k in rows if all(s in k for s in subs)]
    assert len(hit) >= 1, (subs, list(rows))
    return sum(hbm(rows[k]) for k in hit) / len(hit)


out = {}
meth = "rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum (separate passes); bytes = RDREQ x 128 + WRREQ x 64"
# conv kernels over one DRUNet call (B = 32, 320 x 320): launch-weighted mean over the instantiations
num = den = 0
for k, r in dr.items():
    if "conv3x3_split2d_kernel" in k:
        n = 196
        num += hbm(r) * n; den += n
out["conv3x3_split2d_kernel"] = {"bytes_per_launch": round(num / den), "launches_averaged": den, "commit": commit,
                                 "config": {"batch": 32, "height": 320, "width": 320}, "method": meth + " on scripts/bench_ops.py drunet"}
k0 = [k for k in dr if k.startswith("conv3x3_kernel")][0]
out["conv3x3_kernel"] = {"bytes_per_launch": round(hbm(dr[k0])), "launches_averaged": 7, "commit": commit,
                         "config": {"batch": 32, "height": 320, "width": 320}, "method": meth + " on scripts/bench_ops.py drunet"}
ops = {
    "op:MultiCoilMRI.A@cfg2": (32, [("mri_cols_expand_fwd_kernel", "320"), ("fft_rows_static_v4_kernel", "320, 8, 8, 5")]),
    "op:MultiCoilMRI.A_adjoint@cfg2": (32, [("fft_rows_static_v4_kernel", "320, 5, 8, 8"), ("fft_cols_static_kernel", "320", "C2CIo"), ("mri_coil_combine_kernel",)]),
    "op:MultiCoilMRI.A_adjoint_A@cfg2": (32, [("mri_cols_expand_fwd_kernel", "320"), ("mri_rows_normal_kernel", "320"), ("fft_cols_static_kernel", "320", "C2CIo"), ("mri_coil_combine_kernel",)]),
}
for name, (batch, parts) in ops.items():
    out[name] = {"bytes_per_call": round(sum(pick(mr, *p) for p in parts)), "batch": batch, "commit": commit,
                 "kernels": [" ".join(p) for p in parts], "method": meth + " on scripts/bench_ops.py mri2d mri3d"}
rops = {
    "op:Tomography.A@cfg3": [("radon_pack_image2",), ("radon_fwd_tiled_kernel<8, false",), ("radon_fwd_tiled_kernel<8, true",)],
    "op:Tomography.A_adjoint@cfg3": [("radon_pack_sino2",), ("radon_adj_tiled_kernel",)],
}
for name, parts in rops.items():
    out[name] = {"bytes_per_call": round(sum(pick(rr, *p) for p in parts)), "batch": 8, "commit": commit,
                 "kernels": [" ".join(p) for p in parts], "method": meth + " on scripts/bench_ops.py radon"}
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v.get("bytes_per_launch") or v.get("bytes_per_call")) for k, v in out.items()}, indent=1))
