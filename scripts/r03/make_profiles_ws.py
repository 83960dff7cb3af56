"""Round-3 artefacts of the Winograd bf16-split conv for profiles/: HBM requests per launch of the conv kernels over one
DRUNet call (rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum, separate passes kept under gpurun_out/pmc_wrd, pmc_wwr;
bytes = RDREQ x 128 [gfx950 wide-read correction, MI355X_MICROARCH.md] + WRREQ x 64) -> profiles/pmc/r03_drunet_wsplit_hbm.csv and
the `conv3x3_wsplit_kernel` / `conv3x3_kernel` entries of profiles/pmc_traffic.json.  usage: python scripts/r03/make_profiles_ws.py <commit>"""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
commit = sys.argv[1]


def per_kernel(d):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(G, d) + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(\(anonymous.*|\(dinv_drunet.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: (agg[k] / cnt[k], cnt[k]) for k in agg}


rd, wr = per_kernel("pmc_wrd"), per_kernel("pmc_wwr")
with open(os.path.join(P, "pmc", "r03_drunet_wsplit_hbm.csv"), "w") as f:
    f.write("# scripts/bench_ops.py drunet (DRUNet calls at B=32, 320x320, bf16-split precision = Winograd F(2,3) ResBlock convs): TCC_EA0 requests per launch\n")
    f.write("kernel,launches,TCC_EA0_RDREQ_sum,TCC_EA0_WRREQ_sum,HBM_MB\n")
    for k in sorted(rd):
        if any(s in k for s in ("at::native", "rocclr", "Cijk")) or k not in wr:
            continue
        f.write('"%s",%d,%d,%d,%.1f\n' % (k, rd[k][1], rd[k][0], wr[k][0], (rd[k][0] * 128 + wr[k][0] * 64) / 1e6))
meth = "rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum (separate passes); bytes = RDREQ x 128 + WRREQ x 64 on scripts/bench_ops.py drunet"
pt = json.load(open(os.path.join(P, "pmc_traffic.json")))
num = den = 0
for k in rd:
    if "conv3x3_wsplit_kernel" in k and k in wr:
        n = rd[k][1]
        num += (rd[k][0] * 128 + wr[k][0] * 64) * n; den += n
pt["conv3x3_wsplit_kernel"] = {"bytes_per_launch": round(num / den), "launches_averaged": den, "commit": commit,
                               "config": {"batch": 32, "height": 320, "width": 320}, "method": meth}
json.dump(pt, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(pt["conv3x3_wsplit_kernel"])
