"""merge the HBM-request counters of two rocprofv3 --pmc passes (TCC_EA0_RDREQ_sum, TCC_EA0_WRREQ_sum) over
`scripts/bench_ops.py drunet_fp32` into profiles/pmc_traffic.json: bytes per launch of every conv kernel of the fp32 setting
= RDREQ x 128 B (gfx950 tallies a 128-byte request of a wide coalesced read as one 64-byte unit: MI355X_MICROARCH.md, HBM
section) + WRREQ x 64 B, averaged over the launches of the DRUNet calls.
usage: python scripts/r04/merge_pmc.py <rdreq dir> <wrreq dir> <commit>"""
import collections, csv, glob, json, os, re, sys

def collect(d, counter):
    """total requests per kernel family and the number of CALLS: a Winograd F(4x4) convolution is one launch over the whole
    tiles (template argument SPLIT = false) plus, with a tail, one launch over the tail parts (SPLIT = true) - the bytes of
    both belong to the one call"""
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"].startswith(counter):
                m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
                if m:
                    tot[m.group(1)] += float(r["Counter_Value"])
                    if not re.search(r"wino4_kernel<[^>]*true>\(", r["Kernel_Name"]):
                        n[m.group(1)] += 1
    return tot, n

rd, nrd = collect(sys.argv[1], "TCC_EA0_RDREQ")
wr, nwr = collect(sys.argv[2], "TCC_EA0_WRREQ")
path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles", "pmc_traffic.json")
out = json.load(open(path))
for k in rd:
    if ("wino" in k or k == "conv3x3_kernel") and nrd[k] and nwr.get(k):
        key = k if "wino4" in k else k + "@fp32"
        out[key] = {"bytes_per_launch": round(rd[k] / nrd[k] * 128 + wr[k] / nwr[k] * 64), "launches_averaged": nrd[k],
                    "read_requests_per_launch": round(rd[k] / nrd[k]), "write_requests_per_launch": round(wr[k] / nwr[k]),
                    "commit": sys.argv[3], "config": {"batch": 32, "height": 320, "width": 320},
                    "method": "rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum (separate passes); bytes = RDREQ x 128 + WRREQ x 64 on scripts/bench_ops.py drunet_fp32"}
        print(key, out[key]["bytes_per_launch"])
json.dump(out, open(path, "w"), indent=1)
