"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes over `scripts/bench_ops.py drunet` (one DRUNet call at B=32,
320x320): HBM bytes per launch of each conv kernel = TCC_EA0_RDREQ x 64 B x 2 (gfx950 counts a 128-B request of a wide
coalesced read as one 64-B unit: MI355X_MICROARCH.md, HBM section) + TCC_EA0_WRREQ x 64 B, averaged over its launches.
usage: python scripts/make_pmc_traffic.py <rdreq dir> <wrreq dir> <commit> > profiles/pmc_traffic.json"""
import collections, csv, glob, json, re, sys

def collect(d, counter):
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"].startswith(counter):
                m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
                if not m:
                    continue
                k = m.group(1)
                tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n

rd, nrd = collect(sys.argv[1], "TCC_EA0_RDREQ")
wr, nwr = collect(sys.argv[2], "TCC_EA0_WRREQ")
out = {}
for k in rd:
    if "conv3x3" in k and nrd[k] and nwr.get(k):
        out[k] = {"bytes_per_launch": round(rd[k] / nrd[k] * 64 * 2 + wr[k] / nwr[k] * 64), "launches_averaged": nrd[k],
                  "read_requests_per_launch": round(rd[k] / nrd[k]), "write_requests_per_launch": round(wr[k] / nwr[k]),
                  "commit": sys.argv[3], "config": {"batch": 32, "height": 320, "width": 320},
                  "method": "rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum (separate passes) on scripts/bench_ops.py drunet"}
print(json.dumps(out, indent=1))
