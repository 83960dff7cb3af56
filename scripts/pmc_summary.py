"""average the counters of a rocprofv3 --pmc run per kernel: python scripts/pmc_summary.py <dir> [kernel substring]"""
import collections, csv, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if pat in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    for k, v in agg.items():
        print(k, {c: round(x / max(cnt[k][c], 1)) for c, x in v.items()}, "launches", max(cnt[k].values()))
