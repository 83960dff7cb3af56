"""per-kernel averages of rocprofv3 --pmc passes (csv): python scripts/r05/pmc_table.py <dir prefix> [kernel regex]
one line per kernel (first launch config seen per name x grid), counters of all passes <prefix>_* merged"""
import collections, csv, glob, re, sys
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
agg = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + "*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if pat and not pat.search(n):
            continue
        short = re.sub(r"\(anonymous namespace\)::|dinv::|HIP_vector_type<float, 2u>|void ", "", n)
        short = re.sub(r"\(.*$", "", short)[:90]
        key = (short, r.get("Grid_Size", ""))
        d = agg.setdefault(key, collections.defaultdict(list))
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in agg.items():
    print(k, "grid", g)
    print("   ", {c: round(sum(v) / len(v)) for c, v in d.items()})
