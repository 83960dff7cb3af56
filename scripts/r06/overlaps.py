"""for the kernels whose name matches `pattern` in a rocprofv3 rocpd database: duration, and which OTHER kernels ran during their interval
(name -> overlapped microseconds), to tell a slow kernel from a kernel that shares the device.   python scripts/r06/overlaps.py file.db pattern"""
import collections, re, sqlite3, sys

c = sqlite3.connect(sys.argv[1])
pat = re.compile(sys.argv[2])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r"<.*$|\(.*$", "", re.sub(r"\(anonymous namespace\)::|dinv::|void ", "", n))[:40]
starts = [r[1] for r in rows]
import bisect
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for idx, (n, s, e) in enumerate(rows):
    if not pat.search(n):
        continue
    k = short(n)
    a = agg[k]
    a[0] += 1
    a[1] += (e - s) / 1e3
    lo = max(0, bisect.bisect_left(starts, s) - 64)
    for m, s2, e2 in rows[lo: bisect.bisect_right(starts, e) + 1]:
        if (s2, e2) == (s, e) and m == n:
            continue
        ov = min(e, e2) - max(s, s2)
        if ov > 0:
            a[2][short(m)] += ov / 1e3
for k, (cnt, tot, ov) in agg.items():
    print(f"{k:42s} n={cnt:5d} mean={tot / cnt:8.1f} us   overlapped by: " + ", ".join(f"{m} {v / cnt:.1f} us" for m, v in ov.most_common(4)))
