"""timeline of a rocprofv3 kernel trace (rocpd database): per kernel name total busy time, and the IDLE time between consecutive
kernels on the device (end of one to start of the next) - what a HIP graph / a fused launch can and cannot remove.
    python scripts/r06/gaps.py file.db [skip_first_n_kernels]"""
import collections, re, sqlite3, sys

c = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = c.execute("select name, start, end from kernels order by start").fetchall()[skip:]
busy = collections.Counter()
cnt = collections.Counter()
gap_after = collections.defaultdict(list)
tot_gap = 0.0
for i, (n, s, e) in enumerate(rows):
    short = re.sub(r"\(anonymous namespace\)::|dinv::|void ", "", n)
    short = re.sub(r"<.*$|\(.*$", "", short)[:60]
    busy[short] += (e - s) / 1e3
    cnt[short] += 1
    if i + 1 < len(rows):
        g = (rows[i + 1][1] - e) / 1e3
        if g < 200:          # larger gaps are host-side pauses between steps, not launch gaps
            gap_after[short].append(g)
            tot_gap += max(g, 0.0)
span = (rows[-1][2] - rows[0][1]) / 1e3
print(f"kernels {len(rows)}  span {span / 1e3:.2f} ms  busy {sum(busy.values()) / 1e3:.2f} ms  small gaps {tot_gap / 1e3:.2f} ms")
for k, v in busy.most_common(25):
    g = sorted(gap_after.get(k, [0.0]))
    print(f"{k:62s} n={cnt[k]:6d} busy={v / 1e3:9.2f} ms  mean={v / cnt[k]:8.1f} us   gap after: med={g[len(g) // 2]:6.1f} mean={sum(g) / len(g):6.1f} us")
