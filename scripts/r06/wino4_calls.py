"""From a rocprofv3 rocpd database of bench.py: the F(4x4) kernel's calls (a call = its whole-tile launch + the tail-split launch where one
exists), their mean duration - what bench.py's `roofline.avg_launch_ms` measures with HIP events - and, because the batch lanes run two
launch sequences concurrently, the time the kernel family OCCUPIES the device (union of the launches' intervals) per call.
    python scripts/r06/wino4_calls.py file.db"""
import re, sqlite3, sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels where name like '%conv3x3_wino4_kernel%' order by start").fetchall()
split = re.compile(r"wino4_kernel<[^>]*, true, (false|true)>")      # template argument SPLIT = true: the tail-split launch of a call
calls = sum(0 if split.search(n) else 1 for n, _, _ in rows)
total = sum(e - s for _, s, e in rows) / 1e6
union, cur_s, cur_e = 0.0, None, None
for _, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    union += cur_e - cur_s
union /= 1e6
big = [e - s for n, s, e in rows if not split.search(n) and e - s >= 0.3e6]
print(f"# conv3x3_wino4_kernel, launches of at least 0.3 ms (the 16-slice lanes of the bench's 32-slice batch; the short ones are the 4-slice "
      f"layer-error and 8-slice parity runs of the same process): {len(big)} calls, mean {sum(big) / max(len(big), 1) / 1e6:.4f} ms")
print(f"# conv3x3_wino4_kernel: {len(rows)} launches = {calls} calls (tail-split launches counted into their call); sum of durations {total:.1f} ms "
      f"= mean {total / max(calls, 1):.4f} ms per call (to compare with roofline.avg_launch_ms); the launches of the two batch lanes overlap: "
      f"the kernel occupies the device for {union:.1f} ms = {union / max(calls, 1):.4f} ms per call (concurrency {total / max(union, 1e-9):.2f})")
