"""per-kernel durations from a rocprofv3 rocpd database (rocprofv3 --kernel-trace writes <name>_results.db on this image):
name (shortened), workgroups, registers, launches, median / min microseconds.   python scripts/r05/kstats.py file.db [regex]"""
import collections, re, sqlite3, sys

c = sqlite3.connect(sys.argv[1])
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
rows = c.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, scratch_size, lds_size from kernels order by start").fetchall()
agg = collections.OrderedDict()
for n, s, e, g, w, vg, ag, sc, lds in rows:
    if pat and not pat.search(n):
        continue
    short = re.sub(r"\(anonymous namespace\)::|dinv::|HIP_vector_type<float, 2u>|void ", "", n)
    short = re.sub(r"\(.*$", "", short)
    agg.setdefault((short[:100], g // max(w, 1), vg + ag, sc, lds), []).append((e - s) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(f"{k[0]:102s} wgs={k[1]:6d} regs={k[2]:4d} scr={k[3]:4d} lds={k[4]:6d} n={len(v):4d} med={v[len(v) // 2]:8.1f} min={v[0]:8.1f}")
