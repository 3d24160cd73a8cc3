"""Per-kernel totals of a rocprofv3 rocpd database (the default output of ROCm 7.2 without --output-format csv):
usage: db_kernels.py results.db [skip_first_n_dispatches_per_kernel]"""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, grid_x, grid_y from kernels order by start"))
d = collections.defaultdict(list)
for n, s, e, gx, gy in rows:
    d[n.split("(")[0][:70]].append(e - s)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v) / 1e3:12.1f} us {100 * sum(v) / tot:5.1f}%  n={len(v):5d}  avg {sum(v) / len(v) / 1e3:9.1f} us  {k}")
print(f"{tot / 1e3:12.1f} us total")
