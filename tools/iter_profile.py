#!/usr/bin/env python3
"""Per-iteration kernel table from a rocprofv3 --kernel-trace database (rocpd sqlite): python tools/iter_profile.py <results.db> <marker kernel> [n_iterations]
The iterations are delimited by launches of the marker kernel (k_gather_rays: one ray-batch gather per training iteration)."""
import sqlite3, collections, sys
db, marker = sys.argv[1], sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 300
rows = list(sqlite3.connect(db).cursor().execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if marker in r[0]]
seg = rows[idx[-N - 1]:idx[-1]]
wall = (seg[-1][2] - seg[0][1]) / N
busy = sum(e - s for _, s, e in seg) / N
print(f"last {N} iterations: {wall / 1e3:.1f} us per iteration, GPU busy {busy / 1e3:.1f} us, {len(seg) / N:.1f} launches\n")
print("| kernel | launches / iteration | us / iteration |\n|---|---|---|")
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in seg:
    k = n.split("(")[0].replace("void ", "")[:100]
    agg[k][0] += e - s; agg[k][1] += 1
rest = 0.0
for i, (k, (t, c)) in enumerate(sorted(agg.items(), key=lambda x: -x[1][0])):
    if i < 14: print(f"| `{k}` | {c / N:.2f} | {t / N / 1e3:.1f} |")
    else: rest += t / N / 1e3
print(f"| (all others) | | {rest:.1f} |")
