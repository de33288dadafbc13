#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats results .db into a small text summary for profiles/.
usage: prof_summary.py <results.db> <out.md> [title]"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else db
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w") as f:
    f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats; durations in ns.\n\n")
    f.write("| kernel | calls | total_ns | avg_ns | % |\n|---|---|---|---|---|\n")
    for name, calls, tot, avg, pct in rows[:25]:
        short = name if len(name) < 110 else name[:107] + "..."
        f.write(f"| `{short}` | {calls} | {tot:.0f} | {avg:.0f} | {pct:.2f} |\n")
print(open(out).read()[:1500])
