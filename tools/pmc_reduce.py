#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (counter_collection / kernel_stats) to small per-kernel summaries.
usage: pmc_reduce.py <dir> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
res = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            if not (k.startswith("void k_") or k.startswith("k_")):
                continue
            a = agg[k][row.get("Counter_Name", "?")]
            a[0] += float(row.get("Counter_Value", 0)); a[1] += 1
    # a kernel dispatch contributes one row per counter (already summed over XCDs/dims by rocprofv3 when aggregated)
    res["counters"] = {k: {c: {"sum": v[0], "rows": v[1]} for c, v in cs.items()} for k, cs in agg.items()}
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        res["kernel_stats"] = [row for row in csv.DictReader(fh)][:20]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:3000])
