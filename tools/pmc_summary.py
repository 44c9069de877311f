#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from a *_counter_collection.csv."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if pat in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-28s n=%-5d avg=%.0f" % (c, len(vals), sum(vals) / len(vals)))
