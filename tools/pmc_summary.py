#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from a *_counter_collection.csv.
With the TCC_EA0 request counters present it also prints HBM-side bytes per launch:
  read  = (RDREQ - RDREQ_32B) * 64 B + RDREQ_32B * 32 B     (x2 for 16 B/lane streaming reads on gfx950, see
          /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]": 128-B requests are tallied at 64 B)
  write = WRREQ_64B * 64 B + (WRREQ - WRREQ_64B) * 32 B     (uncalibrated)
usage: pmc_summary.py counter_collection.csv [kernel-substring]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if pat in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    avg = {}
    for c, vals in sorted(v.items()):
        avg[c] = sum(vals) / len(vals)
        print("   %-32s n=%-5d avg=%.0f" % (c, len(vals), avg[c]))
    if "TCC_EA0_RDREQ_sum" in avg:
        rd = (avg["TCC_EA0_RDREQ_sum"] - avg.get("TCC_EA0_RDREQ_32B_sum", 0)) * 64 + avg.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
        print("   => read  bytes/launch (raw)      %.0f   (x2 if 16 B/lane streaming: %.0f)" % (rd, 2 * rd))
    if "TCC_EA0_WRREQ_sum" in avg:
        wr = avg.get("TCC_EA0_WRREQ_64B_sum", 0) * 64 + (avg["TCC_EA0_WRREQ_sum"] - avg.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32
        print("   => write bytes/launch            %.0f" % wr)
