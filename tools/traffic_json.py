#!/usr/bin/env python3
"""HBM-side traffic of bench.py's kernels from a rocprofv3 --pmc TCC_EA0_* counter_collection.csv -> JSON for bench.py.
  read  = (RDREQ - RDREQ_32B) * 64 B + RDREQ_32B * 32 B, x2 (gfx950: 128-B requests of 16 B/lane streaming reads are tallied at 64 B,
          /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]");  write = WRREQ_64B * 64 B + (WRREQ - WRREQ_64B) * 32 B
`gemm_bytes_per_launch` = the k_gemm_nn_plain average; `step_bytes` = sum over the kernels of the CNN step (every kernel launched about as
often as cs_fwd, i.e. once per step) of their per-launch averages.
usage: traffic_json.py counter_collection.csv out.json [source-note]"""
import collections, csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
per = {}
for k, v in agg.items():
    a = {c: sum(x) / len(x) for c, x in v.items()}
    n = max(len(x) for x in v.values())
    if "TCC_EA0_RDREQ_sum" not in a:
        continue
    rd = 2 * ((a["TCC_EA0_RDREQ_sum"] - a.get("TCC_EA0_RDREQ_32B_sum", 0)) * 64 + a.get("TCC_EA0_RDREQ_32B_sum", 0) * 32)
    wr = a.get("TCC_EA0_WRREQ_64B_sum", 0) * 64 + (a.get("TCC_EA0_WRREQ_sum", 0) - a.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32
    per[k] = {"launches": n, "read": round(rd), "write": round(wr)}
steps = max((p["launches"] for k, p in per.items() if k.startswith("cs_fwd")), default=0)
gemm = [p for k, p in per.items() if "k_gemm_nn_plain" in k]
step_k = {k: p for k, p in per.items() if steps and 0.9 * steps <= p["launches"] <= 1.1 * steps and "k_gemm_nn_plain" not in k}
out = {"source": sys.argv[3] if len(sys.argv) > 3 else sys.argv[1],
       "gemm_bytes_per_launch": (gemm[0]["read"] + gemm[0]["write"]) if gemm else None,
       "step_bytes": sum(p["read"] + p["write"] for p in step_k.values()) if step_k else None,
       "step_kernels": {k[:80]: p for k, p in step_k.items()}, "steps_counted": steps}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("gemm_bytes_per_launch", "step_bytes", "steps_counted")}))
