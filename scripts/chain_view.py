#!/usr/bin/env python
"""One group's launch chain out of a rocprofv3 kernel trace of pl_estimate_batch: python scripts/chain_view.py trace.csv [stream_id] [nth k_prepare_g on it]
Prints the kernels of ONE stream between two stage-A launches (k_prepare_g with db stage) in time order with gaps; and a per-stream
summary (busy share of the steady-state window)."""
import csv, sys, collections
path = sys.argv[1]
rows = [r for r in csv.DictReader(open(path))]
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"]); r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pl::", "")
rows.sort(key=lambda r: r["s"])
t0, t1 = rows[0]["s"], max(r["e"] for r in rows)
cut = t0 + 0.5 * (t1 - t0)
by = collections.defaultdict(list)
for r in rows:
    if r["s"] >= cut:
        by[r["Stream_Id"]].append(r)
print("| stream | queue(s) | kernels | busy ms | busy % of window | gaps > 50 us: count, total ms |")
for s, v in sorted(by.items(), key=lambda kv: int(kv[0])):
    busy = sum(r["e"] - r["s"] for r in v)
    gaps = [b["s"] - a["e"] for a, b in zip(v, v[1:]) if b["s"] - a["e"] > 50000]
    print(f"| {s} | {sorted(set(r['Queue_Id'] for r in v))} | {len(v)} | {busy/1e6:.1f} | {100*busy/(t1-cut):.0f} | {len(gaps)}, {sum(gaps)/1e6:.1f} |")
sid = sys.argv[2] if len(sys.argv) > 2 else max(by, key=lambda k: len(by[k]))
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 0
v = by[sid]
# a group starts with k_prepare_g NOT preceded (within the chain) by a mask kernel of the same group... simply: k_prepare_g following a k_lm / k_mask
starts = [i for i, r in enumerate(v) if r["n"].startswith("k_prepare_g") and (i == 0 or not v[i-1]["n"].startswith("k_mask"))]
starts = [i for i, r in enumerate(v) if r["n"].startswith("k_prepare_g") and (i == 0 or v[i-1]["n"].startswith("k_lm"))] or starts
if len(starts) > nth + 1:
    a, b = starts[nth], starts[nth + 1]
else:
    a, b = 0, len(v)
print(f"\nstream {sid}: kernels {a}..{b-1}, wall {(v[b-1]['e'] - v[a]['s'])/1e3:.0f} us, kernel time {sum(r['e']-r['s'] for r in v[a:b])/1e3:.0f} us")
prev = v[a]["s"]
for r in v[a:b]:
    print(f"{(r['s']-v[a]['s'])/1e3:9.1f} us  dur {(r['e']-r['s'])/1e3:8.1f}  gap {(r['s']-prev)/1e3:7.1f}  {r['n'][:40]:40s} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}")
    prev = r["e"]
