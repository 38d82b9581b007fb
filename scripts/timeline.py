#!/usr/bin/env python
"""Print one step of a rocprofv3 kernel trace (csv) in time order: start offset, duration, gap to the previous
kernel's end, kernel name, grid.  Usage: python scripts/timeline.py <kernel_trace.csv> [first_kernel_substr] [nth]"""
import csv
import sys


def main(path, anchor="k_sample_delta", nth=3):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    if len(starts) <= nth + 1:
        nth = max(0, len(starts) - 2)
    a, b = starts[nth], starts[nth + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = t0
    print(f"step = kernels {a}..{b - 1}, wall {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name:48s} grid {r.get('Grid_Size', r.get('Grid_Size_X', '?'))}")
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), *(int(x) for x in sys.argv[3:4]))
