#!/usr/bin/env python
"""GPU occupancy of a rocprofv3 kernel trace (csv): wall time, union of kernel intervals (GPU not idle), and per
kernel the summed duration plus the time during which it was the ONLY kernel running.  Used to judge how well the
multi-stream mode fills the device.  Usage: python scripts/busy.py <kernel_trace.csv> [skip_first_fraction]"""
import csv
import sys
from collections import defaultdict


def main(path, skip=0.3):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
            for r in csv.DictReader(open(path))]
    rows.sort()
    t_begin, t_end = rows[0][0], max(r[1] for r in rows)
    cut = t_begin + skip * (t_end - t_begin)
    rows = [r for r in rows if r[0] >= cut]
    t_begin = rows[0][0]
    ev = []
    for s, e, n in rows:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort()
    active = defaultdict(int)
    nact = 0
    last = t_begin
    busy = 0
    solo = defaultdict(int)
    conc_hist = defaultdict(int)
    for t, d, n in ev:
        if nact > 0:
            busy += t - last
            conc_hist[min(nact, 9)] += t - last
            if nact == 1:
                solo[[k for k, v in active.items() if v > 0][0]] += t - last
        last = t
        active[n] += d
        nact += d
    wall = t_end - t_begin
    tot = defaultdict(int)
    cnt = defaultdict(int)
    for s, e, n in rows:
        tot[n] += e - s
        cnt[n] += 1
    print(f"wall {wall / 1e6:.3f} ms, GPU non-idle {busy / 1e6:.3f} ms ({100.0 * busy / wall:.1f} %)")
    print("time with k kernels in flight: " + ", ".join(f"{k}: {100.0 * v / wall:.1f} %" for k, v in sorted(conc_hist.items())))
    print("| kernel | calls | summed us | avg us | solo us |")
    print("|---|---|---|---|---|")
    for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
        print(f"| `{n[:44]}` | {cnt[n]} | {v / 1e3:.1f} | {v / 1e3 / cnt[n]:.1f} | {solo[n] / 1e3:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], *(float(x) for x in sys.argv[2:3]))
