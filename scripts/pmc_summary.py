#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (one counter_collection.csv per pass) per kernel.
    python scripts/pmc_summary.py gpurun_out/pmc1/p_counter_collection.csv [more csv ...]
Full-batch launches (grid >= 100000 threads) of the scoring kernels are reported separately."""
import collections
import csv
import sys


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        for x in csv.DictReader(open(path)):
            name = x["Kernel_Name"].split("(")[0].replace("void ", "")
            big = int(x["Grid_Size"]) >= 100000
            agg[(name, "full batch" if big else "small")][x["Counter_Name"]].append(float(x["Counter_Value"]))
    counters = sorted({c for d in agg.values() for c in d})
    print("| kernel | launches | n | " + " | ".join(counters) + " |")
    print("|---|---|---|" + "---|" * len(counters))
    for (name, tag), d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        n = max(len(v) for v in d.values())
        print(f"| `{name}` | {tag} | {n} | " + " | ".join(f"{sum(d[c])/len(d[c]):.4g}" if c in d else "" for c in counters) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
