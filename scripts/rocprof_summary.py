#!/usr/bin/env python
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the markdown summary committed under profiles/.

    python scripts/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_....md
Per kernel: calls, total / average / min / max duration (us), share of GPU time.  Kernels that are launched
both on the full hypothesis batch and on a handful of refined models (k_score) are additionally split by
grid size so that the dominant launches can be compared with bench.py's HIP-event timing.
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select * from kernels").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ix else [c for c in cols if "name" in c][0]
    groups = {}
    for r in rows:
        name = r[ix[name_c]]
        dur = (r[ix["end"]] - r[ix["start"]]) / 1e3
        gx = r[ix["grid_x"]] if "grid_x" in ix else r[ix.get("grid_size", 0)]
        wx = r[ix["workgroup_x"]] if "workgroup_x" in ix else 1
        gy = r[ix["grid_y"]] if "grid_y" in ix else 1
        wy = r[ix["workgroup_y"]] if "workgroup_y" in ix else 1
        blocks = (gx // max(wx, 1)) * (gy // max(wy, 1))
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        key = (short, "full batch" if blocks >= 512 else "small") if "k_score" in short or "k_lm" in short else (short, "")
        groups.setdefault(key, []).append((dur, blocks, r[ix["vgpr_count"]] if "vgpr_count" in ix else None,
                                           r[ix["sgpr_count"]] if "sgpr_count" in ix else None,
                                           r[ix["lds_size"]] if "lds_size" in ix else None))
    total = sum(d for v in groups.values() for d, *_ in v)
    print("| kernel | launches | calls | total us | avg us | min us | max us | % GPU time | blocks (max) | VGPR | SGPR | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for (short, tag), v in sorted(groups.items(), key=lambda kv: -sum(d for d, *_ in kv[1])):
        ds = [d for d, *_ in v]
        print(f"| `{short}` | {tag} | {len(ds)} | {sum(ds):.1f} | {sum(ds)/len(ds):.2f} | {min(ds):.2f} | {max(ds):.2f} | "
              f"{100*sum(ds)/total:.1f} | {max(b for _, b, *_ in v)} | {v[0][2]} | {v[0][3]} | {v[0][4]} |")
    print(f"\ntotal kernel time: {total:.1f} us over {sum(len(v) for v in groups.values())} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
