#!/usr/bin/env python
"""Assemble the committed profiles/r03_* summaries from the outputs of scripts/gpu_evidence_r03.sh
(gpurun_out/evidence_r03) and refresh profiles/pmc_traffic.json (the PMC constants bench.py's roofline uses)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "evidence_r03")
PR = os.path.join(ROOT, "profiles")
TAG = "r03"
WORK = {"p3p_5000": ("k_score_mfma<10>", 320, 5000), "relpose_5000": ("k_score_mfma2<1, 10>", 320, 5000),
        "fund_10000": ("k_score_mfma2<2, 12>", 384, 10000), "hom_10000": ("k_score_mfmah<10>", 320, 10000)}


def rd(name):
    return open(os.path.join(EV, name)).read()


def bench(name):
    return json.loads(rd(name + ".json").strip().splitlines()[-1])


def rows(md):
    lines = [ln for ln in md.splitlines() if ln.startswith("|")]
    cols = [c.strip() for c in lines[0].strip().strip("|").split("|")]
    out = []
    for ln in lines[2:]:
        cells = [c.strip() for c in ln.strip().strip("|").split("|")]
        out.append(dict(zip(cols, cells)))
    return out


def find(rws, kernel, launches=None):
    for r in rws:
        if r["kernel"].strip("`").replace("pl::", "") == kernel and (launches is None or r.get("launches") == launches):
            return r
    return None


def detail(name):
    return json.load(open(os.path.join(EV, name + ".json")))


def main():
    d = bench("bench_default")
    s1 = bench("bench_s1")
    dd = detail("detail_default")  # {"line": the printed line, "reports": the complete per-workload reports}
    out = {}
    out[f"{TAG}_bench_line.json"] = json.dumps(d, indent=1) + "\n"
    out[f"{TAG}_bench_detail.json"] = json.dumps(dd["reports"], indent=1) + "\n"
    reports = dd["reports"]
    # ---- kernel-trace summaries ----
    p16 = rd("prof_default.md")
    r16 = find(rows(p16), "k_score_mfma_g<10>", "full batch") or find(rows(p16), "k_score_mfma_g<10>")
    out[f"{TAG}_bench_default_groups_kernel_trace.md"] = (
        f"# {TAG} — `python bench.py` (primary workload p3p_5000: {d['config']['problems_per_gpu_per_step']} problems per step through pl_ransac_batch, "
        f"lock-step groups of 16, 8 groups in flight, {d['config']['distinct_scenes']} distinct scenes) under rocprofv3 --kernel-trace --stats\n\n"
        "Command (GPU box, from /tmp): `rocprofv3 --kernel-trace --stats -d ... -- python bench.py --no-parity --no-cpu-baseline "
        "--no-secondary --steps 5` (scripts/gpu_evidence_r03.sh).  Kernels with the suffix _g are group launches (problem index = "
        "blockIdx.z): one launch serves 16 problems.\n"
        f"bench.py of the same configuration (full default run, profiles/{TAG}_bench_line.json): {d['value']:.4g} hypotheses/s; "
        f"roofline.avg_launch_ms = {d['roofline']['avg_launch_ms']:.3f} is ONE problem's share of a group's scoring launch (HIP events on "
        f"the group's stream, several groups sharing the device); rocprof average of the whole group launch below: "
        f"{float(r16['avg us']) / 1e3:.3f} ms.\n\n" + p16 +
        "\n## Device occupancy of the same configuration (scripts/busy.py on the csv kernel trace)\n\n```\n" + rd("busy_default.txt") + "```\n")
    p1 = rd("prof_s1.md")
    r1 = find(rows(p1), "k_score_mfma<10>", "full batch")
    hpl = s1["roofline"]["hypotheses_per_launch"]
    out[f"{TAG}_bench_p3p5000_1stream_kernel_trace.md"] = (
        f"# {TAG} — `python bench.py --mode streams --streams 1` under rocprofv3 --kernel-trace --stats (one problem at a time)\n\n"
        f"bench.py of the same configuration: {s1['value']:.4g} hypotheses/s = {1e3 * hpl / s1['value']:.3f} ms per 100000-iteration "
        f"problem (Python call overhead included), roofline.solo_avg_launch_ms = {s1['roofline']['solo_avg_launch_ms']:.4f}; rocprof "
        f"average of k_score_mfma<10> below: {float(r1['avg us']) / 1e3:.4f} ms ({hpl / 1e3:.1f} k hypotheses x 5000 correspondences per "
        f"launch = {hpl * 5000 / (float(r1['avg us']) * 1e-6):.3g} point-hypotheses/s).\n\n" + p1 +
        "\n## One problem in time order (scripts/timeline.py: start offset, duration, gap to the previous kernel's end)\n\n```\n" + rd("timeline_s1.txt") + "```\n")
    for w in ("relpose_5000", "fund_10000", "hom_10000"):
        b = reports[w]
        out[f"{TAG}_bench_{w}_1stream_kernel_trace.md"] = (
            f"# {TAG} — `python bench.py --workload {w} --mode streams --streams 1 --steps 3` under rocprofv3 --kernel-trace --stats\n\n"
            f"One problem at a time (every kernel with the device to itself).  bench.py default (grouped, profiles/{TAG}_bench_line.json "
            f"config.{w}_hyp_per_s; complete report: profiles/{TAG}_bench_detail.json): {b['value']:.4g} hypotheses/s, "
            f"{b['iterations_per_s']:.4g} iterations/s; dominant kernel with the device to itself {b['roofline']['solo_avg_launch_ms']:.4f} ms "
            f"per launch.\n\n" + rd(f"prof_{w}.md") +
            f"\n## The same workload as bench.py runs it by default (`--workload {w} --steps 3`: pl_ransac_batch, groups of 16)\n\n" + rd(f"profg_{w}.md"))
    bm = reports.get("batch_mixed", {})
    out[f"{TAG}_bench_batch_mixed_kernel_trace.md"] = (
        f"# {TAG} — `python bench_batch.py --problems 4096 --streams 8 --steps 2` (configs[4], grouped launches) under rocprofv3 --kernel-trace --stats\n\n"
        f"bench.py's batch_mixed leg of the default run: {bm.get('problems_per_s', 0):.5g} problems/s ({bm.get('value', 0):.4g} hypotheses/s, "
        f"2048 problems per step, 8 host threads).  Kernels with the suffix _g are the group launches (problem index = blockIdx.z); "
        "the others serve the few problems that leave the group path.\n\n" + rd("prof_batch.md") +
        "\n## Device occupancy (scripts/busy.py)\n\n```\n" + rd("busy_batch.txt") + "```\n")
    out[f"{TAG}_generator_full_device.md"] = (
        f"# {TAG} — the 5-point generator on a FULL device: `scripts/exp/genbench 1600000 16 2` (1.6 M iterations = 16 problems x 100 k, "
        "what a group launch of the grouped bench is) under rocprofv3 --kernel-trace --stats\n\n```\n" + rd("genbench.txt").strip().splitlines()[-2] + "\n```\n\n" + rd("prof_gen.md"))
    # ---- PMC ----
    traffic = {"_comment": "PMC constants of the dominant kernels (the streaming scorers), measured with rocprofv3 --pmc in separate passes "
               "(scripts/gpu_evidence_r03.sh; summaries in profiles/r03_pmc_*.md), one problem at a time. FETCH_SIZE (KB) is doubled per "
               "MI355X_MICROARCH.md (gfx950 under-reports wide streaming reads by 2x); WRITE_SIZE (KB) is taken as is. "
               "valu_insts_per_hypothesis_chunk = SQ_INSTS_VALU of one launch / (hypotheses x point chunks of the launch): bench.py's "
               "valu_issue roofline multiplies it back with the hypotheses and chunks of its own launches; it cannot re-measure it "
               "(PMC needs the profiler)."}
    for w, (kernel, chunk_pts, n) in WORK.items():
        md = rd(f"pmc_{w}.md")
        r = find(rows(md), kernel, "full batch")
        f = lambda k: float(r[k])
        # hypotheses of one launch of the PMC pass = one problem's whole run (--mode streams: one batch of 100000
        # iterations); the grouped default run cuts long runs into several batches when the arena budget asks for it
        hyp = json.loads([ln for ln in rd(f"pmc_grbm_{w}.log").splitlines() if ln.startswith("{")][-1])["roofline"]["hypotheses_per_launch"]
        chunks = (n + chunk_pts - 1) // chunk_pts
        cycles = f("GRBM_GUI_ACTIVE") / 8
        valu_busy = f("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cycles
        mfma_busy = f("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cycles if "SQ_VALU_MFMA_BUSY_CYCLES" in r and r["SQ_VALU_MFMA_BUSY_CYCLES"] else 0.0
        traffic[w] = {"kernel": kernel, "fetch_size_kb": f("FETCH_SIZE"), "write_size_kb": f("WRITE_SIZE"),
                      "traffic_bytes_per_launch": (2 * f("FETCH_SIZE") + f("WRITE_SIZE")) * 1024.0, "hypotheses_per_launch": hyp,
                      "points_per_chunk": chunk_pts, "valu_insts_per_launch": f("SQ_INSTS_VALU"),
                      "valu_insts_per_hypothesis_chunk": f("SQ_INSTS_VALU") / (hyp * chunks), "valu_busy": round(valu_busy, 3),
                      "mfma_busy": round(mfma_busy, 3), "kernel_cycles": cycles, "source": f"profiles/{TAG}_pmc_{w}.md"}
        out[f"{TAG}_pmc_{w}.md"] = (
            f"# {TAG} — PMC counters, workload {w} (`python bench.py --workload {w} --mode streams --streams 1 --steps 2 --warmup 1 --no-parity "
            "--no-cpu-baseline --no-secondary`; counters in separate rocprofv3 --pmc passes with --kernel-trace only)\n\n"
            f"Dominant kernel `{kernel}`: {hyp / 1e3:.1f} k hypotheses x {n} correspondences per launch in {chunks} chunks of {chunk_pts}; "
            f"GRBM_GUI_ACTIVE / 8 = {cycles:.4g} cycles; SQ_INSTS_VALU = {f('SQ_INSTS_VALU'):.4g} = "
            f"{f('SQ_INSTS_VALU') / (hyp * chunks):.1f} per (hypothesis, chunk); SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs = "
            f"{f('SQ_ACTIVE_INST_VALU') * 4 / 1024:.4g} cycles = {100 * valu_busy:.0f} % VALU-busy"
            + (f"; matrix pipe {100 * mfma_busy:.0f} % busy" if mfma_busy else "") +
            f"; HBM: FETCH_SIZE {f('FETCH_SIZE') / 1024:.1f} MB (x2 on gfx950) + WRITE_SIZE {f('WRITE_SIZE') / 1024:.1f} MB = "
            f"{(2 * f('FETCH_SIZE') + f('WRITE_SIZE')) / 1024:.0f} MB per launch.\n\nPer-launch averages of every kernel of the pass:\n\n" + md)
    for name, text in out.items():
        open(os.path.join(PR, name), "w").write(text)
        print("wrote", name)
    json.dump(traffic, open(os.path.join(PR, "pmc_traffic.json"), "w"), indent=1)
    for w in WORK:
        t = traffic[w]
        print(w, "valu/hyp-chunk", round(t["valu_insts_per_hypothesis_chunk"], 2), "valu_busy", t["valu_busy"], "mfma", t["mfma_busy"],
              "traffic MB", round(t["traffic_bytes_per_launch"] / 1e6, 1))


if __name__ == "__main__":
    main()
