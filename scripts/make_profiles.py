#!/usr/bin/env python
"""Assemble the committed profiles/ summaries of a round from the outputs of scripts/gpu_evidence.sh
(gpurun_out/evidence): usage  python scripts/make_profiles.py r01_g"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "evidence")
PR = os.path.join(ROOT, "profiles")


def bench(name):
    return json.loads(open(os.path.join(EV, name + ".json")).read().strip().splitlines()[-1])


def row_of(md, kernel):
    for ln in md.splitlines():
        if ln.startswith("| `pl::" + kernel):
            return [c.strip() for c in ln.strip().strip("|").split("|")]
    return None


def main(tag):
    out = {}
    d16, d1 = bench("bench_default"), bench("bench_s1")
    prof16 = open(os.path.join(EV, "prof_default.md")).read()
    busy = open(os.path.join(EV, "busy_default.txt")).read()
    r16 = row_of(prof16, "k_score_mfma")
    out[f"{tag}_bench_default_16streams_kernel_trace.md"] = (
        f"# {tag} — `python bench.py` under rocprofv3 --kernel-trace --stats (16 problems in flight, 128 per step, MI355X)\n\n"
        "Command (GPU box, from /tmp): `rocprofv3 --kernel-trace --stats -d ... -- python bench.py --no-cpu-baseline` (scripts/gpu_evidence.sh).\n"
        f"bench.py of the same configuration: {d16['value']:.3g} hypotheses/s, roofline.avg_launch_ms = {d16['roofline']['avg_launch_ms']:.3f} "
        f"(HIP events, 16 launches sharing the device); rocprof average of the same kernel below: {float(r16[4]) / 1e3:.3f} ms.\n\n"
        + prof16 + "\n## Device occupancy of the same configuration (scripts/busy.py on the csv kernel trace)\n\n```\n" + busy + "```\n")
    p1 = open(os.path.join(EV, "prof_s1.md")).read()
    r1 = row_of(p1, "k_score_mfma")
    hyp_per_launch = d1["roofline"]["algorithmic_bytes_per_launch"] / (d1["config"]["correspondences"] * 40)
    out[f"{tag}_bench_p3p5000_1stream_kernel_trace.md"] = (
        f"# {tag} — `python bench.py --streams 1` under rocprofv3 --kernel-trace --stats (one problem at a time)\n\n"
        f"bench.py of the same configuration: {d1['value']:.3g} hypotheses/s, "
        f"{1e3 * hyp_per_launch / d1['value']:.2f} ms per 100000-iteration problem, roofline.avg_launch_ms = "
        f"{d1['roofline']['avg_launch_ms']:.3f}; rocprof average of k_score_mfma<10> below: {float(r1[4]) / 1e3:.3f} ms "
        f"({hyp_per_launch / 1e3:.1f} k hypotheses x 5000 correspondences per launch = "
        f"{hyp_per_launch * 5000 / (float(r1[4]) * 1e-6):.2g} point-hypotheses/s).\n\n" + p1)
    for w in ("relpose_5000", "fund_10000", "hom_10000"):
        b = bench("bench_" + w)
        out[f"{tag}_bench_{w}_1stream_kernel_trace.md"] = (
            f"# {tag} — `python bench.py --workload {w} --streams 1 --steps 3` under rocprofv3 --kernel-trace --stats\n\n"
            f"bench.py --workload {w} (16 problems in flight): {b['value']:.3g} hypotheses/s, {b['config']['iterations_per_s']:.3g} iterations/s.\n\n"
            + open(os.path.join(EV, f"prof_{w}.md")).read())
    # PMC
    pmc = open(os.path.join(EV, "pmc_p3p.md")).read()
    head = pmc.splitlines()[:2]
    cols = [c.strip() for c in head[0].strip().strip("|").split("|")]
    r = dict(zip(cols, row_of(pmc, "k_score_mfma")))
    f = lambda k: float(r[k])
    cycles = f("GRBM_GUI_ACTIVE") / 8
    launch_ms = d1["roofline"]["avg_launch_ms"]
    ghz = cycles / (launch_ms * 1e-3) / 1e9
    valu_busy = f("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cycles
    mfma_busy = f("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cycles
    waves_per_hyp = 16  # chunks of 320 correspondences at N = 5000
    valu_per = f("SQ_INSTS_VALU") / (hyp_per_launch * waves_per_hyp)
    fetch_kb, write_kb = f("FETCH_SIZE"), f("WRITE_SIZE")
    traffic = (2 * fetch_kb + write_kb) * 1024
    pf = open(os.path.join(EV, "pmc_fund.md")).read()
    colsf = [c.strip() for c in pf.splitlines()[0].strip().strip("|").split("|")]
    rf = dict(zip(colsf, row_of(pf, "k_score_queue")))
    bf = bench("bench_fund_10000")
    hyp_f = bf["roofline"]["algorithmic_bytes_per_launch"] / (10000 * 32)
    traffic_f = (2 * float(rf["FETCH_SIZE"]) + float(rf["WRITE_SIZE"])) * 1024
    cmds = [ln for ln in open(os.path.join(ROOT, "scripts", "gpu_evidence.sh")).read().splitlines() if "--pmc" in ln]
    pmc_name = f"{tag}_pmc_k_score_mfma.md"
    out[pmc_name] = (
        f"# {tag} — PMC counters of the streaming scorers\n\nCommands (MI355X box, from /tmp; counters in separate passes with "
        "`--kernel-trace --output-format csv` only — scripts/gpu_evidence.sh):\n```\n" + "\n".join(cmds) + "\n```\n"
        f"Per launch averages, k_score_mfma<10> (P3P, config 1: {hyp_per_launch / 1e3:.1f} k hypotheses x 5000 correspondences; SQ_* in "
        "quad-cycles where they count cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs):\n\n" + "\n".join(head) + "\n"
        + "| " + " | ".join(r[c] for c in cols) + " |\n\n"
        f"Reading: {launch_ms:.3f} ms per launch at {ghz:.2f} GHz (GRBM_GUI_ACTIVE / 8 = {cycles:.3g} cycles); SQ_INSTS_VALU = "
        f"{valu_per:.1f} per (hypothesis, wavefront of 320 points); SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs = "
        f"{f('SQ_ACTIVE_INST_VALU') * 4 / 1024:.3g} cycles = {100 * valu_busy:.0f} % VALU-busy; SQ_VALU_MFMA_BUSY_CYCLES / 1024 = "
        f"{f('SQ_VALU_MFMA_BUSY_CYCLES') / 1024:.3g} cycles = {100 * mfma_busy:.0f} % matrix-pipe busy (32 cycles per 32x32x8 f16 tile); HBM "
        f"traffic FETCH_SIZE {fetch_kb / 1e3:.1f} MB (x2 on gfx950 = {2 * fetch_kb / 1e3:.1f} MB) + WRITE_SIZE {write_kb / 1e3:.1f} MB = "
        f"{traffic / 1e6:.0f} MB vs {d1['roofline']['algorithmic_bytes_per_launch'] / 1e9:.1f} GB algorithmic.\n\n"
        f"k_score_queue<2, 6> (7-point F, config 3: {hyp_f / 1e3:.1f} k hypotheses x 10000 correspondences): FETCH_SIZE "
        f"{float(rf['FETCH_SIZE']) / 1e3:.1f} MB (x2), WRITE_SIZE {float(rf['WRITE_SIZE']) / 1e3:.1f} MB => {traffic_f / 1e6:.0f} MB per launch.\n\n"
        "All kernels of the P3P pass:\n\n" + pmc)
    tj = {
        "_comment": "HBM bytes per launch of the dominant kernel (the streaming scorers k_score_mfma / k_score_queue), measured with "
                    "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/gpu_evidence.sh; summary in profiles/" + pmc_name +
                    "). FETCH_SIZE (KB) is doubled per MI355X_MICROARCH.md (gfx950 under-reports wide streaming reads by 2x); WRITE_SIZE (KB) "
                    "is taken as is (it matches hypotheses x chunks x 12 B of partials). bench.py quotes the entry of its workload in "
                    "roofline.traffic; it cannot re-measure it (PMC needs the profiler).",
        "p3p_5000": {"kernel": "k_score_mfma<10>", "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
                     "traffic_bytes_per_launch": traffic, "hypotheses_per_launch": round(hyp_per_launch),
                     "valu_busy": round(valu_busy, 2), "mfma_busy": round(mfma_busy, 2),
                     "valu_insts_per_wave_hypothesis": round(valu_per, 1), "effective_clock_ghz": round(ghz, 2),
                     "source": "profiles/" + pmc_name},
        "fund_10000": {"kernel": "k_score_queue<2, 6>", "fetch_size_kb": float(rf["FETCH_SIZE"]),
                       "write_size_kb": float(rf["WRITE_SIZE"]), "traffic_bytes_per_launch": traffic_f,
                       "hypotheses_per_launch": round(hyp_f), "source": "profiles/" + pmc_name},
    }
    for name, text in out.items():
        open(os.path.join(PR, name), "w").write(text)
        print("wrote", name)
    json.dump(tj, open(os.path.join(PR, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in tj["p3p_5000"].items()}, indent=0))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01_g")
