#!/usr/bin/env python
"""Assemble the committed profiles/r04_* files from gpurun_out/evidence_r04 (scripts/gpu_evidence_r04.sh a / b / c)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "evidence_r04")
PR = os.path.join(ROOT, "profiles")


def rd(name):
    p = os.path.join(EV, name)
    return open(p).read() if os.path.exists(p) else ""


def wr(name, text):
    with open(os.path.join(PR, name), "w") as f:
        f.write(text if text.endswith("\n") else text + "\n")


def main():
    line = json.loads(rd("bench_default.json").strip().splitlines()[-1])
    wr("r04_bench_line.json", json.dumps(line, indent=1))
    det = json.loads(rd("detail_default.json"))
    wr("r04_bench_detail.json", json.dumps(det["reports"], indent=1))
    wr("r04_pytest_gpu.log", rd("pytest_gpu.log"))
    c = line["config"]
    s1 = json.loads(rd("bench_s1.json").strip().splitlines()[-1])
    wr("r04_bench_default_groups_kernel_trace.md",
       f"# r04 - `python bench.py` (primary workload p3p_5000: {c['problems_per_gpu_per_step']} problems per step through pl_ransac_batch, lock-step groups of 16, 8 groups in "
       f"flight, {c['distinct_scenes']} distinct scenes) under rocprofv3 --kernel-trace --stats\n\nCommand (GPU box, from /tmp): `rocprofv3 --kernel-trace --stats -- python bench.py "
       "--no-parity --no-cpu-baseline --no-secondary --steps 5` (scripts/gpu_evidence_r04.sh b).  The bench line of the same build (scripts/gpu_evidence_r04.sh a, "
       f"`--steps 20 --warmup 5`): {line['value']:.4g} hypotheses/s, {line['ms_per_step']:.1f} ms per step, roofline frac {line['roofline']['frac']:.3f} "
       f"(peak priced at the kernel's instruction mix: {line['roofline']['issue_cycles_per_instruction']:.2f} issue cycles per VALU instruction; all-half-rate reading "
       f"{line['roofline']['frac_if_all_half_rate']:.3f}).\n\n" + rd("prof_default.md") + "\n## Device occupancy (scripts/busy.py)\n\n```\n" + rd("busy_default.txt") + "```\n")
    wr("r04_bench_p3p5000_1stream_kernel_trace.md",
       f"# r04 - one problem at a time (`bench.py --mode streams --streams 1`): {s1['ms_per_step'] / s1['config']['problems_per_gpu_per_step']:.3f} ms per 100 k-iteration P3P problem\n\n" + rd("prof_s1.md"))
    for w in ("relpose_5000", "fund_10000", "hom_10000"):
        wr(f"r04_bench_{w}_1stream_kernel_trace.md", f"# r04 - `bench.py --workload {w} --mode streams --streams 1` under rocprofv3 --kernel-trace --stats\n\n" + rd(f"prof_{w}.md"))
        wr(f"r04_bench_{w}_groups_kernel_trace.md", f"# r04 - `bench.py --workload {w}` (grouped) under rocprofv3 --kernel-trace --stats\n\n" + rd(f"profg_{w}.md"))
    for w in ("p3p_5000", "relpose_5000", "fund_10000", "hom_10000"):
        wr(f"r04_pmc_{w}.md", f"# r04 - PMC passes of `bench.py --workload {w} --mode streams --streams 1` (separate rocprofv3 --pmc runs: SQ set 1, SQ set 2, FETCH_SIZE, WRITE_SIZE; "
                              "FETCH_SIZE in KB, doubled in profiles/pmc_traffic.json per MI355X_MICROARCH.md)\n\nRound 3 for comparison: profiles/r03_pmc_*.md (k_score_mfma<10>: "
                              "SQ_INSTS_VALU 6.868e7 per launch before the one-survivor-per-lane expansion path).\n\n" + rd(f"pmc_{w}.md"))
    wr("r04_bench_batch_mixed_kernel_trace.md",
       "# r04 - configs[4]: `pl_estimate_batch`, 4096 mixed default-option problems per call, steady state\n\n"
       f"Driver's leg in the bench line of this build: **{c['batch_mixed_problems_per_s']:.0f} problems/s** ({c['batch_mixed_hyp_per_s']:.3g} hypotheses/s, parity "
       f"{c['batch_mixed_parity_ok']}).\n\nTrace: `rocprofv3 --kernel-trace -- python scripts/batch_sweep.py 4096 10:0:3` = 3 warm-up + 4 timed calls back to back, nothing else on the "
       "device; the figures below are over the LAST HALF of the kernel span (the timed calls).  Round 3's \"52 % non-idle, 14 817 copyBuffer\" came from a trace of one warm-up + two "
       "timed calls of bench_batch.py (arenas still growing) and one 4-byte copy per problem.\n\n```\n" + rd("busy_sweep.txt") + "```\n\n## Share of GPU time by kernel (same window)\n\n"
       + rd("shares_sweep.txt") + "\n## Where the workers' time goes (POSELIB_AMD_GROUP_TIMING=1, per call, 10 workers)\n\n```\n" + rd("batch_timing.log") + "```\n\n## rocprofv3 --stats of the whole run (warm-up included)\n\n"
       + rd("prof_sweep.md"))
    wr("r04_batch_chain.md", "# r04 - launch chain of one group of pl_estimate_batch (scripts/chain_view.py on the steady-state trace): start offset, duration, gap to the previous kernel on the stream\n\n```\n"
       + rd("chain_sweep.txt") + "```\n")
    wr("r04_batch_sweep.md",
       "# r04 - pl_estimate_batch, 4096 problems per call: host threads x group size x step budget (scripts/batch_sweep.py; every setting returns the same iterations / inliers / hypotheses)\n\n"
       "`threads:group:steps` - group 0 = the library's choice (follows the call: 64 ... 256), steps = batch steps before a group's unfinished problems are regrouped (0: never).\n\n```\n"
       + rd("batch_sweep.log") + "```\n\nExact-summation mode (POSELIB_AMD_LM_ORDERED=1):\n\n```\n" + rd("batch_sweep_ordered.log") + "```\n\n"
       "Experiments on the same workload that did not move it (round 4, other boxes of the pool; 10 threads, budget 3): LM without LDS staging of the points (two LM workgroups per CU) 75.9 k "
       "vs 75.5 k; LM workgroups of 256 lanes (four per CU) 77.6 k; both 75.3 k; flag-prefetching / 128-row-slot variants of the ordered kernel: see k_lm_ordered's comment.  "
       "GPU_MAX_HW_QUEUES 8 / 16 / 32 at 8 threads: 68.0 / 68.4 / 69.1 k (before the later changes); 4 (the runtime's default): 59.8 k.\n")
    lm = ["# r04 - one LM iteration of one refinement task (scripts/time_lm.py: pl_refine_model on resident problems, slope between a 2- and a 40-iteration run)\n",
          "tree = k_lm (default: reference order up to 256 correspondences, tree beyond); ordered = k_lm_ordered (POSELIB_AMD_LM_ORDERED=1: reference order at every n).\n"]
    for loss in ("truncated", "cauchy"):
        t, o = rd(f"time_lm_tree_{loss}.log").splitlines(), rd(f"time_lm_ordered_{loss}.log").splitlines()
        lm.append(f"\n## {loss.upper()} loss\n\n| estimator, n | tree: us per LM iteration | ordered |\n|---|---|---|")
        for a, b in zip(t, o):
            if "->" in a and "->" in b:
                lm.append(f"| {a[:16].strip()} | {a.split('->')[1].replace('us per LM iteration', '').strip()} | {b.split('->')[1].replace('us per LM iteration', '').strip()} |")
    wr("r04_lm_timing.md", "\n".join(lm))
    wr("r04_chain_add.md", "# r04 - scripts/exp/chain_add.cc: the cost of a sequential fp64 sum on one gfx950 wavefront\n\n```\n" + rd("chain_add.log") + "```\n")
    wr("r04_mfma_f64_order.md", "# r04 - scripts/exp/mfma_f64_order.cc: v_mfma_f64_* accumulate as a k-ordered chain of fused multiply-adds, bit for bit\n\n```\n" + rd("mfma_f64_order.log") + "```\n")
    wr("r04_focal_estimators_timing.log", rd("focal_timing.log"))
    for src, dst in (("focal_threads.log", "r04_focal_estimators_threads.log"), ("p35_phases.log", "r04_p35pf_phases_one_lane_per_sample.log"),
                     ("prof_focal.md", "r04_focal_estimators_kernel_trace.md")):
        if rd(src):
            wr(dst, rd(src))
    # PMC constants of the dominant kernels (the expansion of the survivor bits changed in round 4: fewer instructions per hypothesis)
    WORK = {"p3p_5000": ("k_score_mfma<10>", 320, 5000), "relpose_5000": ("k_score_mfma2<1, 10>", 320, 5000),
            "fund_10000": ("k_score_mfma2<2, 12>", 384, 10000), "hom_10000": ("k_score_mfmah<10>", 320, 10000)}

    def rows(md):
        lines = [ln for ln in md.splitlines() if ln.startswith("|")]
        cols = [c.strip() for c in lines[0].strip().strip("|").split("|")]
        return [dict(zip(cols, [c.strip() for c in ln.strip().strip("|").split("|")])) for ln in lines[2:]]

    p = os.path.join(PR, "pmc_traffic.json")
    old = json.load(open(p))
    traffic = {"_comment": old.get("_comment", "") + "  Round 4: re-measured (scripts/gpu_evidence_r04.sh b, profiles/r04_pmc_*.md) after the survivor-bit "
               "expansion of the three matrix-core scorers got its one-survivor-per-lane path; bench.py prices the VALU peak at each kernel's instruction mix "
               "(profiles/valu_mix.json, profiles/r04_valu_issue.md)."}
    for w, (kernel, chunk_pts, n) in WORK.items():
        md = rd(f"pmc_{w}.md")
        grbm = [ln for ln in rd(f"pmc_grbm_{w}.log").splitlines() if ln.startswith("{")]
        r = next((x for x in rows(md) if x["kernel"].strip("`").replace("pl::", "") == kernel and x.get("launches") == "full batch"), None) if md else None
        if not r or not grbm or "GRBM_GUI_ACTIVE" not in r or not r["GRBM_GUI_ACTIVE"]:
            traffic[w] = old[w]  # (this evidence run has no complete PMC set for the workload: keep the committed constants)
            continue
        f = lambda k: float(r[k])
        hyp = json.loads(grbm[-1])["roofline"]["hypotheses_per_launch"]
        chunks = (n + chunk_pts - 1) // chunk_pts
        cycles = f("GRBM_GUI_ACTIVE") / 8
        traffic[w] = {"kernel": kernel, "fetch_size_kb": f("FETCH_SIZE"), "write_size_kb": f("WRITE_SIZE"),
                      "traffic_bytes_per_launch": (2 * f("FETCH_SIZE") + f("WRITE_SIZE")) * 1024.0, "hypotheses_per_launch": hyp,
                      "points_per_chunk": chunk_pts, "valu_insts_per_launch": f("SQ_INSTS_VALU"),
                      "valu_insts_per_hypothesis_chunk": f("SQ_INSTS_VALU") / (hyp * chunks),
                      "valu_busy": round(f("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cycles, 3),
                      "mfma_busy": round(f("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cycles, 3), "kernel_cycles": cycles,
                      "source": f"profiles/r04_pmc_{w}.md"}
        print(w, "valu/hyp-chunk", round(traffic[w]["valu_insts_per_hypothesis_chunk"], 2), "(r03:", round(old[w]["valu_insts_per_hypothesis_chunk"], 2), ") valu_busy",
              traffic[w]["valu_busy"], "traffic MB", round(traffic[w]["traffic_bytes_per_launch"] / 1e6, 1))
    json.dump(traffic, open(p, "w"), indent=1)
    print("profiles/r04_* written")


if __name__ == "__main__":
    main()
