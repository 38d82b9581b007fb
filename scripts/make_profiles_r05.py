#!/usr/bin/env python
"""Assemble the committed profiles/r05_* files from gpurun_out/evidence_r05 (scripts/gpu_evidence_r05.sh a / b / c)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "evidence_r05")
PR = os.path.join(ROOT, "profiles")


def rd(name):
    p = os.path.join(EV, name)
    return open(p).read() if os.path.exists(p) else ""


def wr(name, text):
    with open(os.path.join(PR, name), "w") as f:
        f.write(text if text.endswith("\n") else text + "\n")


def main():
    line = json.loads(rd("bench_default.json").strip().splitlines()[-1])
    wr("r05_bench_line.json", json.dumps(line, indent=1))
    det = json.loads(rd("detail_default.json"))
    wr("r05_bench_detail.json", json.dumps(det["reports"], indent=1))
    wr("r05_pytest_gpu.log", rd("pytest_gpu.log"))
    c = line["config"]
    s1 = json.loads(rd("bench_s1.json").strip().splitlines()[-1])
    wr("r05_bench_default_groups_kernel_trace.md",
       f"# r05 - `python bench.py` (primary workload p3p_5000: {c['problems_per_gpu_per_step']} problems per step through pl_ransac_batch, lock-step groups of 16, 8 groups in "
       f"flight, {c['distinct_scenes']} distinct scenes) under rocprofv3 --kernel-trace --stats\n\nCommand (GPU box, from /tmp): `rocprofv3 --kernel-trace --stats -- python bench.py "
       "--no-parity --no-cpu-baseline --no-secondary --steps 5` (scripts/gpu_evidence_r05.sh b).  The bench line of the same build (scripts/gpu_evidence_r05.sh a, "
       f"`--steps 20 --warmup 5`): {line['value']:.4g} hypotheses/s, {line['ms_per_step']:.1f} ms per step, roofline frac {line['roofline']['frac']:.3f} "
       f"(peak priced at the kernel's instruction mix: {line['roofline']['issue_cycles_per_instruction']:.2f} issue cycles per VALU instruction; all-half-rate reading "
       f"{line['roofline']['frac_if_all_half_rate']:.3f}).\n\n" + rd("prof_default.md") + "\n## Device occupancy (scripts/busy.py)\n\n```\n" + rd("busy_default.txt") + "```\n")
    wr("r05_bench_p3p5000_1stream_kernel_trace.md",
       f"# r05 - one problem at a time (`bench.py --mode streams --streams 1`): {s1['ms_per_step'] / s1['config']['problems_per_gpu_per_step']:.3f} ms per 100 k-iteration P3P problem\n\n" + rd("prof_s1.md"))
    for w in ("relpose_5000", "fund_10000", "hom_10000"):
        wr(f"r05_bench_{w}_1stream_kernel_trace.md", f"# r05 - `bench.py --workload {w} --mode streams --streams 1` under rocprofv3 --kernel-trace --stats\n\n" + rd(f"prof_{w}.md"))
        wr(f"r05_bench_{w}_groups_kernel_trace.md", f"# r05 - `bench.py --workload {w}` (grouped) under rocprofv3 --kernel-trace --stats\n\n" + rd(f"profg_{w}.md"))
    for w in ("p3p_5000", "relpose_5000", "fund_10000", "hom_10000"):
        wr(f"r05_pmc_{w}.md", f"# r05 - PMC passes of `bench.py --workload {w} --mode streams --streams 1` (separate rocprofv3 --pmc runs: SQ set 1, SQ set 2, FETCH_SIZE, WRITE_SIZE; "
                              "FETCH_SIZE in KB, doubled in profiles/pmc_traffic.json per MI355X_MICROARCH.md)\n\nRound 3 for comparison: profiles/r03_pmc_*.md (k_score_mfma<10>: "
                              "SQ_INSTS_VALU 6.868e7 per launch before the one-survivor-per-lane expansion path).\n\n" + rd(f"pmc_{w}.md"))
    wr("r05_bench_batch_mixed_kernel_trace.md",
       "# r05 - configs[4]: `pl_estimate_batch`, 4096 mixed default-option problems per call, steady state\n\n"
       f"Driver's leg in the bench line of this build: **{c['batch_mixed_problems_per_s']:.0f} problems/s** ({c['batch_mixed_hyp_per_s']:.3g} hypotheses/s, parity "
       f"{c['batch_mixed_parity_ok']}).\n\nTrace: `rocprofv3 --kernel-trace -- python scripts/batch_sweep.py 4096 10:0:3` = 3 warm-up + 4 timed calls back to back, nothing else on the "
       "device; the figures below are over the LAST HALF of the kernel span (the timed calls).  Round 3's \"52 % non-idle, 14 817 copyBuffer\" came from a trace of one warm-up + two "
       "timed calls of bench_batch.py (arenas still growing) and one 4-byte copy per problem.\n\n```\n" + rd("busy_sweep.txt") + "```\n\n## Share of GPU time by kernel (same window)\n\n"
       + rd("shares_sweep.txt") + "\n## Where the workers' time goes (POSELIB_AMD_GROUP_TIMING=1, per call, 10 workers)\n\n```\n" + rd("batch_timing.log") + "```\n\n## rocprofv3 --stats of the whole run (warm-up included)\n\n"
       + rd("prof_sweep.md"))
    wr("r05_batch_chain.md", "# r05 - launch chain of one group of pl_estimate_batch (scripts/chain_view.py on the steady-state trace): start offset, duration, gap to the previous kernel on the stream\n\n```\n"
       + rd("chain_sweep.txt") + "```\n")
    wr("r05_batch_sweep.md",
       "# r05 - pl_estimate_batch, 4096 problems per call: host threads x group size x step budget (scripts/batch_sweep.py; every setting returns the same iterations / inliers / hypotheses)\n\n"
       "`threads:group:steps` - group 0 = the library's choice (follows the call: 64 ... 256), steps = batch steps before a group's unfinished problems are regrouped (0: never).\n\n```\n"
       + rd("batch_sweep.log") + "```\n\nExact-summation mode (POSELIB_AMD_LM_ORDERED=1):\n\n```\n" + rd("batch_sweep_ordered.log") + "```\n\n"
       "Experiments on the same workload that did not move it (round 4, other boxes of the pool; 10 threads, budget 3): LM without LDS staging of the points (two LM workgroups per CU) 75.9 k "
       "vs 75.5 k; LM workgroups of 256 lanes (four per CU) 77.6 k; both 75.3 k; flag-prefetching / 128-row-slot variants of the ordered kernel: see k_lm_ordered's comment.  "
       "GPU_MAX_HW_QUEUES 8 / 16 / 32 at 8 threads: 68.0 / 68.4 / 69.1 k (before the later changes); 4 (the runtime's default): 59.8 k.\n")
    wr("r05_batch_sizes.md",
       "# r05 - configs[4] as BASELINE words it: 4096 problems sharded over 8 ranks = 512 per call.  pl_estimate_batch at 256 ... 4096 problems per call (scripts/batch_sweep.py, 10 workers)\n\n```\n"
       + "".join(f"{n:5d} problems per call: {ln.strip()}\n" for n, ln in zip((256, 512, 1024, 2048), rd("batch_sizes.log").splitlines())) + rd("batch_sweep.log") + "```\n\nSeveral calls in flight from as many host threads (scripts/batch_overlap.py; pl_estimate_batch leases one of four worker pools per call since round 5):\n\n```\n"
       + rd("batch_overlap.log") + "```\n\nbench.py (N = 1 line of the same build): batch_mixed_problems_per_s " + f"{c['batch_mixed_problems_per_s']:.0f}" + " at 4096 per call, batch_mixed_512_problems_per_s "
       + f"{c.get('batch_mixed_512_problems_per_s', float('nan')):.0f}" + " (calls one after the other), batch_mixed_512_x4_in_flight_problems_per_s " + f"{c.get('batch_mixed_512_x4_in_flight_problems_per_s', float('nan')):.0f}" + ".\n\n"
       "A call is a handful of launch chains whose LENGTH is latency (per group ~6 waits of ~0.9 ms: ~15 dependent launches, the LO's up to 25 and the final bundle's up to 100 LM iterations at 8 us): "
       "t(call) = 5.4 ms + 11.2 us x problems.  One call at a time a 512-problem shard therefore runs at 57 % of the 4096-problem rate; a rank that keeps four shards in flight gets 80 %.\n")
    wr("r05_generator_full_device.md", "# r05 - the 5-point generator on a FULL device: `scripts/exp/genbench 1600000 16 3` (16 problems x 100 k iterations), flat root isolation (default) against round 4's kernel "
       "(POSELIB_AMD_REL_ROOTS_V1=1), same box\n\n```\n" + rd("genbench_roots_ab.log") + "```\n\n## rocprofv3 --kernel-trace --stats, default build\n\n" + rd("prof_gen_v3.md") + "\n## ... round 4's root kernel\n\n" + rd("prof_gen_v1.md"))
    lm = ["# r05 - one LM iteration of one refinement task (scripts/time_lm.py: pl_refine_model on resident problems, slope between a 2- and a 40-iteration run)\n",
          "tree = k_lm (default for poses and homographies: reference order up to 256 correspondences, tree beyond); ordered = k_lm_ordered (POSELIB_AMD_LM_ORDERED=1: reference order at every n).  "
          "Fundamental matrices take k_lm_ordered in BOTH columns (their default since round 5: the sign of a refined F needs bit-identical sums; round 4's tree kernel: 13.5 / 17.1 / 21.4 / 31.1 / 47.3 us).  "
          "Round 4 for comparison (tree): abs 10.0 / 12.7 / 15.5 / 22.0 / 35.2, rel 9.6 / 12.1 / 14.6 / 20.7 / 32.3, hom 13.6 / 17.2 / 19.4 / 27.0 / 40.3 us - the difference is the block reduction (BlockReduceT, r05_lm_profile.md).\n"]
    for loss in ("truncated", "cauchy"):
        t, o = rd(f"time_lm_tree_{loss}.log").splitlines(), rd(f"time_lm_ordered_{loss}.log").splitlines()
        lm.append(f"\n## {loss.upper()} loss\n\n| estimator, n | tree: us per LM iteration | ordered |\n|---|---|---|")
        for a, b in zip(t, o):
            if "->" in a and "->" in b:
                lm.append(f"| {a[:16].strip()} | {a.split('->')[1].replace('us per LM iteration', '').strip()} | {b.split('->')[1].replace('us per LM iteration', '').strip()} |")
    wr("r05_lm_timing.md", "\n".join(lm))
    pass  # (round-4 experiment, not re-run)
    pass  # (round-4 experiment, not re-run)
    wr("r05_focal_estimators_timing.log", rd("focal_timing.log"))
    for src, dst in (("focal_threads.log", "r05_focal_estimators_threads.log"), ("prof_focal.md", "r05_focal_estimators_kernel_trace.md"),
                     ("soak_focal_device_vs_reference.md", "r05_soak_focal_device_vs_reference.md"), ("select_bench.log", "r05_select_bench.md"),
                     ("lm_profile.md", "r05_lm_profile_final_build.md")):
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
    # (the comment is REWRITTEN, not appended to: round 4's script grew it by a sentence per run)
    traffic = {"_comment": "PMC constants of the dominant kernels (the streaming scorers), measured with rocprofv3 --pmc in separate passes (scripts/gpu_evidence_r05.sh b; "
               "summaries in profiles/r05_pmc_*.md), one problem at a time.  FETCH_SIZE (KB) is doubled per MI355X_MICROARCH.md (gfx950 under-reports wide streaming reads by 2x); "
               "WRITE_SIZE (KB) is taken as is.  valu_insts_per_hypothesis_chunk = SQ_INSTS_VALU of one launch / (hypotheses x point chunks of the launch): bench.py's valu_issue "
               "roofline multiplies it back with the hypotheses and chunks of its own launches (PMC needs the profiler) and prices the VALU peak at each kernel's instruction mix "
               "(profiles/valu_mix.json, profiles/r04_valu_issue.md).  A workload without a complete PMC set in the latest evidence run keeps its earlier constants (see `source`)."}
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
                      "source": f"profiles/r05_pmc_{w}.md"}
        print(w, "valu/hyp-chunk", round(traffic[w]["valu_insts_per_hypothesis_chunk"], 2), "(r03:", round(old[w]["valu_insts_per_hypothesis_chunk"], 2), ") valu_busy",
              traffic[w]["valu_busy"], "traffic MB", round(traffic[w]["traffic_bytes_per_launch"] / 1e6, 1))
    json.dump(traffic, open(p, "w"), indent=1)
    print("profiles/r05_* written")


if __name__ == "__main__":
    main()
