#!/usr/bin/env python
"""Assemble the committed profiles/r06_* files from gpurun_out/evidence_r06 (scripts/gpu_evidence_r06.sh a / b)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "evidence_r06")
PR = os.path.join(ROOT, "profiles")


def rd(name):
    p = os.path.join(EV, name)
    return open(p).read() if os.path.exists(p) else ""


def wr(name, text):
    with open(os.path.join(PR, name), "w") as f:
        f.write(text if text.endswith("\n") else text + "\n")


WORK = {"p3p_5000": ("k_score_mfma<10>", 320, 5000), "relpose_5000": ("k_score_mfma2<1, 10>", 320, 5000),
        "fund_10000": ("k_score_mfma2<2, 12>", 384, 10000), "hom_10000": ("k_score_mfmah<10>", 320, 10000)}


def refresh_pmc_constants():
    """profiles/pmc_traffic.json from the PMC passes of scripts/gpu_evidence_r06.sh p (round 6: the headline scorer has a new filter,
    all four have a new reduction in their exact pass) - run BEFORE the bench of part a, which prices its roofline with them"""
    def rows(md):
        lines = [ln for ln in md.splitlines() if ln.startswith("|")]
        cols = [c.strip() for c in lines[0].strip().strip("|").split("|")]
        return [dict(zip(cols, [c.strip() for c in ln.strip().strip("|").split("|")])) for ln in lines[2:]]

    p = os.path.join(PR, "pmc_traffic.json")
    old = json.load(open(p))
    traffic = {"_comment": old.get("_comment", "").split("  Round 6:")[0] + "  Round 6: re-measured (scripts/gpu_evidence_r06.sh p, profiles/r06_pmc_*.md): "
               "k_score_mfma has three half-planes per pair and 32 hypotheses per tile, every scorer sums the runs of its exact pass over DPP."}
    for w, (kernel, chunk_pts, n) in WORK.items():
        md = rd(f"pmc_{w}.md")
        grbm = [ln for ln in rd(f"pmc_grbm_{w}.log").splitlines() if ln.startswith("{")]
        r = next((x for x in rows(md) if x["kernel"].strip("`").replace("pl::", "") == kernel and x.get("launches") == "full batch"), None) if md else None
        need = ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES")
        if not r or not grbm or any(not r.get(k) for k in need):
            traffic[w] = old[w]  # (this evidence run has no complete PMC set for the workload: keep the committed constants)
            print(w, "kept", old[w].get("source"))
            continue
        f = lambda k: float(r[k])
        hyp = json.loads(grbm[-1])["roofline"]["hypotheses_per_launch"]
        chunks = (n + chunk_pts - 1) // chunk_pts
        cycles = f("GRBM_GUI_ACTIVE") / 8
        traffic[w] = {"kernel": kernel, "fetch_size_kb": f("FETCH_SIZE"), "write_size_kb": f("WRITE_SIZE"),
                      "traffic_bytes_per_launch": (2 * f("FETCH_SIZE") + f("WRITE_SIZE")) * 1024.0, "hypotheses_per_launch": hyp,
                      "points_per_chunk": chunk_pts, "valu_insts_per_launch": f("SQ_INSTS_VALU"),
                      "valu_insts_per_hypothesis_chunk": f("SQ_INSTS_VALU") / (hyp * chunks),
                      "mfma_insts_per_launch": float(r["SQ_INSTS_MFMA"]) if r.get("SQ_INSTS_MFMA") else None,
                      "valu_busy": round(f("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cycles, 3),
                      "mfma_busy": round(f("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cycles, 3), "kernel_cycles": cycles,
                      "source": f"profiles/r06_pmc_{w}.md"}
        print(w, "valu/hyp-chunk", round(traffic[w]["valu_insts_per_hypothesis_chunk"], 2), "(before:", round(old[w]["valu_insts_per_hypothesis_chunk"], 2), ") valu_busy",
              traffic[w]["valu_busy"], "traffic MB", round(traffic[w]["traffic_bytes_per_launch"] / 1e6, 1))
        wr(f"r06_pmc_{w}.md", f"# r06 - PMC passes of `bench.py --workload {w} --mode streams --streams 1` (separate rocprofv3 --pmc runs: SQ set 1, SQ set 2, "
                              "GRBM, FETCH_SIZE, WRITE_SIZE; FETCH_SIZE in KB, doubled in profiles/pmc_traffic.json per MI355X_MICROARCH.md)\n\n" + md)
    json.dump(traffic, open(p, "w"), indent=1)


def main():
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":
        refresh_pmc_constants()
        return
    line = json.loads(rd("bench_default.json").strip().splitlines()[-1])
    wr("r06_bench_line.json", json.dumps(line, indent=1))
    det = json.loads(rd("detail_default.json"))
    wr("r06_bench_detail.json", json.dumps(det["reports"], indent=1))
    wr("r06_pytest_gpu.log", rd("pytest_gpu.log"))
    c = line["config"]
    s1 = json.loads(rd("bench_s1.json").strip().splitlines()[-1])
    rf = line["roofline"]
    wr("r06_bench_default_groups_kernel_trace.md",
       f"# r06 - `python bench.py` (primary workload p3p_5000: {c['problems_per_gpu_per_step']} problems per step through pl_ransac_batch, "
       f"lock-step groups of 16, 8 groups in flight, {c['distinct_scenes']} distinct scenes) under rocprofv3 --kernel-trace --stats\n\n"
       "Command (GPU box, from /tmp): `rocprofv3 --kernel-trace --stats -- python bench.py --no-parity --no-cpu-baseline --no-secondary "
       "--steps 5` (scripts/gpu_evidence_r06.sh b).  The bench line of the same build (scripts/gpu_evidence_r06.sh a, `--steps 20 "
       f"--warmup 5`): {line['value']:.4g} hypotheses/s, {line['ms_per_step']:.1f} ms per step, roofline frac {rf['frac']:.3f} (peak priced "
       f"at the kernel's instruction mix: {rf['issue_cycles_per_instruction']:.2f} issue cycles per VALU instruction; all-half-rate reading "
       f"{rf['frac_if_all_half_rate']:.3f}); the timed kernel's average launch by HIP events {rf['avg_launch_ms']:.4f} ms, its solo launch "
       f"{rf['solo_avg_launch_ms']:.4f} ms.\n\n" + rd("prof_default.md") + "\n## Device occupancy (scripts/busy.py)\n\n```\n" + rd("busy_default.txt") + "```\n")
    wr("r06_bench_p3p5000_1stream_kernel_trace.md",
       f"# r06 - one problem at a time (`bench.py --mode streams --streams 1`): {s1['ms_per_step'] / s1['config']['problems_per_gpu_per_step']:.3f} "
       "ms per 100 k-iteration P3P problem\n\n" + rd("prof_s1.md"))
    for w in ("relpose_5000",):
        wr(f"r06_bench_{w}_1stream_kernel_trace.md", f"# r06 - `bench.py --workload {w} --mode streams --streams 1` under rocprofv3 --kernel-trace --stats\n\n" + rd(f"prof_{w}.md"))
        wr(f"r06_bench_{w}_groups_kernel_trace.md", f"# r06 - `bench.py --workload {w}` (grouped) under rocprofv3 --kernel-trace --stats\n\n" + rd(f"profg_{w}.md"))
    for w in ("p3p_5000", "relpose_5000", "fund_10000", "hom_10000"):
        if not rd(f"pmc_{w}.md"):
            continue
        wr(f"r06_pmc_{w}.md", f"# r06 - PMC passes of `bench.py --workload {w} --mode streams --streams 1` (separate rocprofv3 --pmc runs: SQ set 1, SQ set 2, "
                              "GRBM, FETCH_SIZE, WRITE_SIZE; FETCH_SIZE in KB, doubled in profiles/pmc_traffic.json per MI355X_MICROARCH.md)\n\n" + rd(f"pmc_{w}.md"))
    sizes = rd("batch_sizes.log").splitlines()
    wr("r06_batch_sizes.md",
       "# r06 - configs[4] as BASELINE words it: 4096 problems sharded over 8 ranks = 512 per call.  pl_estimate_batch at 256 ... 4096 problems per call "
       "(scripts/batch_sweep.py, 9 workers)\n\n```\n" + "".join(f"{n:5d} problems per call: {ln.strip()}\n" for n, ln in zip((256, 512, 1024, 2048), sizes))
       + rd("batch_sweep.log") + "```\n\n512 per call with the batches of a group's members DOUBLING as in round 5 (POSELIB_AMD_GROUP_JUMP=0):\n\n```\n"
       + rd("batch_512_doubling.log") + "```\n\nSeveral calls in flight from as many host threads (scripts/batch_overlap.py):\n\n```\n" + rd("batch_overlap.log")
       + "```\n\nWhere the workers' time goes in a 512-problem call (POSELIB_AMD_GROUP_TIMING=1):\n\n```\n" + rd("batch_timing_512.log") + "```\n\n"
       f"bench.py (N = 1 line of the same build): batch_mixed_problems_per_s {c['batch_mixed_problems_per_s']:.0f} at 4096 per call, batch_mixed_512_problems_per_s "
       f"{c.get('batch_mixed_512_problems_per_s', float('nan')):.0f} (calls one after the other), batch_mixed_512_x4_in_flight_problems_per_s "
       f"{c.get('batch_mixed_512_x4_in_flight_problems_per_s', float('nan')):.0f}.\n\n"
       "Round 5: t(call) = 5.4 ms + 11.2 us x problems - 512 per call at 57 % of the 4096-per-call rate.  Round 6: from its second step on a group's member "
       "asks for the iterations its loop is known to need (up to 8192) instead of doubling; the 5-point problems with 60 - 70 % outliers then finish inside "
       "the group's first round of steps and the call loses its second round: 10.5 -> 8.8 ms, 70 % of the 4096-per-call rate.  What is left is the chain of "
       "one group: stage A, two or three steps (each bound by its k_lm launch: ~370 LO tasks of a 57-problem group), the tail (final refinement, mask, "
       "bundle).  Kernel trace of such a call: profiles/r06_bench_batch_512_kernel_trace.md.\n")
    wr("r06_bench_batch_512_kernel_trace.md", "# r06 - `scripts/batch_sweep.py 512 9:0:3` (3 warm-up + 4 timed pl_estimate_batch calls of 512 mixed problems) under "
       "rocprofv3 --kernel-trace --stats\n\n" + rd("prof_b512.md"))
    wr("r06_batch_group_members.md", "# r06 - OPENCV cameras, PROSAC and warm starts inside the lock-step groups of pl_estimate_batch (scripts/batch_cameras.py)\n\n"
       "VERDICT r5 next 7: \"a 4096-problem OPENCV batch within 15 % of the SIMPLE_PINHOLE rate\"; round 5 ran such items one at a time (~1 / 20 of the rate).  "
       "`grouped` / `solo` / `fallback`: pl_last_batch_report of the last call.\n\n" + rd("batch_cameras.md") +
       "\nPROSAC pays the host-side draws of every member's samples per step (sampling.cc:85-136 is sequential); the warm starts of a group are scored by one launch and "
       "refined by one k_lm launch in front of the lock-step loop (a first form with two synchronisations per member: 26.8 k problems/s); started from the true pose "
       "hardly any minimal model beats the incumbent, so the batch has almost no local optimisations left to run - hence the rate above the cold one.\n")
    wr("r06_focal_batch.md", rd("focal_batch.md").replace("# r05 -", "# r06 -") + "\n## One problem at a time (scripts/time_focal_estimators.py)\n\n```\n" + rd("focal_timing.log") + "```\n")
    wr("r06_focal_estimators_kernel_trace.md", "# r06 - `scripts/focal_threads.py 1` under rocprofv3 --kernel-trace --stats (the focal estimators: batch calls and single problems)\n\n" + rd("prof_focal.md"))
    print("profiles/r06_* written")


if __name__ == "__main__":
    main()
