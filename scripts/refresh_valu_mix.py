#!/usr/bin/env python
"""Regenerate profiles/valu_mix.json (the instruction mixes bench.py prices the scorers' VALU peak with) from the built
poselib_amd/csrc/kernels.o:   python scripts/refresh_valu_mix.py"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import valu_mix  # noqa: E402

KERNELS = {  # name in the bench line -> (substring of the mangled name, hypotheses per tile-loop iteration, iterations per chunk)
    "k_score_mfma<10>": ("k_score_mfmaILi10EEEvNS_8PointSet", 32, 10),
    "k_score_mfma2<1,10>": ("k_score_mfma2ILi1ELi10EEEvNS_8PointSet", None, None),
    "k_score_mfma2<2,12>": ("k_score_mfma2ILi2ELi12EEEvNS_8PointSet", None, None),
    "k_score_mfmah<10>": ("k_score_mfmahILi10EEEvNS_8PointSet", None, None),
}
path = os.path.join(ROOT, "profiles", "valu_mix.json")
old = json.load(open(path))
out = {}
for name, (pat, hyp, its) in KERNELS.items():
    r = valu_mix.analyse(os.path.join(ROOT, "poselib_amd", "csrc", "kernels.o"), pat)
    if hyp:
        r["hot_loop_hypotheses_per_iteration"] = hyp
        r["hot_loop_iterations_per_chunk"] = 1 if r.get("hot_loop_is_unrolled_stretch") else its
    out[name] = r
    print(name, r.get("hot_loop"))
cyc = dict(old["_class_cycles"])
cyc["mfma_pipe"] = 32.6
cyc["mfma_pipe_source"] = "scripts/exp/overlap.cc (round 3: 32.6 cycles per v_mfma_f32_32x32x16_f16), profiles/r06_overlap2.md (round 6: the pipe's time does not run under the vector instructions that consume its results)"
out["_class_cycles"] = cyc
out["_note"] = old["_note"] + "  Round 6: k_score_mfma<10>'s tile loop is software-pipelined and unrolled completely: the straight-line stretch from its first MFMA to the first branch behind its last one is counted as ONE iteration per tile of 32 hypotheses (hot_loop_is_unrolled_stretch)."
json.dump(out, open(path, "w"), indent=1)
