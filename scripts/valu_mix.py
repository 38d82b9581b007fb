#!/usr/bin/env python
"""Classify the vector instructions of a gfx950 kernel by issue cost (profiles/r04_valu_issue.md) and price its hot loop.

    python scripts/valu_mix.py poselib_amd/csrc/kernels.o k_score_mfmaILi10E          # substring of the mangled name
    python scripts/valu_mix.py poselib_amd/csrc/kernels.o k_score_mfmaILi10E --json   # what bench.py reads (committed:
                                                                                       # profiles/valu_mix.json)
Classes (measured, 8 wavefronts per SIMD): full rate 2.3 cycles per wave64 instruction - v_mul_f32, v_add_f32, v_add/sub_u32,
v_and/or/xor_b32, v_mov_b32, v_fma_f32 with at most one VGPR source; quarter rate 8.3 - v_rcp/rsq/sqrt/exp/log_f32, f16 FMA; 16.4 -
fp64 transcendentals; MFMA: 8.4 issue cycles (scripts/exp/overlap.cc); everything else half rate 4.2.
The hot loop is the innermost backward branch that encloses a v_mfma (or, without MFMA, the innermost loop with the most vector
instructions)."""
import json
import os
import re
import subprocess
import sys
import tempfile

FULL = 2.3
HALF = 4.2
QUARTER = 8.3
F64_TRANS = 16.4
MFMA_ISSUE = 8.4
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

_FULL_OPS = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32",
             "v_xor_b32", "v_mov_b32", "v_nop"}
_QUARTER_OPS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_fma_f16", "v_rcp_iflag_f32"}
_F64_TRANS = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}


def classify(mnemonic, operands):
    """-> (class name, cycles) of one vector instruction; None for non-vector-ALU instructions."""
    m = mnemonic
    if not m.startswith("v_"):
        return None
    if m.startswith("v_mfma") or m.startswith("v_smfmac"):
        return "mfma", MFMA_ISSUE
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", m)
    dpp = "dpp" in m or "row_" in operands or "quad_perm" in operands
    if m in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"):
        return "half", HALF
    if base in _F64_TRANS:
        return "f64_trans", F64_TRANS
    if base in _QUARTER_OPS:
        return "quarter", QUARTER
    if dpp:
        return "half", HALF
    if base in _FULL_OPS:
        return "full", FULL
    if base == "v_fma_f32":
        srcs = [o.strip() for o in operands.split(",")[1:4]]
        nv = sum(1 for o in srcs if re.match(r"^-?\|?v\d+|^-?\|?v\[", o))
        return ("full", FULL) if nv <= 1 else ("half", HALF)
    return "half", HALF


def disassemble(obj):
    """gfx950 code object of a host object / shared library -> list of (symbol, [(addr, mnemonic, operands)])."""
    d = tempfile.mkdtemp(prefix="valu_mix_")
    b = os.path.join(d, os.path.basename(obj))
    subprocess.check_call(["cp", obj, b])
    subprocess.call([OBJDUMP, "--offloading", b], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = [f for f in os.listdir(d) if "gfx950" in f]
    if not co:
        raise SystemExit(f"no gfx950 code object in {obj}")
    txt = subprocess.check_output([OBJDUMP, "-d", os.path.join(d, co[0])], stderr=subprocess.DEVNULL).decode()
    kernels, cur = [], None
    for line in txt.splitlines():
        mm = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if mm:
            cur = (mm.group(1), [])
            kernels.append(cur)
            continue
        mm = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if mm and cur is not None:
            cur[1].append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
    return kernels


def loops(ins):
    """backward branches: (start index, end index) with the branch at `end`."""
    addr_to_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    out = []
    for i, (a, m, ops) in enumerate(ins):
        if m.startswith("s_cbranch") or m == "s_branch":
            mm = re.match(r"^(\d+)", ops)
            if not mm:
                continue
            off = int(mm.group(1))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in addr_to_idx:
                out.append((addr_to_idx[tgt], i))
    return out


def mix(ins):
    c = {"full": 0, "half": 0, "quarter": 0, "f64_trans": 0, "mfma": 0}
    cyc = 0.0
    for _, m, ops in ins:
        r = classify(m, ops)
        if r:
            c[r[0]] += 1
            cyc += r[1]
    c["valu"] = c["full"] + c["half"] + c["quarter"] + c["f64_trans"]
    c["issue_cycles"] = round(cyc, 1)
    return c


def analyse(obj, pattern):
    ks = [k for k in disassemble(obj) if pattern in k[0]]
    if not ks:
        raise SystemExit(f"no kernel matching {pattern}")
    name, ins = ks[0]
    ls = loops(ins)
    hot = None
    with_mfma = [(s, e) for (s, e) in ls if any(m.startswith("v_mfma") for _, m, _ in ins[s:e + 1])]
    cands = with_mfma or ls
    if cands:
        if with_mfma:
            hot = min(cands, key=lambda se: se[1] - se[0])
        else:
            hot = max(cands, key=lambda se: sum(1 for _, m, _ in ins[se[0]:se[1] + 1] if m.startswith("v_")) / (1 + 0.01 * (se[1] - se[0])))
    res = {"kernel": name, "whole_kernel_static": mix(ins)}
    mf = [k for k, (_, m, _) in enumerate(ins) if m.startswith("v_mfma")]
    n_hot_mfma = sum(1 for _, m, _ in ins[hot[0]:hot[1] + 1] if m.startswith("v_mfma")) if (hot and with_mfma) else 0
    if mf and (not with_mfma or n_hot_mfma > 12):
        # a completely unrolled tile loop (k_score_mfma since round 6): the straight-line stretch from the first MFMA to the first
        # branch behind the last one takes the loop's place (ONE "iteration" that holds all point groups of a tile)
        end = mf[-1]
        while end + 1 < len(ins) and not ins[end + 1][1].startswith("s_cbranch") and not ins[end + 1][1].startswith("s_branch"):
            end += 1
        hot = (mf[0], end)
        res["hot_loop_is_unrolled_stretch"] = True
    if hot:
        res["hot_loop"] = mix(ins[hot[0]:hot[1] + 1])
        rest = ins[:hot[0]] + ins[hot[1] + 1:]
        res["outside_hot_loop_static"] = mix(rest)
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    r = analyse(args[0], args[1])
    if "--json" in sys.argv:
        print(json.dumps(r))
    else:
        print(r["kernel"])
        for k in ("hot_loop", "outside_hot_loop_static", "whole_kernel_static"):
            if k in r:
                m = r[k]
                per = m["issue_cycles"] / max(1, m["valu"] + m["mfma"])
                print(f"  {k:26s} valu {m['valu']:5d} (full {m['full']}, half {m['half']}, quarter {m['quarter']}, f64 trans {m['f64_trans']})"
                      f"  mfma {m['mfma']}  issue cycles {m['issue_cycles']}  = {per:.2f} per instruction")
