"""The coefficient / elimination-template tables of the two focal-length solvers (poselib_amd/csrc/pl_focal_templates.h,
oracle/src/focal_templates.inc) are GENERATED from the solvers' equations by scripts/gen_focal_templates.py.  CPU tests:
  * the committed tables are what the generator writes (nobody edited them by hand, product and oracle hold the same data);
  * where the reference's sources are present (this container): every one of the reference's 235 + 280 coefficient formulas
    (solvers/p35pf.cc:89-520, solvers/relpose_6pt_focal.cc:58-1030) is, term for term and in the written order, what the generator
    derived from the equations, and the reference's index tables place the same coefficients at the same matrix positions - the
    restatement is the reference's template, not merely a solver with the same roots."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gen_focal_templates as G  # noqa: E402

REF = "/root/reference/PoseLib/solvers"


def test_committed_tables_are_the_generators_output(tmp_path):
    tabs = [G.p35_tables(), G.six_tables()]
    body = "\n\n".join(G.emit(t, "") for t in tabs)
    product = open(os.path.join(ROOT, "poselib_amd", "csrc", "pl_focal_templates.h")).read()
    oracle = open(os.path.join(ROOT, "oracle", "src", "focal_templates.inc")).read()
    assert body in product and body in oracle
    assert [len(t["coeffs"]) for t in tabs] == [235, 280]
    assert [(t["nrows"], t["ncols"], len(t["entries"])) for t in tabs] == [(25, 35, 475), (31, 46, 814)]


def _formulas(src, decl, end):
    blk = src[src.index(decl):src.index(end)]
    out = {}
    for stmt in blk.split(";"):
        m = re.match(r"\s*coeffs\[(\d+)\]\s*=\s*(.*)$", stmt.strip(), re.S)
        if m:
            out[int(m.group(1))] = re.sub(r"\s+", " ", m.group(2))
    return out


def _terms(expr):
    """-> [(multiplier, (indices ...))] in the written order; pow(d[i], n) = n times the index i"""
    out = []
    for sign, body in re.findall(r"([+-]?)\s*((?:[^+-])+)", expr):
        body = body.strip()
        if not body:
            continue
        mult, idx = (-1 if sign == "-" else 1), []
        for f in (x.strip() for x in body.split("*")):
            if re.fullmatch(r"\d+", f):
                mult *= int(f)
            elif (m := re.fullmatch(r"d\[(\d+)\]", f)):
                idx.append(int(m.group(1)))
            elif (m := re.fullmatch(r"std::pow\(d\[(\d+)\], (\d)\)", f)):
                idx += [int(m.group(1))] * int(m.group(2))
            else:
                raise ValueError(f)
        out.append((mult, tuple(idx)))
    return out


def _table(src, name):
    m = re.search(name + r"\[\]\s*=\s*\{(.*?)\}", src, re.S)
    return [int(x) for x in m.group(1).replace("\n", " ").split(",")]


@pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference absent")
@pytest.mark.parametrize("which", ["p35pf", "six"])
def test_generated_templates_are_the_references(which):
    if which == "p35pf":
        src = open(os.path.join(REF, "p35pf.cc")).read()
        tb, rows = G.p35_tables(), 25
        formulas = _formulas(src, "double coeffs[235];", "static const int coeffs0_ind")
    else:
        src = open(os.path.join(REF, "relpose_6pt_focal.cc")).read()
        tb, rows = G.six_tables(), 31
        formulas = _formulas(src, "Eigen::VectorXd coeffs(280);", "static const int coeffs0_ind")
    assert len(formulas) == len(tb["coeffs"])
    for k, expr in formulas.items():
        mine = [(tb["coeffs"][k][t], t) for t in sorted(tb["coeffs"][k], key=G.term_key)]
        assert _terms(expr) == mine, (which, k)
    ref_entries = set()
    for ci, pos in zip(_table(src, "coeffs0_ind"), _table(src, "C0_ind")):
        ref_entries.add((pos % rows, pos // rows, ci))
    for ci, pos in zip(_table(src, "coeffs1_ind"), _table(src, "C1_ind")):
        ref_entries.add((pos % rows, rows + pos // rows, ci))
    assert ref_entries == set(tb["entries"])
