#!/usr/bin/env python3
"""Write tests/golden/std_cynmix.json: the MORPHOLOGY division (characters 1-166, datatype standard) of the reference's
examples/cynmix.nex, as the taxon names and the rows of that first interleave block, plus the log-likelihood the reference's scalar
build (oracle/_ref/mb_scalar) prints for it on a fixed tree under Mk + gamma, coding=variable (tests/test_std_dropin.py builds the
NEXUS text from these; the GPU box has no /root/reference).

    python tools/gen_std_fixture.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import std_cases  # noqa: E402
from tools import refrun  # noqa: E402


def main():
    src = os.path.join(refrun.REFERENCE, "examples", "cynmix.nex") if hasattr(refrun, "REFERENCE") else "/root/reference/examples/cynmix.nex"
    names, rows = [], []
    with open(src) as fh:
        lines = fh.read().split("\n")
    i = next(k for k, l in enumerate(lines) if l.strip().lower() == "matrix") + 1
    while lines[i].strip():
        n, seq = lines[i].split()
        names.append(n)
        rows.append(seq)
        i += 1
    assert len(names) == 32 and all(std_cases.nchar_of(r) == 166 for r in rows), (len(names), [std_cases.nchar_of(r) for r in rows])
    fix = {"source": "examples/cynmix.nex, characters 1-166 (Nylander et al. 2004, Syst. Biol. 53:47-67)", "names": names, "rows": rows}
    out = {}
    for key, kw in std_cases.CYNMIX_CONFIGS.items():
        text = std_cases.cynmix_nexus(fix, None, **kw)
        o, row = refrun.run_mb_with_samples(os.path.join(ROOT, "oracle", "_ref", "mb_scalar"), text)
        assert "Analysis completed" in o, o[-1500:]
        out[key] = row["LnL"] if "LnL" in row else row["lnLike"]
    fix["lnL_mb_scalar"] = out
    with open(os.path.join(ROOT, "tests", "golden", "std_cynmix.json"), "w") as fh:
        json.dump(fix, fh, indent=0)
    print(out)


def hymfossil():
    """tests/golden/std_hymfossil.json: the morphology (characters 1-353, datatype standard) of examples/hymfossil.nex -- its taxa that
    have any morphological data --, the ordered and excluded characters its MrBayes block names, and what the reference's scalar build
    prints for it on a fixed tree."""
    import re
    with open("/root/reference/examples/hymfossil.nex") as fh:
        text = fh.read()
    body = text[text.lower().index("matrix") + 6:]
    body = body[:body.index(";")]
    body = re.sub(r"\[[^\]]*\]", "", body)
    seqs, order = {}, []
    for line in body.split("\n"):
        f = line.split()
        if len(f) < 2:
            continue
        if f[0] not in seqs:
            seqs[f[0]] = ""
            order.append(f[0])
        if std_cases.nchar_of(seqs[f[0]]) < 353:
            seqs[f[0]] += "".join(f[1:])
    rows = []
    for n in order:
        r, k, i = seqs[n], 0, 0
        while k < 353:                              # the first 353 characters (a polymorphism in braces is one)
            if r[i] in "{(":
                i = r.index("}" if r[i] == "{" else ")", i)
            i += 1
            k += 1
        rows.append(r[:i])
    names = [n for n, r in zip(order, rows) if set(r) - set("?-")]
    rows = [r for r in rows if set(r) - set("?-")]
    block = text[text.lower().index("begin mrbayes"):]
    def charset(name):
        m = re.search(r"charset\s+%s\s*=([^;]*);" % name, block)
        return [int(x) for x in m.group(1).split()]
    fix = {"source": "examples/hymfossil.nex, characters 1-353 (Ronquist et al. 2012, Syst. Biol. 61:973-999)", "names": names, "rows": rows, "nchar": 353,
           "ordered": charset("morph_ordered"), "excluded": charset("morph_excluded") + charset("morph_constant")}
    out = {}
    for key, kw in std_cases.HYM_CONFIGS.items():
        o, row = refrun.run_mb_with_samples(os.path.join(ROOT, "oracle", "_ref", "mb_scalar"), std_cases.hymfossil_nexus(fix, None, **kw))
        assert "Analysis completed" in o, o[-1500:]
        out[key] = row["LnL"] if "LnL" in row else row["lnLike"]
    fix["lnL_mb_scalar"] = out
    with open(os.path.join(ROOT, "tests", "golden", "std_hymfossil.json"), "w") as fh:
        json.dump(fix, fh, indent=0)
    print(len(names), "taxa", out)


if __name__ == "__main__":
    main()
    hymfossil()
