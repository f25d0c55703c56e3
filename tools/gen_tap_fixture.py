#!/usr/bin/env python3
"""Write tests/golden/primates_tap.json: the reference's own integration check (testing/test1.nex style: 20 000
generations, 2 runs x 4 chains, nst=mixed rates=invgamma) run with the REAL reference's native kernels
(oracle/_ref/mb) on the primates patterns (columns re-expanded from tests/golden/primates_gtr_g4.npz), recording
the statistics the engine-driven run must reproduce within the reference's own statistical tolerance."""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrbayes_amd import data as mbdata      # noqa: E402
from tools import refrun                    # noqa: E402


def alignment_from_fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "primates_gtr_g4.npz"))
    bits, w = z["bits"], z["weights"].astype(int)
    inv = {}
    for ch in "ACGTRYMKSWHBVDN":
        inv.setdefault(mbdata.dna_bits(ch), ch)
    seqs = []
    for t in range(bits.shape[0]):
        seqs.append("".join(inv[int(b)] * int(n) for b, n in zip(bits[t], w)))
    return ["t%d" % (i + 1) for i in range(bits.shape[0])], seqs


def tap_nexus(names, seqs, ngen, beagle=None):
    s = "#NEXUS\nbegin data;\n  dimensions ntax=%d nchar=%d;\n  format datatype=dna gap=- missing=?;\n  matrix\n" % (len(names), len(seqs[0]))
    for n, q in zip(names, seqs):
        s += "%s  %s\n" % (n, q)
    s += "  ;\nend;\nbegin mrbayes;\n  set autoclose=yes nowarnings=yes seed=4711 swapseed=4711;\n"
    s += "  lset nst=mixed rates=invgamma;\n"
    if beagle:
        s += "  set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=%s;\n" % beagle
    s += "  mcmc ng=%d;\n  sump;\nend;\n" % ngen
    return s


def parse(out):
    m = re.search(r'Likelihood of best state for "cold" chain of run 1 was (-?[0-9.]+)', out)
    tl = re.search(r"^\s*TL\s+([0-9.]+)", out, re.M)
    return {"completed": len(re.findall(r"Analysis completed in", out)), "best_cold_lnL_run1": float(m.group(1)) if m else None,
            "TL_mean": float(tl.group(1)) if tl else None}


if __name__ == "__main__":
    names, seqs = alignment_from_fixture()
    out, wall = refrun.run_mb(refrun.REF_MB, tap_nexus(names, seqs, 20000))
    res = parse(out)
    res["wall_s"] = wall
    res["source"] = "oracle/_ref/mb (native FMA kernels), ngen=20000, nruns=2, nchains=4, nst=mixed rates=invgamma"
    print(res)
    with open(os.path.join(ROOT, "tests", "golden", "primates_tap.json"), "w") as fh:
        json.dump(res, fh, indent=1)
