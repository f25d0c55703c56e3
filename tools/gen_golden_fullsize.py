#!/usr/bin/env python3
"""Goldens of the SURVEY 8(f) analyses at bench-like sizes, from the real reference (VERDICT r03 "next round" item 4):

  * covarion DNA 200 x 10 000 and covarion protein 100 x 3 000: lnL of oracle/_ref/mb_scalar;
  * ancestral states of three constrained nodes on the configs[1] shape (DNA 500 x 20 000, GTR + gamma-4): the whole first row
    of mb_scalar's .p file (lnL and every state probability);
  * a partitioned analysis of two divisions of 200 x 10 000 (what the BEAGLE v3 build runs as ONE multi-partition instance):
    the initial lnL of oracle/_ref/mb (native FMA kernels);
  * the parsimony model (Likelihood_Pars) on the configs[3] shape, DNA 1000 x 50 000: appended to tests/golden/parsmodel.json.

-> tests/golden/fullsize.json + tests/golden/fullsize_anc.npz.  Run in the build container (minutes: the reference's pattern
compression is quadratic in the alignment length).    python tools/gen_golden_fullsize.py [covarion anc v3 pars]
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_mrbayes_dropin as dropin   # noqa: E402
from tests import test_reports_dropin as rep      # noqa: E402
from tools import gen_golden_pars, refrun         # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
V3 = dict(ntaxa=200, nsites=20000, first=10000, same_shape=True)
PARS_C4 = {"name": "dna_c4_shape", "ntaxa": 1000, "nsites": 50000, "nstates": 4, "seed": 7, "tree_seed": 3, "p_gap": 0.0}


def main(what):
    path = os.path.join(GOLD, "fullsize.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out["source"] = "tools/gen_golden_fullsize.py: oracle/_ref/mb_scalar (reports, covarion), oracle/_ref/mb (partitioned analysis)"
    if "covarion" in what:
        for case in ("covarion_dna_200x10000", "covarion_protein_100x3000"):
            _, h, r = rep._row0(rep.REF_SCALAR, case, None)
            out[case] = {"lnLike": r[h.index("lnLike")], "columns": len(h)}
            print(case, out[case])
    if "anc" in what:
        case = "dna_anc_500x20000"
        _, h, r = rep._row0(rep.REF_SCALAR, case, None)
        np.savez_compressed(os.path.join(GOLD, "fullsize_anc.npz"), row=np.asarray(r, dtype=np.float64).astype(np.float32))
        out[case] = {"lnLike": r[h.index("lnLike")], "columns": len(h), "header_sha1": hashlib.sha1("\t".join(h).encode()).hexdigest(),
                     "lnLike_column": h.index("lnLike"), "state_probability_columns": sum(1 for x in h if x.startswith("p("))}
        print(case, out[case])
    if "v3" in what:
        native = refrun.initial_lnl(refrun.run_mb(refrun.REF_MB, dropin._partitioned_nexus(None, **V3))[0])
        out["partitioned_2x200x10000"] = {"initial_lnL_native": native, "args": V3}
        print("partitioned", native)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    if "pars" in what:
        ppath = os.path.join(GOLD, "parsmodel.json")
        pm = json.load(open(ppath))
        text, _ = refrun.run_mb(refrun.REF_MB, gen_golden_pars.nexus(PARS_C4), timeout=7200)
        lnl = refrun.initial_lnl(text)
        pm["cases"] = [c for c in pm["cases"] if c["name"] != PARS_C4["name"]] + [dict(PARS_C4, lnL_reference=lnl)]
        with open(ppath, "w") as fh:
            json.dump(pm, fh, indent=1)
        print("parsimony model at the configs[3] shape", lnl)


if __name__ == "__main__":
    main(sys.argv[1:] or ["covarion", "anc", "v3", "pars"])
