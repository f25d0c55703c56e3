#!/usr/bin/env python3
"""Golden vectors for the device parsimony scorer from the reference's OWN parsimony-model likelihood (Likelihood_Pars,
src/likelihood.c:7593-7700: lnL = -(tree length + number of characters) * ln(number of states)): oracle/_ref/mb with
`lset parsmodel=yes` on a fixed tree prints `Chain 1 -- <lnL>`.  Writes tests/golden/parsmodel.json (run in the build container)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrbayes_amd import data as mbdata, tree as mbtree     # noqa: E402
from tools import refrun                                   # noqa: E402

CASES = [{"name": "dna", "ntaxa": 20, "nsites": 400, "nstates": 4, "seed": 11, "tree_seed": 12, "p_gap": 0.03},
         {"name": "dna_large", "ntaxa": 120, "nsites": 3000, "nstates": 4, "seed": 13, "tree_seed": 14, "p_gap": 0.05},
         {"name": "dna_gappy", "ntaxa": 40, "nsites": 1000, "nstates": 4, "seed": 17, "tree_seed": 18, "p_gap": 0.3}]
# (the reference's parsimony model does not survive protein data here: its run stops before the first log-likelihood)
_AA = "ARNDCQEGHILKMFPSTWYV"


def nexus(case):
    st = mbdata.synthetic_states(case["ntaxa"], case["nsites"], case["nstates"], case["seed"], 0.15, case["p_gap"])
    tr = mbtree.random_tree(case["ntaxa"], case["tree_seed"], brlen=0.05)
    names = ["t%d" % (i + 1) for i in range(st.shape[0])]
    if case["nstates"] == 4:
        seqs, dt = ["".join("ACGT-"[x] for x in row) for row in st], "dna"
    else:
        seqs, dt = ["".join(_AA[x] if x < 20 else "-" for x in row) for row in st], "protein"
    s = "#NEXUS\nbegin data;\n  dimensions ntax=%d nchar=%d;\n  format datatype=%s interleave=no gap=- missing=?;\n  matrix\n" % (len(names), len(seqs[0]), dt)
    for n, q in zip(names, seqs):
        s += "%s  %s\n" % (n, q)
    s += "  ;\nend;\nbegin mrbayes;\n  set autoclose=yes nowarnings=yes seed=12345 swapseed=12345 precision=15;\n  lset parsmodel=yes;\nend;\n"
    s += "begin trees;\n  tree t = [&U] %s\nend;\nbegin mrbayes;\n  startvals tau=t;\n" % tr.to_newick(names)
    s += "  mcmc ngen=1 samplefreq=1 printfreq=1 diagnfreq=1 filename=pm;\nend;\n"       # (nruns / nchains left alone: the parsimony model trips over them)
    return s


def main():
    out = []
    for case in CASES:
        text, _ = refrun.run_mb(refrun.REF_MB, nexus(case))
        lnl = refrun.initial_lnl(text)
        out.append(dict(case, lnL_reference=lnl))
        print(case["name"], lnl)
    with open(os.path.join(ROOT, "tests", "golden", "parsmodel.json"), "w") as fh:
        json.dump({"source": "oracle/_ref/mb, lset parsmodel=yes, Chain 1 initial log-likelihood (tools/gen_golden_pars.py)", "cases": out}, fh, indent=1)


if __name__ == "__main__":
    main()
