#!/usr/bin/env python3
"""Generate the golden log-likelihood fixtures under tests/golden/ by running the REAL reference
(oracle/_ref/mb, mb_scalar, mb_fp64 -- built from /root/reference/src by oracle/Makefile).

Known-answer recipe: data + a `trees` block + `startvals` that pin the tree, branch
lengths and every model parameter, then `mcmc ngen=1 nchains=1 nruns=1`; the generation-0 row of
the .p file (precision=15) is the reference's full-tree log-likelihood of exactly that state
(LogLike, reference src/mcmc.c:16306, printed by PrintStatesToFiles).

Runs only in the build container (needs oracle/_ref and, for the example data sets,
/root/reference/examples).  The fixtures it writes are committed:
   tests/golden/<case>.json   parameters, newick (taxon numbers), lnL per reference build
   tests/golden/<case>.npz    per-taxon state bit-sets of the compressed patterns + weights
                               (derived data; synthetic cases store only their generator seeds)
   tests/golden/aa_wag.json   WAG exchangeabilities/frequencies read out of the reference tables
"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrbayes_amd import data as mbdata            # noqa: E402
from mrbayes_amd import tree as mbtree            # noqa: E402
from mrbayes_amd.model import AA_ORDER, NUC, sense_codons  # noqa: E402

REF = os.environ.get("MB_REFERENCE", "/root/reference")
BINS = {"fma": "mb", "scalar": "mb_scalar", "fp64": "mb_fp64"}
GOLD = os.path.join(ROOT, "tests", "golden")


def run_mb(binary, nexus_text, workdir, outname):
    path = os.path.join(workdir, "run.nex")
    with open(path, "w") as fh:
        fh.write(nexus_text)
    exe = os.path.join(ROOT, "oracle", "_ref", binary)
    res = subprocess.run([exe, "run.nex"], cwd=workdir, capture_output=True, text=True, timeout=3600)
    out = res.stdout
    if "Error" in out and "Chain 1 --" not in out:
        raise RuntimeError(out[-3000:])
    m = re.search(r"Chain 1 -- (-?[0-9.]+) --", out)
    calc = re.search(r"Using standard (\S+) likelihood calculator[^\n]*\(([a-z]+)-precision\)", out)
    with open(os.path.join(workdir, outname + ".p")) as fh:
        lines = [l for l in fh.read().splitlines() if l and not l.startswith("[")]
    header = lines[0].split("\t")
    row0 = lines[1].split("\t")
    vals = dict(zip(header, row0))
    return {"screen_lnL": float(m.group(1)), "lnL": float(vals["lnLike"]), "params": vals,
            "calculator": calc.group(1) + "/" + calc.group(2) if calc else "?"}


def nexus_data(names, seqs, datatype):
    s = "#NEXUS\nbegin data;\n  dimensions ntax=%d nchar=%d;\n" % (len(names), len(seqs[0]))
    s += "  format datatype=%s interleave=no gap=- missing=?;\n  matrix\n" % datatype
    for n, q in zip(names, seqs):
        s += "%s  %s\n" % (n, q)
    s += "  ;\nend;\n"
    return s


def case_text(data_block, lset, prset, newick_named, startvals, outname):
    s = data_block
    s += "begin mrbayes;\n  set autoclose=yes nowarnings=yes seed=12345 swapseed=12345 precision=15;\n"
    s += "  %s\n" % lset
    if prset:
        s += "  %s\n" % prset
    s += "end;\nbegin trees;\n  tree t = [&U] %s\nend;\n" % newick_named
    s += "begin mrbayes;\n  startvals tau=t V=t %s;\n" % startvals
    s += "  mcmc ngen=1 nchains=1 nruns=1 samplefreq=1 printfreq=1 filename=%s;\nend;\n" % outname
    return s


def bits_array(pat):
    arr = np.zeros((pat.ntaxa, pat.npatterns), dtype=np.uint64)
    for t in range(pat.ntaxa):
        arr[t] = np.array(pat.bits[t], dtype=np.uint64)
    return arr


def states_to_seqs(states, datatype):
    if datatype == "dna":
        alpha = NUC + "-"
        return ["".join(alpha[x] for x in row) for row in states]
    if datatype == "protein":
        alpha = AA_ORDER + "-"
        return ["".join(alpha[x] for x in row) for row in states]
    nucs, _ = sense_codons()
    trip = ["".join(NUC[n] for n in c) for c in nucs] + ["---"]
    return ["".join(trip[x] for x in row) for row in states]


def fmt_vec(v):
    return "(" + ",".join("%.15g" % x for x in v) + ")"


def extract_wag():
    """Read the WAG exchangeabilities and frequencies out of the reference's tables
    (src/model.c:17332-17431) -- published numbers (Whelan & Goldman 2001), stored as a fixture."""
    src = open(os.path.join(REF, "src", "model.c")).read()
    s = np.zeros((20, 20))
    for i, j, v in re.findall(r"aaWAG\[\s*(\d+)\]\[\s*(\d+)\]\s*=\s*([0-9.eE+-]+);", src):
        s[int(i), int(j)] = float(v)
    pi = np.zeros(20)
    for i, v in re.findall(r"wagPi\[\s*(\d+)\]\s*=\s*([0-9.eE+-]+);", src):
        pi[int(i)] = float(v)
    assert np.allclose(s, s.T) and abs(pi.sum() - 1) < 1e-3
    with open(os.path.join(GOLD, "aa_wag.json"), "w") as fh:
        json.dump({"order": AA_ORDER, "exchangeability": s.tolist(), "pi": pi.tolist(),
                   "source": "reference src/model.c:17332-17431 (aaWAG, wagPi)"}, fh)


def emit(case, datatype, names, seqs, tr, lset, prset, startvals, model, store_patterns=True, synth=None):
    data_block = nexus_data(names, seqs, "dna" if datatype == "codon" else datatype)
    newick = tr.to_newick(names)
    result = {"case": case, "datatype": datatype, "model": model, "ntaxa": len(names),
              "newick": tr.to_newick(None), "lnL": {}, "calculator": {}}
    for key, binary in BINS.items():
        with tempfile.TemporaryDirectory() as wd:
            r = run_mb(binary, case_text(data_block, lset, prset, newick, startvals, "gold"), wd, "gold")
        result["lnL"][key] = r["lnL"]
        result["calculator"][key] = r["calculator"]
        if key == "fma":
            result["reference_params_row0"] = {k: v for k, v in r["params"].items()
                                               if k not in ("Gen", "lnLike", "lnPrior")}
        print("  %-8s %-10s lnL = %.9f" % (case, key, r["lnL"]), flush=True)
    if synth is not None:
        result["synthetic"] = synth
    pat = mbdata.patterns_from_sequences(seqs, datatype, names)
    result["npatterns"] = pat.npatterns
    result["nsites"] = float(pat.weights.sum())
    if store_patterns:
        np.savez_compressed(os.path.join(GOLD, case + ".npz"), bits=bits_array(pat), weights=pat.weights)
    with open(os.path.join(GOLD, case + ".json"), "w") as fh:
        json.dump(result, fh, indent=1)


def main():
    os.makedirs(GOLD, exist_ok=True)
    only = set(sys.argv[1:])
    extract_wag()

    def want(c):
        return not only or c in only

    ex = os.path.join(REF, "examples")
    # ---- primates: GTR+G4 and GTR+I+G4, explicit tree/params -------------------------------
    names, seqs, _ = mbdata.read_nexus_matrix(os.path.join(ex, "primates.nex"))
    tr = mbtree.random_tree(len(names), seed=11, brlen=None, brlen_mean=0.06)
    rev = [0.10, 0.30, 0.05, 0.08, 0.40, 0.07]
    pi = [0.35, 0.25, 0.15, 0.25]
    if want("primates_gtr_g4"):
        emit("primates_gtr_g4", "dna", names, seqs, tr, "lset nst=6 rates=gamma ngammacat=4;", "",
             "Revmat=%s Pi=%s Alpha=(0.6)" % (fmt_vec(rev), fmt_vec(pi)),
             {"kind": "gtr", "revmat": rev, "pi": pi, "alpha": 0.6, "ncat": 4, "pinvar": 0.0})
    if want("primates_gtr_ig4"):
        emit("primates_gtr_ig4", "dna", names, seqs, tr, "lset nst=6 rates=invgamma ngammacat=4;", "",
             "Revmat=%s Pi=%s Alpha=(0.45) Pinvar=(0.2)" % (fmt_vec(rev), fmt_vec(pi)),
             {"kind": "gtr", "revmat": rev, "pi": pi, "alpha": 0.45, "ncat": 4, "pinvar": 0.2})
    if want("primates_gtr_equal"):
        emit("primates_gtr_equal", "dna", names, seqs, tr, "lset nst=6 rates=equal;", "",
             "Revmat=%s Pi=%s" % (fmt_vec(rev), fmt_vec(pi)),
             {"kind": "gtr", "revmat": rev, "pi": pi, "alpha": None, "ncat": 1, "pinvar": 0.0})

    # ---- avian ovomucoids: WAG+G4 ----------------------------------------------------------
    if want("avian_wag_g4"):
        names, seqs, dt = mbdata.read_nexus_matrix(os.path.join(ex, "avian_ovomucoids.nex"))
        tr = mbtree.random_tree(len(names), seed=12, brlen=None, brlen_mean=0.04)
        emit("avian_wag_g4", "protein", names, seqs, tr, "lset rates=gamma ngammacat=4;",
             "prset aamodelpr=fixed(wag);", "Alpha=(0.8)",
             {"kind": "wag", "alpha": 0.8, "ncat": 4, "pinvar": 0.0})

    # ---- replicase: codon M3 ---------------------------------------------------------------
    if want("replicase_m3"):
        names, seqs, dt = mbdata.read_nexus_matrix(os.path.join(ex, "replicase.nex"))
        tr = mbtree.random_tree(len(names), seed=13, brlen=None, brlen_mean=0.05)
        emit("replicase_m3", "codon", names, seqs, tr, "lset nucmodel=codon omegavar=M3;",
             "prset statefreqpr=fixed(equal);", "",
             {"kind": "m3", "nst": 1, "omega": None, "omega_freq": None, "pi": "equal"})

    # ---- synthetic DNA with 5 % gaps (ambiguous / missing tip path), small ------------------
    if want("synth_dna_gaps"):
        st = mbdata.synthetic_states(40, 3000, 4, seed=21, p_mut=0.15, p_gap=0.05)
        names = ["t%d" % (i + 1) for i in range(40)]
        tr = mbtree.random_tree(40, seed=22, brlen=None, brlen_mean=0.05)
        emit("synth_dna_gaps", "dna", names, states_to_seqs(st, "dna"), tr,
             "lset nst=6 rates=gamma ngammacat=4;", "",
             "Revmat=%s Pi=%s Alpha=(1.3)" % (fmt_vec(rev), fmt_vec(pi)),
             {"kind": "gtr", "revmat": rev, "pi": pi, "alpha": 1.3, "ncat": 4, "pinvar": 0.0},
             store_patterns=False,
             synth={"ntaxa": 40, "nsites": 3000, "nstates": 4, "seed": 21, "p_mut": 0.15, "p_gap": 0.05,
                    "tree_seed": 22, "brlen_mean": 0.05})

    # ---- synthetic DNA 500 x 20 000 (BASELINE config 2), constant 0.05 branches -------------
    if want("synth_dna_500x20k"):
        st = mbdata.synthetic_states(500, 20000, 4, seed=7, p_mut=0.15)
        names = ["t%d" % (i + 1) for i in range(500)]
        tr = mbtree.random_tree(500, seed=3, brlen=0.05)
        emit("synth_dna_500x20k", "dna", names, states_to_seqs(st, "dna"), tr,
             "lset nst=6 rates=gamma ngammacat=4;", "",
             "Revmat=%s Pi=%s Alpha=(1.0)" % (fmt_vec([1 / 6.0] * 6), fmt_vec([0.25] * 4)),
             {"kind": "gtr", "revmat": [1 / 6.0] * 6, "pi": [0.25] * 4, "alpha": 1.0, "ncat": 4, "pinvar": 0.0},
             store_patterns=False,
             synth={"ntaxa": 500, "nsites": 20000, "nstates": 4, "seed": 7, "p_mut": 0.15, "p_gap": 0.0,
                    "tree_seed": 3, "brlen": 0.05})

    # ---- synthetic AA 60 x 1500 WAG+G4 and codon 30 x 600 M3 (smaller cousins of configs 3, 5)
    if want("synth_aa_wag"):
        st = mbdata.synthetic_states(60, 1500, 20, seed=5, p_mut=0.15)
        names = ["t%d" % (i + 1) for i in range(60)]
        tr = mbtree.random_tree(60, seed=9, brlen=0.05)
        emit("synth_aa_wag", "protein", names, states_to_seqs(st, "protein"), tr,
             "lset rates=gamma ngammacat=4;", "prset aamodelpr=fixed(wag);", "Alpha=(1.0)",
             {"kind": "wag", "alpha": 1.0, "ncat": 4, "pinvar": 0.0}, store_patterns=False,
             synth={"ntaxa": 60, "nsites": 1500, "nstates": 20, "seed": 5, "p_mut": 0.15, "p_gap": 0.0,
                    "tree_seed": 9, "brlen": 0.05})
    if want("synth_codon_m3"):
        st = mbdata.synthetic_states(30, 600, 61, seed=6, p_mut=0.15)
        names = ["t%d" % (i + 1) for i in range(30)]
        tr = mbtree.random_tree(30, seed=10, brlen=0.05)
        emit("synth_codon_m3", "codon", names, states_to_seqs(st, "codon"), tr,
             "lset nucmodel=codon omegavar=M3;", "prset statefreqpr=fixed(equal);", "",
             {"kind": "m3", "nst": 1, "omega": None, "omega_freq": None, "pi": "equal"}, store_patterns=False,
             synth={"ntaxa": 30, "nsites": 600, "nstates": 61, "seed": 6, "p_mut": 0.15, "p_gap": 0.0,
                    "tree_seed": 10, "brlen": 0.05})

    # ---- the bench.py workloads themselves (BASELINE configs 2-5 at full size): bench.py's printed lnL is pinned
    # against these reference values.  Same generator seeds / parameters as bench.CONFIGS + synthetic_division.
    def bench_case(case, kind, ntaxa, nsites, seed, tree_seed, p_gap=0.0, pinvar=0.0):
        nstates = {"gtr": 4, "wag": 20, "m3": 61}[kind]
        st = mbdata.synthetic_states(ntaxa, nsites, nstates, seed=seed, p_mut=0.15, p_gap=p_gap)
        names = ["t%d" % (i + 1) for i in range(ntaxa)]
        tr = mbtree.random_tree(ntaxa, seed=tree_seed, brlen=0.05)
        synth = {"ntaxa": ntaxa, "nsites": nsites, "nstates": nstates, "seed": seed, "p_mut": 0.15, "p_gap": p_gap,
                 "tree_seed": tree_seed, "brlen": 0.05}
        if kind == "gtr":
            rates = "invgamma" if pinvar > 0 else "gamma"
            emit(case, "dna", names, states_to_seqs(st, "dna"), tr, "lset nst=6 rates=%s ngammacat=4;" % rates, "",
                 "Revmat=%s Pi=%s Alpha=(1.0)%s" % (fmt_vec(rev), fmt_vec(pi), " Pinvar=(%.15g)" % pinvar if pinvar > 0 else ""),
                 {"kind": "gtr", "revmat": rev, "pi": pi, "alpha": 1.0, "ncat": 4, "pinvar": pinvar},
                 store_patterns=False, synth=synth)
        elif kind == "wag":
            emit(case, "protein", names, states_to_seqs(st, "protein"), tr, "lset rates=gamma ngammacat=4;",
                 "prset aamodelpr=fixed(wag);", "Alpha=(1.0)", {"kind": "wag", "alpha": 1.0, "ncat": 4, "pinvar": 0.0},
                 store_patterns=False, synth=synth)
        else:
            emit(case, "codon", names, states_to_seqs(st, "codon"), tr, "lset nucmodel=codon omegavar=M3;",
                 "prset statefreqpr=fixed(equal);", "",
                 {"kind": "m3", "nst": 1, "omega": None, "omega_freq": None, "pi": "equal"}, store_patterns=False,
                 synth=synth)
    for case, cfg in (("bench_c2", ("gtr", 500, 20000, 7, 3)), ("bench_c3", ("wag", 200, 10000, 5, 9)),
                      ("bench_c5", ("m3", 100, 5000, 6, 10)), ("bench_c4", ("gtr", 1000, 50000, 6, 10)),
                      # the bench shapes with 5 % missing data (the missing-state tip path at full size), and
                      # configs[1]'s shape with invariable sites on top (the +I path: per-site values read back every evaluation)
                      ("bench_c2_gaps", ("gtr", 500, 20000, 7, 3, 0.05)), ("bench_c3_gaps", ("wag", 200, 10000, 5, 9, 0.05)),
                      ("bench_c2_ig_gaps", ("gtr", 500, 20000, 7, 3, 0.05, 0.2))):
        if case in only:                 # (minutes of reference CPU time each: only on request)
            bench_case(case, *cfg)


if __name__ == "__main__":
    main()
