"""TEST / BENCH INFRASTRUCTURE ONLY: time the REAL reference binary (oracle/_ref/mb, built from
/root/reference/src by oracle/Makefile) on a synthetic alignment + fixed tree, the way SURVEY §8(d) and
BASELINE.md §3 prescribe: two-point differencing of "Analysis used X seconds of CPU time" at two ngen
values with `prset topologypr=fixed(t) brlenspr=fixed(t)`, so that every generation is a parameter move
that re-evaluates the whole tree (TouchAllTreeNodes, reference src/proposal.c:17682) and parse / compress /
set-up time cancels.  Used by bench.py's cpu_baseline leg and by tools/gen_golden.py; never by the product.
"""
import os
import re
import subprocess
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MB = os.path.join(ROOT, "oracle", "_ref", "mb")

_NUC = "ACGT-"


def reference_available():
    return os.path.exists(REF_MB) and os.access(REF_MB, os.X_OK)


def _nexus(names, seqs, datatype, lset, prset, newick, ngen, fname):
    s = "#NEXUS\nbegin data;\n  dimensions ntax=%d nchar=%d;\n" % (len(names), len(seqs[0]))
    s += "  format datatype=%s interleave=no gap=- missing=?;\n  matrix\n" % datatype
    for n, q in zip(names, seqs):
        s += "%s  %s\n" % (n, q)
    s += "  ;\nend;\nbegin mrbayes;\n  set autoclose=yes nowarnings=yes seed=12345 swapseed=12345;\n  %s\nend;\n" % lset
    s += "begin trees;\n  tree t = [&U] %s\nend;\n" % newick
    s += "begin mrbayes;\n  %s\n  prset topologypr=fixed(t) brlenspr=fixed(t);\n" % (prset or "")
    s += "  mcmc ngen=%d nchains=1 nruns=1 samplefreq=%d printfreq=%d diagnfreq=%d filename=%s;\nend;\n" % (
        ngen, max(ngen, 1), max(ngen, 1), max(ngen, 1), fname)
    return s


def _run(text, workdir):
    with open(os.path.join(workdir, "run.nex"), "w") as fh:
        fh.write(text)
    t0 = time.time()
    res = subprocess.run([REF_MB, "run.nex"], cwd=workdir, capture_output=True, text=True, timeout=1800)
    wall = time.time() - t0
    m = re.search(r"Analysis used ([0-9.]+) seconds of CPU time", res.stdout)
    if not m:
        m2 = re.search(r"Analysis completed in ([0-9]+) seconds", res.stdout)
        if not m2:
            raise RuntimeError("reference run failed:\n" + res.stdout[-2000:])
        return float(m2.group(1)), wall, res.stdout
    return float(m.group(1)), wall, res.stdout


def time_reference_dna(states, tree, ngen_lo=10, ngen_hi=60, lset="lset nst=6 rates=gamma ngammacat=4;"):
    """states: int array [ntaxa][nsites] (0..3, 4 = gap).  Returns dict with seconds per full-tree
    evaluation (CPU time, one core) measured on the reference's FMA/AVX kernels."""
    names = ["t%d" % (i + 1) for i in range(states.shape[0])]
    seqs = ["".join(_NUC[x] for x in row) for row in states]
    newick = tree.to_newick(names)
    out = {}
    with tempfile.TemporaryDirectory() as wd:
        for tag, ngen in (("lo", ngen_lo), ("hi", ngen_hi)):
            cpu, wall, stdout = _run(_nexus(names, seqs, "dna", lset, "", newick, ngen, "bench"), wd)
            out[tag] = (ngen, cpu, wall)
            calc = re.search(r"Using standard (\S+) likelihood calculator", stdout)
            out["calculator"] = calc.group(1) if calc else "?"
            npat = re.search(r"has (\d+) unique site patterns", stdout)
            if npat:
                out["npatterns"] = int(npat.group(1))
    dgen = out["hi"][0] - out["lo"][0]
    dt = out["hi"][1] - out["lo"][1]
    out["sec_per_eval"] = dt / dgen
    return out
