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
REF_MB_FP64 = os.path.join(ROOT, "oracle", "_ref", "mb_fp64")   # the reference with CLFlt = double (src/bayes.h:110-112)
REF_MB_AMD = os.path.join(ROOT, "oracle", "_ref", "mb_amd")     # the unmodified reference linked to OUR libhmsbeagle.so
REF_MB_EMU = os.path.join(ROOT, "oracle", "_ref", "mb_emu")     # same objects, TEST-ONLY host-emulation engine

REF_MB_AMD_V3 = os.path.join(ROOT, "oracle", "_ref", "mb_amd_v3")   # ... with MrBayes' BEAGLE v3 code path compiled in
REF_MB_EMU_V3 = os.path.join(ROOT, "oracle", "_ref", "mb_emu_v3")
# the reference + our ABI + the device-parsimony binding (integration/mrbayes/, integration/mrbayes/patches/patch_pars.py)
REF_MB_AMD_PARS = os.path.join(ROOT, "oracle", "_ref", "mb_amd_pars")
REF_MB_EMU_PARS = os.path.join(ROOT, "oracle", "_ref", "mb_emu_pars")
# the reference + every binding: device parsimony, pattern compression, reports / covarion, device eigen-systems (oracle/Makefile: ref-amd-full)
REF_MB_AMD_FULL = os.path.join(ROOT, "oracle", "_ref", "mb_amd_full")
REF_MB_EMU_FULL = os.path.join(ROOT, "oracle", "_ref", "mb_emu_full")

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


def known_answer_nexus(states, tree, revmat, pi, alpha, beagle=None, ngen=1, fname="ka"):
    """NEXUS text of the known-answer recipe: data + fixed tree + startvals, GTR+G4.
    beagle: None (native kernels) or the `beaglescaling` value ("dynamic" / "always")."""
    names = ["t%d" % (i + 1) for i in range(states.shape[0])]
    seqs = ["".join(_NUC[x] for x in row) for row in states]
    s = "#NEXUS\nbegin data;\n  dimensions ntax=%d nchar=%d;\n" % (len(names), len(seqs[0]))
    s += "  format datatype=dna interleave=no gap=- missing=?;\n  matrix\n"
    for n, q in zip(names, seqs):
        s += "%s  %s\n" % (n, q)
    s += "  ;\nend;\nbegin mrbayes;\n  set autoclose=yes nowarnings=yes seed=12345 swapseed=12345 precision=15;\n"
    s += "  lset nst=6 rates=gamma ngammacat=4;\n"
    if beagle:
        s += "  set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=%s;\n" % beagle
    s += "end;\nbegin trees;\n  tree t = [&U] %s\nend;\n" % tree.to_newick(names)
    vec = lambda v: "(" + ",".join("%.15g" % x for x in v) + ")"
    s += "begin mrbayes;\n  startvals tau=t V=t Revmat=%s Pi=%s Alpha=(%.15g);\n" % (vec(revmat), vec(pi), alpha)
    s += "  mcmc ngen=%d nchains=1 nruns=1 samplefreq=%d printfreq=%d diagnfreq=%d filename=%s;\nend;\n" % (
        ngen, max(ngen, 1), max(ngen, 1), max(ngen, 1), fname)
    return s


def run_mb(binary, nexus_text, timeout=1800, env=None, keep=None, argv_prefix=None):
    """Run a MrBayes binary on a NEXUS text in a scratch directory -> (stdout, wall seconds); with `keep` (file names
    the run writes) -> (stdout, wall seconds, {name: text}).  `argv_prefix`: a launcher in front of the binary."""
    with tempfile.TemporaryDirectory() as wd:
        with open(os.path.join(wd, "run.nex"), "w") as fh:
            fh.write(nexus_text)
        t0 = time.time()
        res = subprocess.run(list(argv_prefix or []) + [binary, "run.nex"], cwd=wd, capture_output=True, text=True, timeout=timeout,
                             env=dict(os.environ, **(env or {})))
        wall = time.time() - t0
        if keep is None:
            return res.stdout + res.stderr, wall
        files = {}
        for name in keep:
            path = os.path.join(wd, name)
            if os.path.exists(path):
                with open(path) as fh:
                    files[name] = fh.read()
        return res.stdout + res.stderr, wall, files


def initial_lnl(stdout):
    m = re.search(r"Chain 1 -- (-?[0-9.]+) --", stdout)
    if not m:
        raise RuntimeError("no initial log-likelihood in the MrBayes output:\n" + stdout[-2000:])
    return float(m.group(1))


def time_reference(kind, states, tree, ngen_lo, ngen_hi):
    """Seconds per full-tree evaluation of the reference's own kernels for kind "gtr" (DNA GTR+G4), "wag" (protein, fixed
    WAG+G4) or "m3" (codon M3): fixed tree AND fixed branch lengths, so every generation is a substitution-parameter move
    that re-evaluates the whole tree (for M3 that includes the reference's host-side Q / eigen update, as in a real run)."""
    if kind == "gtr":
        return time_reference_dna(states, tree, ngen_lo, ngen_hi)
    names = ["t%d" % (i + 1) for i in range(states.shape[0])]
    out = {}
    with tempfile.TemporaryDirectory() as wd:
        for tag, ngen in (("lo", ngen_lo), ("hi", ngen_hi)):
            text = model_nexus(kind, states, tree, ngen=ngen, fname="bench", fixed_topology=True)
            text = text.replace("prset topologypr=fixed(t); startvals V=t", "prset topologypr=fixed(t) brlenspr=fixed(t)")
            cpu, wall, stdout = _run(text, wd)
            out[tag] = (ngen, cpu, wall)
            calc = re.search(r"Using standard (\S+) likelihood calculator", stdout)
            out["calculator"] = calc.group(1) if calc else "?"
            npat = re.search(r"has (\d+) unique site patterns", stdout)
            if npat:
                out["npatterns"] = int(npat.group(1))
    out["sec_per_eval"] = (out["hi"][1] - out["lo"][1]) / (out["hi"][0] - out["lo"][0])
    return out


def mcmc_nexus(states, tree_or_none, ngen, beagle=None, nchains=1, fname="mc", fixed_topology=False):
    """A default-move-mix MCMC run (GTR+G4) on the given alignment: what `MCMC gen/s` is quoted on."""
    names = ["t%d" % (i + 1) for i in range(states.shape[0])]
    seqs = ["".join(_NUC[x] for x in row) for row in states]
    s = "#NEXUS\nbegin data;\n  dimensions ntax=%d nchar=%d;\n" % (len(names), len(seqs[0]))
    s += "  format datatype=dna interleave=no gap=- missing=?;\n  matrix\n"
    for n, q in zip(names, seqs):
        s += "%s  %s\n" % (n, q)
    s += "  ;\nend;\nbegin mrbayes;\n  set autoclose=yes nowarnings=yes seed=12345 swapseed=12345;\n"
    s += "  lset nst=6 rates=gamma ngammacat=4;\n"
    if beagle:
        s += "  set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=%s;\n" % beagle
    s += "end;\n"
    if tree_or_none is not None:
        s += "begin trees;\n  tree t = [&U] %s\nend;\nbegin mrbayes;\n  %s;\nend;\n" % (
            tree_or_none.to_newick(names), "prset topologypr=fixed(t); startvals V=t" if fixed_topology else "startvals tau=t V=t")
    s += "begin mrbayes;\n  mcmc ngen=%d nchains=%d nruns=1 samplefreq=%d printfreq=%d diagnfreq=%d filename=%s;\nend;\n" % (
        ngen, nchains, max(ngen, 1), max(ngen, 1), max(ngen, 1), fname)
    return s


def analysis_seconds(stdout):
    m = re.search(r"Analysis used ([0-9.]+) seconds of CPU time", stdout)
    return float(m.group(1)) if m else None


_AA = "ARNDCQEGHILKMFPSTWYV"
_SENSE_CODONS = [a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT" if a + b + c not in ("TAA", "TAG", "TGA")]


def model_nexus(kind, states, tree, ngen=1, beagle=None, fname="mk", fixed_topology=False, precision="single"):
    """Known-answer / short-run NEXUS text for the general-state models of the hot path: kind "wag" (protein,
    fixed WAG + gamma 4) or "m3" (codon, omegavar=M3).  states: int array [ntaxa][nsites], value >= nstates = gap."""
    names = ["t%d" % (i + 1) for i in range(states.shape[0])]
    if kind == "wag":
        seqs = ["".join(_AA[x] if x < 20 else "-" for x in row) for row in states]
        datatype, lset = "protein", "prset aamodelpr=fixed(wag); lset rates=gamma ngammacat=4;"
    elif kind == "m3":
        seqs = ["".join(_SENSE_CODONS[x] if x < 61 else "---" for x in row) for row in states]
        datatype, lset = "dna", "lset nucmodel=codon omegavar=M3;"
    else:
        raise ValueError(kind)
    s = "#NEXUS\nbegin data;\n  dimensions ntax=%d nchar=%d;\n" % (len(names), len(seqs[0]))
    s += "  format datatype=%s interleave=no gap=- missing=?;\n  matrix\n" % datatype
    for n, q in zip(names, seqs):
        s += "%s  %s\n" % (n, q)
    s += "  ;\nend;\nbegin mrbayes;\n  set autoclose=yes nowarnings=yes seed=12345 swapseed=12345 precision=15;\n  %s\n" % lset
    if beagle:
        s += "  set usebeagle=yes beagledevice=gpu beagleprecision=%s beaglescaling=%s;\n" % (precision, beagle)
    s += "end;\nbegin trees;\n  tree t = [&U] %s\nend;\n" % tree.to_newick(names)
    s += "begin mrbayes;\n  %s;\n" % ("prset topologypr=fixed(t); startvals V=t" if fixed_topology else "startvals tau=t V=t")
    s += "  mcmc ngen=%d nchains=1 nruns=1 samplefreq=%d printfreq=%d diagnfreq=%d filename=%s;\nend;\n" % (
        ngen, max(ngen, 1), max(ngen, 1), max(ngen, 1), fname)
    return s


def run_mb_with_samples(binary, nexus_text, timeout=1800, env=None):
    """Like run_mb, but also returns the first sampled row of the run's .p file as a dict (column -> float):
    the generation-0 row holds the starting values MrBayes drew, e.g. the M3 omega-class frequencies."""
    with tempfile.TemporaryDirectory() as wd:
        with open(os.path.join(wd, "run.nex"), "w") as fh:
            fh.write(nexus_text)
        res = subprocess.run([binary, "run.nex"], cwd=wd, capture_output=True, text=True, timeout=timeout,
                             env=dict(os.environ, **(env or {})))
        row = {}
        for f in sorted(os.listdir(wd)):
            if f.endswith(".p"):
                with open(os.path.join(wd, f)) as fh:
                    lines = [l for l in fh.read().splitlines() if l and not l.startswith("[")]
                if len(lines) >= 2:
                    row = dict(zip(lines[0].split("\t"), (float(x) for x in lines[1].split("\t"))))
                break
        return res.stdout + res.stderr, row
