"""The remaining data types MrBayes sends through its BEAGLE seam, end to end through the real binary: restriction sites (2 states:
CondLikeDown_Bin / Likelihood_Res, reference src/likelihood.c:78, :7134, with the three ascertainment codings -- the correction for the
unobservable patterns is MrBayes' own host code over the engine's site values, src/mbbeagle.c:1322-1358) and the RNA doublet model
(16 states: CondLikeDown_Gen with SetNucQMatrix's 16 x 16 matrix).  Both run on instantiations of the general-state tree walk.

Oracle: the reference's SCALAR build (oracle/_ref/mb_scalar) on the same start state -- every parameter fixed, the tree given -- to the
digits of the .p file.  (Its FMA/SSE build prints -225.79 for the doublet case where the scalar build and the engine print -783.61:
the SIMD kernels are not a usable oracle for that model.)"""
import os

import numpy as np
import pytest

from mrbayes_amd import tree as mbtree
from tools import refrun

REF = os.path.join(refrun.ROOT, "oracle", "_ref")
NT = 9


def _tree():
    return mbtree.random_tree(NT, 4, brlen=0.08), ["t%d" % (i + 1) for i in range(NT)]


def _restriction(coding, beagle, nsites=160):
    rng = np.random.default_rng(5)
    tr, names = _tree()
    base = rng.integers(0, 2, size=nsites)
    s = "#NEXUS\nbegin data;\n dimensions ntax=%d nchar=%d;\n format datatype=restriction gap=- missing=?;\n matrix\n" % (NT, nsites)
    for n in names:
        r = base.copy()
        flip = rng.random(nsites) < 0.25
        r[flip] = 1 - r[flip]
        s += "%s %s\n" % (n, "".join("?" if rng.random() < 0.03 else str(x) for x in r))
    s += ";\nend;\nbegin trees;\n tree t = [&U] %s\nend;\nbegin mrbayes;\n set autoclose=yes nowarnings=yes seed=3 swapseed=3 precision=15;\n" % tr.to_newick(names)
    s += " lset coding=%s rates=gamma ngammacat=4;\n prset statefreqpr=fixed(0.35,0.65) shapepr=fixed(0.7);\n" % coding
    if beagle:
        s += " set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=%s;\n" % beagle
    return s + " startvals tau=t V=t;\n mcmc ngen=1 nchains=1 nruns=1 samplefreq=1 printfreq=1 filename=x;\nend;\n"


def _doublet(beagle, npairs=60):
    rng = np.random.default_rng(6)
    tr, names = _tree()
    comp = {"A": "U", "U": "A", "C": "G", "G": "C"}
    stem = rng.choice(list("ACGU"), size=npairs)
    n = 2 * npairs
    s = "#NEXUS\nbegin data;\n dimensions ntax=%d nchar=%d;\n format datatype=rna gap=- missing=?;\n matrix\n" % (NT, n)
    for nm in names:
        a = stem.copy()
        mut = rng.random(npairs) < 0.2
        a[mut] = rng.choice(list("ACGU"), size=int(mut.sum()))
        b = np.array([comp[x] for x in a])
        mm = rng.random(npairs) < 0.1
        b[mm] = rng.choice(list("ACGU"), size=int(mm.sum()))
        s += "%s %s\n" % (nm, "".join(a) + "".join(b[::-1]))
    s += ";\nend;\nbegin trees;\n tree t = [&U] %s\nend;\nbegin mrbayes;\n set autoclose=yes nowarnings=yes seed=3 swapseed=3 precision=15;\n" % tr.to_newick(names)
    s += " pairs %s;\n lset nucmodel=doublet nst=6 rates=gamma ngammacat=4;\n" % ", ".join("%d:%d" % (i + 1, n - i) for i in range(npairs))
    s += " prset statefreqpr=fixed(equal) shapepr=fixed(0.7) revmatpr=fixed(1,3,1,1,3,1);\n"
    if beagle:
        s += " set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=%s;\n" % beagle
    return s + " startvals tau=t V=t;\n mcmc ngen=1 nchains=1 nruns=1 samplefreq=1 printfreq=1 filename=x;\nend;\n"


_MODELS = {
    "codon_equal": (61, " lset nucmodel=codon omegavar=equal nst=2;\n prset omegapr=fixed(0.3) tratiopr=fixed(2.0) statefreqpr=fixed(equal);\n"),
    "codon_ny98": (61, " lset nucmodel=codon omegavar=ny98 nst=2;\n prset ny98omega1pr=fixed(0.2) ny98omega3pr=fixed(2.5) codoncatfreqs=fixed(0.5,0.3,0.2)"
                       " tratiopr=fixed(2.0) statefreqpr=fixed(equal);\n"),
    "propinv": (4, " lset nst=6 rates=propinv;\n prset pinvarpr=fixed(0.3) revmatpr=fixed(1,2,1,1,2,1) statefreqpr=fixed(0.3,0.2,0.2,0.3);\n"),
    "lnorm": (4, " lset nst=2 rates=lnorm nlnormcat=5;\n prset tratiopr=fixed(2.5) statefreqpr=fixed(0.3,0.2,0.2,0.3);\n"),
    "hky_invgamma": (4, " lset nst=2 rates=invgamma ngammacat=4;\n prset tratiopr=fixed(2.5) shapepr=fixed(0.6) pinvarpr=fixed(0.2) statefreqpr=fixed(0.3,0.2,0.2,0.3);\n"),
    "jc": (4, " lset nst=1 rates=equal;\n prset statefreqpr=fixed(equal);\n"),
}


def _nucleotide_model(kind, beagle):
    """More of SetLikeFunctions' menu on DNA data: single-omega and NY98 codon models (61 states, one / three eigen-systems), +I without
    gamma, lognormal and invariable+gamma rate variation, HKY and JC closed-form matrices -- same harness, every parameter fixed."""
    from mrbayes_amd import data as mbdata
    nstates, model = _MODELS[kind]
    tr, names = _tree()
    if nstates == 61:
        st = mbdata.synthetic_states(NT, 50, 61, 4, 0.15, 0.02)
        seqs = ["".join(refrun._SENSE_CODONS[x] if x < 61 else "---" for x in row) for row in st]
    else:
        st = mbdata.synthetic_states(NT, 200, 4, 4, 0.15, 0.02)
        seqs = ["".join("ACGT"[x] if x < 4 else "-" for x in row) for row in st]
    s = "#NEXUS\nbegin data;\n dimensions ntax=%d nchar=%d;\n format datatype=dna gap=- missing=?;\n matrix\n" % (NT, len(seqs[0]))
    for nm, r in zip(names, seqs):
        s += "%s %s\n" % (nm, r)
    s += ";\nend;\nbegin trees;\n tree t = [&U] %s\nend;\nbegin mrbayes;\n set autoclose=yes nowarnings=yes seed=3 swapseed=3 precision=15;\n" % tr.to_newick(names)
    s += model
    if beagle:
        s += " set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=%s;\n" % beagle
    return s + " startvals tau=t V=t;\n mcmc ngen=1 nchains=1 nruns=1 samplefreq=1 printfreq=1 filename=x;\nend;\n"


CASES = {"restriction_all": lambda b: _restriction("all", b), "restriction_variable": lambda b: _restriction("variable", b),
         "restriction_noabsencesites": lambda b: _restriction("noabsencesites", b), "doublet": _doublet}
for _k in _MODELS:
    CASES[_k] = (lambda kind: (lambda b: _nucleotide_model(kind, b)))(_k)


def _lnl(binary, text):
    out, row = refrun.run_mb_with_samples(binary, text)
    assert "Analysis completed" in out, out[-2000:]
    return (row["LnL"] if "LnL" in row else row["lnLike"]), out


def _check(case, binary, marker):
    for b in (binary, os.path.join(REF, "mb_scalar")):
        if not os.path.exists(b):
            pytest.skip("reference binaries not built (oracle/Makefile)")
    want, _ = _lnl(os.path.join(REF, "mb_scalar"), CASES[case](None))
    for scaling in ("always", "dynamic"):
        got, out = _lnl(binary, CASES[case](scaling))
        assert marker in out and "tree-walk" in out, out[-1500:]           # (2 and 16 states have walk instantiations)
        assert abs(got - want) <= 2e-5 * abs(want), (case, scaling, got, want)


@pytest.mark.parametrize("case", sorted(CASES))
def test_restriction_and_doublet_on_emulated_engine(case):
    _check(case, os.path.join(REF, "mb_emu"), "mbamd")


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_restriction_and_doublet_on_mi355x(case):
    _check(case, os.path.join(REF, "mb_amd"), "mbamd HIP gfx950")
