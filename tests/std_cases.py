"""Standard (morphology) data cases for tests/test_std_dropin.py and tools/gen_std_fixture.py: NEXUS texts with every parameter fixed
and the tree given, so that generation 0 of `mcmc ngen=1` is a known-answer evaluation (the recipe of tools/gen_golden.py)."""
import numpy as np

from mrbayes_amd import tree as mbtree


def nchar_of(row):
    n, i = 0, 0
    while i < len(row):
        if row[i] in "{(":
            i = row.index("}" if row[i] == "{" else ")", i)
        n += 1
        i += 1
    return n


def _tail(beagle, ngen, moves=False):
    s = ""
    if beagle:
        s += " set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=%s;\n" % beagle
    return s + " startvals tau=t V=t;\n mcmc ngen=%d nchains=1 nruns=1 samplefreq=%d printfreq=%d filename=x;\nend;\n" % (ngen, 1 if ngen < 10 else 20, max(1, ngen))


def _clades(newick):
    import re
    out, stack = [], []
    for m in re.finditer(r"\(|\)|t\d+", newick):
        tok = m.group(0)
        if tok == "(":
            stack.append([])
        elif tok == ")":
            c = stack.pop()
            out.append(c)
            if stack:
                stack[-1].extend(c)
        elif stack:
            stack[-1].append(tok)
    return out


def synthetic_nexus(ntax, nchar, beagle, seed=11, coding="variable", rates="gamma", maxstates=5, ordered=(), p_missing=0.04, p_poly=0.0, ngen=1, alpha="fixed(0.7)",
                    ancstates=0, symdir=None, nbetacat=None, again=None):
    """`ancstates` > 0: that many hard constraints taken from the tree and `report ancstates=yes` (the .p file then carries the
    state probabilities of every character at the constrained nodes)."""
    # ntax x nchar characters with 2 ... maxstates states each (every state of a character occurs: MrBayes takes a character's
    # state count from the symbols it sees), a share of missing entries and of two-state polymorphisms.
    rng = np.random.default_rng(seed)
    tr = mbtree.random_tree(ntax, 4, brlen=0.08)
    names = ["t%d" % (i + 1) for i in range(ntax)]
    nst = rng.integers(2, maxstates + 1, size=nchar)
    M = np.zeros((ntax, nchar), dtype=int)
    for c in range(nchar):
        n = int(nst[c])
        col = np.full(ntax, rng.integers(0, n))
        flip = rng.random(ntax) < 0.35
        col[flip] = rng.integers(0, n, size=int(flip.sum()))
        where = rng.permutation(ntax)[:n]
        for s in range(n):
            if (col == s).sum() == 0:
                col[where[s]] = s
        for s in range(n):                       # (a forced state may have overwritten the only copy of another one)
            if (col == s).sum() == 0:
                col[(where[0] + 1 + s) % ntax] = s
        M[:, c] = col
    s = "#NEXUS\nbegin data;\n dimensions ntax=%d nchar=%d;\n format datatype=standard gap=- missing=?;\n matrix\n" % (ntax, nchar)
    for i, nm in enumerate(names):
        row = []
        for c in range(nchar):
            u = rng.random()
            if u < p_missing:
                row.append("?")
            elif u < p_missing + p_poly and nst[c] > 2:
                a, b = sorted(rng.choice(int(nst[c]), size=2, replace=False))
                row.append("(%d%d)" % (a, b))
            else:
                row.append(str(M[i, c]))
        s += "%s %s\n" % (nm, "".join(row))
    s += ";\nend;\nbegin trees;\n tree t = [&U] %s\nend;\nbegin mrbayes;\n set autoclose=yes nowarnings=yes seed=3 swapseed=3 precision=15;\n" % tr.to_newick(names)
    if ordered:
        s += " ctype ordered: %s;\n" % " ".join(str(x) for x in ordered)
    s += " lset coding=%s rates=%s%s;\n" % (coding, rates, " ngammacat=4" if rates == "gamma" else "")
    if rates == "gamma":
        s += " prset shapepr=%s;\n" % alpha
    if symdir:                                   # unequal state frequencies: a symmetric Dirichlet / beta prior, discretised for binary characters
        s += " prset symdirihyperpr=%s;\n" % symdir
    if nbetacat:
        s += " lset nbetacat=%d;\n" % nbetacat
    if ancstates:
        cl = [c for c in _clades(tr.to_newick(names)) if 2 <= len(c) <= max(2, ntax // 2)][:ancstates]
        for i, c in enumerate(cl):
            s += " constraint c%d = %s;\n" % (i + 1, " ".join(c))
        s += " prset topologypr=constraints(%s);\n report ancstates=yes;\n" % ",".join("c%d" % (i + 1) for i in range(len(cl)))
    if again:                                    # a SECOND analysis in the same session, under other settings (the binding's tables are per analysis)
        t = _tail(beagle, ngen)
        return s + t[:t.rindex("end;")] + " %s\n startvals tau=t V=t;\n mcmc ngen=%d nchains=1 nruns=1 samplefreq=1 printfreq=1 filename=y;\nend;\n" % (again, ngen)
    return s + _tail(beagle, ngen)


SYNTHETIC = {
    "mk_gamma_variable": dict(ntax=9, nchar=60),
    "mk_equal_all": dict(ntax=12, nchar=120, coding="all", rates="equal"),
    "ordered_polymorphic": dict(ntax=10, nchar=80, ordered=(3, 7, 11, 20), p_poly=0.03, maxstates=6),
    "informative": dict(ntax=9, nchar=60, coding="informative"),
    "binary_only": dict(ntax=14, nchar=90, maxstates=2),
    "ten_states": dict(ntax=24, nchar=70, maxstates=10, p_missing=0.02),
    # unequal state frequencies of binary characters: the beta categories of the prior are a mixture (numBetaCats = 5; and 3)
    "binary_symdir_fixed": dict(ntax=14, nchar=90, maxstates=2, symdir="fixed(1.0)"),
    "binary_symdir_exponential": dict(ntax=12, nchar=70, maxstates=2, symdir="exponential(1.0)", rates="equal", nbetacat=3, coding="all"),
}
# the same session runs a second analysis under other settings: other rate categories / another prior on the frequencies
SECOND_ANALYSIS = {
    "more_categories": dict(ntax=10, nchar=70, again="lset ngammacat=6; prset shapepr=fixed(0.25);"),
    "frequencies_become_unequal": dict(ntax=12, nchar=80, maxstates=2, again="prset symdirihyperpr=fixed(2.0);"),
}
ANCSTATES = {       # report ancstates=yes on a standard division (CondLikeUp_Std / PrintAncStates_Std on host arrays filled from the device)
    "anc_gamma": dict(ntax=12, nchar=80, ancstates=2),
    "anc_equal_ordered_rooted_like": dict(ntax=16, nchar=100, rates="equal", ordered=(2, 5, 9), maxstates=6, ancstates=3, coding="all"),
    "anc_binary_gamma": dict(ntax=20, nchar=120, maxstates=2, ancstates=3, coding="informative"),
}
BIG = dict(ntax=100, nchar=2000, maxstates=6, seed=5)      # 100 taxa x 2 000 characters, mixed 2-6 states, gamma-4

CYNMIX_CONFIGS = {"mk_gamma_variable": dict(coding="variable", rates="gamma"), "mk_equal_informative": dict(coding="informative", rates="equal")}


def cynmix_nexus(fix, beagle, coding="variable", rates="gamma", ngen=1):
    ntax = len(fix["names"])
    tr = mbtree.random_tree(ntax, 9, brlen=0.06)
    s = "#NEXUS\nbegin data;\n dimensions ntax=%d nchar=166;\n format datatype=standard gap=- missing=?;\n matrix\n" % ntax
    for nm, row in zip(fix["names"], fix["rows"]):
        s += "%s %s\n" % (nm, row)
    s += ";\nend;\nbegin trees;\n tree t = [&U] %s\nend;\nbegin mrbayes;\n set autoclose=yes nowarnings=yes seed=3 swapseed=3 precision=15;\n" % tr.to_newick(fix["names"])
    s += " lset coding=%s rates=%s%s;\n" % (coding, rates, " ngammacat=4" if rates == "gamma" else "")
    if rates == "gamma":
        s += " prset shapepr=fixed(0.55);\n"
    return s + _tail(beagle, ngen)


MCMC_SYMDIR = dict(ntax=12, nchar=120, maxstates=2, ngen=400, alpha="exponential(1.0)", symdir="exponential(1.0)")
HYM_CONFIGS = {"mk_gamma_variable": dict(coding="variable", rates="gamma")}


def hymfossil_nexus(fix, beagle, coding="variable", rates="gamma", ngen=1):
    """The morphology of examples/hymfossil.nex (353 characters, 2 ... 7 states, 44 of them ordered, some excluded) on a fixed tree."""
    ntax = len(fix["names"])
    tr = mbtree.random_tree(ntax, 13, brlen=0.05)
    s = "#NEXUS\nbegin data;\n dimensions ntax=%d nchar=%d;\n format datatype=standard gap=- missing=?;\n matrix\n" % (ntax, fix["nchar"])
    for nm, row in zip(fix["names"], fix["rows"]):
        s += "%s %s\n" % (nm, row)
    s += ";\nend;\nbegin trees;\n tree t = [&U] %s\nend;\nbegin mrbayes;\n set autoclose=yes nowarnings=yes seed=3 swapseed=3 precision=15;\n" % tr.to_newick(fix["names"])
    s += " ctype ordered: %s;\n exclude %s;\n" % (" ".join(str(x) for x in fix["ordered"]), " ".join(str(x) for x in fix["excluded"]))
    s += " lset coding=%s rates=%s%s;\n" % (coding, rates, " ngammacat=4" if rates == "gamma" else "")
    if rates == "gamma":
        s += " prset shapepr=fixed(0.8);\n"
    return s + _tail(beagle, ngen)
