"""Standard (morphology) data on the engine -- the *_Std family of the reference (CondLikeDown_Std src/likelihood.c:1920, CondLikeRoot_Std
:4496, CondLikeScaler_Std :5547, Likelihood_Std :7359, TiProbs_Std :10066; selected at src/mcmc.c:18296-18306 and never handed to
BEAGLE, src/mcmc.c:5741-5771: the characters of a division have different state counts).  The binding
(integration/mrbayes/mbamd_std_glue.c, patches/patch_std.py; oracle/Makefile: ref-amd-std) gives every transition-matrix class of the
division its own engine instance; the matrices stay the reference's own TiProbs_Std.

Oracle: the reference's SCALAR build (oracle/_ref/mb_scalar) on the same start state -- every parameter fixed, the tree given -- to
the digits of the .p file: Mk with and without gamma, the three codings, ordered characters, polymorphic entries, 2 ... 10 states; the
morphology division of the reference's own examples/cynmix.nex (fixture tests/golden/std_cynmix.json, tools/gen_std_fixture.py); at
100 taxa x 2 000 characters; and a default-move MCMC run whose sampled log-likelihoods must be the native chain's."""
import json
import os
import subprocess
import tempfile

import pytest

from tests import std_cases
from tools import refrun

REF = os.path.join(refrun.ROOT, "oracle", "_ref")
GOLD = os.path.join(refrun.ROOT, "tests", "golden", "std_cynmix.json")


def _need(*bins):
    for b in bins:
        if not os.path.exists(b):
            pytest.skip("reference binaries not built (oracle/Makefile)")


def _build_emu():
    """CPU tests: the host-emulation engine and the binaries linked to it are built on demand (build container only)."""
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference sources not present (build container only)")
    from tests.hostemu import build_emu
    build_emu.build()
    subprocess.check_call(["make", "-C", os.path.join(refrun.ROOT, "oracle"), "_ref/mb_scalar", "_ref/mb_emu_std"], stdout=subprocess.DEVNULL)


def _lnl(binary, text, env=None):
    out, row = refrun.run_mb_with_samples(binary, text, env=env)
    assert "Analysis completed" in out, out[-2000:]
    return (row["LnL"] if "LnL" in row else row["lnLike"]), out


def _served(out, marker):
    assert "(standard data):" in out and "transition-matrix classes on" in out and marker in out, out[-1500:]


def _check_synthetic(case, binary, marker, scalings=("always",)):
    _need(binary, os.path.join(REF, "mb_scalar"))
    kw = std_cases.SYNTHETIC[case]
    want, _ = _lnl(os.path.join(REF, "mb_scalar"), std_cases.synthetic_nexus(beagle=None, **kw))
    for scaling in scalings:
        got, out = _lnl(binary, std_cases.synthetic_nexus(beagle=scaling, **kw))
        _served(out, marker)
        assert abs(got - want) <= 1e-5 * abs(want), (case, scaling, got, want)
    # the switch: MBAMD_DEVICE_STD=0 leaves the division on the reference's own kernels (and prints the same number)
    off, out = _lnl(binary, std_cases.synthetic_nexus(beagle="always", **kw), env={"MBAMD_DEVICE_STD": "0"})
    assert "(standard data):" not in out and abs(off - want) <= 1e-5 * abs(want)


def _check_cynmix(binary, marker):
    _need(binary, os.path.join(REF, "mb_scalar"))
    with open(GOLD) as fh:
        fix = json.load(fh)
    for key, kw in std_cases.CYNMIX_CONFIGS.items():
        got, out = _lnl(binary, std_cases.cynmix_nexus(fix, "always", **kw))
        _served(out, marker)
        want = fix["lnL_mb_scalar"][key]
        assert abs(got - want) <= 1e-5 * abs(want), (key, got, want)
        live, _ = _lnl(os.path.join(REF, "mb_scalar"), std_cases.cynmix_nexus(fix, None, **kw))
        assert abs(live - want) <= 1e-9 * abs(want)            # (the fixture is what the reference prints here, too)


def _check_hymfossil(binary, marker):
    """The morphology of the reference's examples/hymfossil.nex (fixture tests/golden/std_hymfossil.json, tools/gen_std_fixture.py:
    107 taxa x 353 characters of 2 ... 7 states, 44 ordered, 10 excluded) -- nine transition-matrix classes."""
    _need(binary, os.path.join(REF, "mb_scalar"))
    with open(os.path.join(refrun.ROOT, "tests", "golden", "std_hymfossil.json")) as fh:
        fix = json.load(fh)
    for key, kw in std_cases.HYM_CONFIGS.items():
        for scaling in ("always", "dynamic"):
            got, out = _lnl(binary, std_cases.hymfossil_nexus(fix, scaling, **kw))
            _served(out, marker)
            assert "9 transition-matrix classes" in out, out[-1500:]
            want = fix["lnL_mb_scalar"][key]
            assert abs(got - want) <= 1e-5 * abs(want), (key, scaling, got, want)


def _check_second_analysis(case, binary, marker):
    """Two analyses in one MrBayes session, the second under other settings: the binding's instances and class tables go with the
    reference's chain memory (FreeChainMemory) and are rebuilt -- both results are the scalar build's."""
    _need(binary, os.path.join(REF, "mb_scalar"))
    kw = std_cases.SECOND_ANALYSIS[case]

    def both(b, beagle, env=None):
        out, _, files = refrun.run_mb(b, std_cases.synthetic_nexus(beagle=beagle, **kw), keep=("x.p", "y.p"), env=env)
        assert out.count("Analysis completed") == 2 and "x.p" in files and "y.p" in files, out[-2500:]
        vals = []
        for f in ("x.p", "y.p"):
            rows = [l.split("\t") for l in files[f].splitlines() if l and not l.startswith("[")]
            i = rows[0].index("LnL") if "LnL" in rows[0] else rows[0].index("lnLike")
            vals.append(float(rows[1][i]))
        return vals, out
    want, _ = both(os.path.join(REF, "mb_scalar"), None)
    got, out = both(binary, "always")
    assert out.count("(standard data):") == 2 and marker in out, out[-2500:]        # set up twice: once per analysis
    assert abs(want[0] - want[1]) > 1e-3 * abs(want[0])                              # (the second analysis IS another model)
    for g, w in zip(got, want):
        assert abs(g - w) <= 1e-5 * abs(w), (case, got, want)


def _p_row(binary, text, env=None):
    """header and first sample row of the run's .p file"""
    out, _, files = refrun.run_mb(binary, text, keep=("x.p",), env=env)
    assert "Analysis completed" in out and "x.p" in files, out[-2000:]
    lines = [l for l in files["x.p"].splitlines() if l and not l.startswith("[")]
    return out, lines[0].split("\t"), [float(x) for x in lines[1].split("\t")]


def _check_ancstates(case, binary, marker):
    """report ancstates=yes: every reported state probability of the constrained nodes against the reference's scalar build."""
    _need(binary, os.path.join(REF, "mb_scalar"))
    kw = std_cases.ANCSTATES[case]
    _, h0, r0 = _p_row(os.path.join(REF, "mb_scalar"), std_cases.synthetic_nexus(beagle=None, **kw))
    probs = [j for j, name in enumerate(h0) if name.startswith("p(")]
    assert len(probs) > 20, h0[:30]
    for scaling in ("always", "dynamic"):
        out, h1, r1 = _p_row(binary, std_cases.synthetic_nexus(beagle=scaling, **kw))
        _served(out, marker)
        assert h1 == h0
        i = h0.index("LnL") if "LnL" in h0 else h0.index("lnLike")
        assert abs(r1[i] - r0[i]) <= 1e-5 * abs(r0[i]), (case, scaling, r1[i], r0[i])
        worst = max(abs(r1[j] - r0[j]) for j in probs)
        assert worst <= 1e-5, (case, scaling, worst, [(h0[j], r0[j], r1[j]) for j in probs if abs(r1[j] - r0[j]) > 1e-5][:5])
        assert all(0.0 <= r1[j] <= 1.0 for j in probs)


def _samples(binary, text, env=None):
    with tempfile.TemporaryDirectory() as wd:
        with open(os.path.join(wd, "run.nex"), "w") as fh:
            fh.write(text)
        res = subprocess.run([binary, "run.nex"], cwd=wd, capture_output=True, text=True, timeout=1800, env=dict(os.environ, **(env or {})))
        assert "Analysis completed" in res.stdout, (res.stdout + res.stderr)[-2000:]
        with open(os.path.join(wd, "x.p")) as fh:
            lines = [l for l in fh.read().splitlines() if l and not l.startswith("[")]
        cols = lines[0].split("\t")
        return [dict(zip(cols, (float(x) for x in l.split("\t")))) for l in lines[1:]], res.stdout


def _check_mcmc(binary, marker, kw=None):
    """Default moves (topology, branch lengths, the gamma shape) for 400 generations: accepted and rejected proposals flip the
    reference's index tables, which name the engine's buffers -- the sampled log-likelihoods are the native chain's.
    With kw: binary characters under symdirihyperpr=exponential(1): the hyperparameter moves, the beta categories' frequencies with it."""
    _need(binary)
    kw = kw or dict(ntax=12, nchar=150, maxstates=5, p_poly=0.02, ngen=400, alpha="exponential(1.0)")
    # (the oracle is the SAME binary with the binding switched off: a BEAGLE build of the reference consumes random numbers the
    #  plain build does not, so mb_scalar walks another chain; generation 0 is pinned against mb_scalar by the other tests)
    want, wout = _samples(binary, std_cases.synthetic_nexus(beagle="always", **kw), env={"MBAMD_DEVICE_STD": "0"})
    assert "(standard data):" not in wout
    got, out = _samples(binary, std_cases.synthetic_nexus(beagle="always", **kw))
    _served(out, marker)
    assert len(got) == len(want) >= 20
    for a, b in zip(got, want):
        ka = "LnL" if "LnL" in a else "lnLike"
        assert a["Gen"] == b["Gen"] and abs(a[ka] - b[ka]) <= 1e-5 * abs(b[ka]) and abs(a["TL"] - b["TL"]) <= 1e-6 * b["TL"], (a, b)


@pytest.mark.parametrize("case", sorted(std_cases.SYNTHETIC))
def test_standard_data_on_emulated_engine(case):
    _build_emu()
    _check_synthetic(case, os.path.join(REF, "mb_emu_std"), "HOST EMULATION")


def test_cynmix_morphology_on_emulated_engine():
    _build_emu()
    _check_cynmix(os.path.join(REF, "mb_emu_std"), "HOST EMULATION")


@pytest.mark.parametrize("case", sorted(std_cases.ANCSTATES))
def test_standard_data_ancestral_states_on_emulated_engine(case):
    """ancestral states of a standard-data division that lives on the engine."""
    _build_emu()
    _check_ancstates(case, os.path.join(REF, "mb_emu_std"), "HOST EMULATION")


def test_standard_data_mcmc_on_emulated_engine():
    _build_emu()
    _check_mcmc(os.path.join(REF, "mb_emu_std"), "HOST EMULATION")


def test_standard_data_mcmc_with_moving_frequencies_on_emulated_engine():
    _build_emu()
    _check_mcmc(os.path.join(REF, "mb_emu_std"), "HOST EMULATION", kw=std_cases.MCMC_SYMDIR)


def test_hymfossil_morphology_on_emulated_engine():
    _build_emu()
    _check_hymfossil(os.path.join(REF, "mb_emu_std"), "HOST EMULATION")


@pytest.mark.parametrize("case", sorted(std_cases.SECOND_ANALYSIS))
def test_second_analysis_in_a_session_on_emulated_engine(case):
    """the binding's state was decided once per process."""
    _build_emu()
    _check_second_analysis(case, os.path.join(REF, "mb_emu_std"), "HOST EMULATION")


def test_cynmix_as_shipped_all_partitions_on_the_emulated_engine():
    """The reference's own examples/cynmix.nex with the model of its manual (morphology Mk+G, four DNA partitions GTR+I+G, unlinked,
    rate multipliers): the morphology partition through the standard-data binding, the DNA partitions through the unmodified BEAGLE
    path, one short chain -- against the same binary with every division on the reference's own kernels (usebeagle=no)."""
    _build_emu()
    src = "/root/reference/examples/cynmix.nex"
    if not os.path.exists(src):
        pytest.skip("reference examples not present (build container only)")
    with open(src) as fh:
        text = fh.read()
    with open(GOLD) as fh:
        names = json.load(fh)["names"]
    from mrbayes_amd import tree as mbtree
    start = mbtree.random_tree(len(names), 9, brlen=0.06).to_newick(names)      # (MrBayes draws a different random start tree with and without BEAGLE)
    block = ("begin trees;\n tree t = [&U] " + start + "\nend;\n"
             "begin mrbayes;\n set autoclose=yes nowarnings=yes seed=7 swapseed=7 precision=12;\n set partition=favored;\n"
             " lset applyto=(1) rates=gamma;\n lset applyto=(2,3,4,5) rates=invgamma nst=6;\n"
             " unlink revmat=(all) pinvar=(all) shape=(all) statefreq=(all);\n prset applyto=(all) ratepr=variable;\n%s"
             " startvals tau=t V=t;\n mcmc ngen=60 nchains=1 nruns=1 samplefreq=10 printfreq=60 filename=x;\nend;\n")
    on = text + block % " set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=always;\n"
    off = text + block % " set usebeagle=no;\n"
    binary = os.path.join(REF, "mb_emu_std")
    out, _, files = refrun.run_mb(binary, on, keep=("x.p",))
    assert "Analysis completed" in out and "(standard data):" in out, out[-2500:]
    assert out.count("Using BEAGLE") >= 4, out[-2500:]                    # the four DNA partitions
    out0, _, files0 = refrun.run_mb(binary, off, keep=("x.p",))
    assert "Analysis completed" in out0 and "(standard data):" not in out0

    def lnls(t):
        rows = [l.split("\t") for l in t.splitlines() if l and not l.startswith("[")]
        i = rows[0].index("LnL") if "LnL" in rows[0] else rows[0].index("lnLike")
        return [float(r[i]) for r in rows[1:]]
    a, b = lnls(files["x.p"]), lnls(files0["x.p"])
    assert len(a) == len(b) >= 5
    assert abs(a[0] - b[0]) <= 2e-6 * abs(b[0]), (a[0], b[0])            # the start state: a known-answer evaluation of all five partitions
    # the chain: against the same binary with only the standard-data binding off (a BEAGLE run draws its random numbers in a different
    # order than a run without: the usebeagle=no chain is another chain from the first move on)
    out1, _, files1 = refrun.run_mb(binary, on, keep=("x.p",), env={"MBAMD_DEVICE_STD": "0"})
    assert "Analysis completed" in out1 and "(standard data):" not in out1
    c = lnls(files1["x.p"])
    assert len(c) == len(a)
    assert all(abs(x - y) <= 1e-5 * abs(y) for x, y in zip(a, c)), (a, c)


def test_hymfossil_as_shipped_all_partitions_on_the_emulated_engine():
    """The reference's examples/hymfossil.nex with ITS OWN model block (114 taxa; a morphology partition with ordered characters,
    excluded characters and coding=variable, six DNA partitions GTR+G, unlinked, rate multipliers; the block is commented out in
    the shipped file and ends in a 100-million-generation run: here it is executed up to that `mcmcp` line): start state against the
    native kernels, a short chain against the same binary with only the standard-data binding off."""
    _build_emu()
    src = "/root/reference/examples/hymfossil.nex"
    if not os.path.exists(src):
        pytest.skip("reference examples not present (build container only)")
    with open(src) as fh:
        text = fh.read()
    i = text.index("[\nbegin mrbayes;")
    data, blk = text[:i], text[i + 2:]
    blk = blk[:blk.index("    mcmcp nruns=4")]
    names = []
    for line in data[data.lower().index("matrix"):].splitlines()[1:]:
        t = line.split()
        if not t or t[0].startswith("["):
            continue
        if t[0] in names:
            break
        names.append(t[0])
    assert len(names) == 114
    from mrbayes_amd import tree as mbtree
    start = mbtree.random_tree(len(names), 9, brlen=0.05).to_newick(names)

    def nexus(beagle):
        tail = (" set autoclose=yes nowarnings=yes seed=7 swapseed=7 precision=12;\n" +
                (" set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=always;\n" if beagle else " set usebeagle=no;\n") +
                " startvals tau=t V=t;\n    mcmc ngen=20 nchains=1 nruns=1 samplefreq=5 printfreq=20 filename=x;\nend;\n")
        return data + "begin trees;\n tree t = [&U] " + start + "\nend;\n" + blk + tail

    def lnls(t):
        rows = [l.split("\t") for l in t.splitlines() if l and not l.startswith("[")]
        k = rows[0].index("LnL") if "LnL" in rows[0] else rows[0].index("lnLike")
        return [float(r[k]) for r in rows[1:]]
    binary = os.path.join(REF, "mb_emu_std")
    out, _, files = refrun.run_mb(binary, nexus(True), keep=("x.p",))
    assert "Analysis completed" in out and "(standard data): 9 transition-matrix classes" in out, out[-2500:]
    assert out.count("Using BEAGLE") == 6, out[-2500:]
    out0, _, files0 = refrun.run_mb(binary, nexus(False), keep=("x.p",))
    out1, _, files1 = refrun.run_mb(binary, nexus(True), keep=("x.p",), env={"MBAMD_DEVICE_STD": "0"})
    a, b, c = lnls(files["x.p"]), lnls(files0["x.p"]), lnls(files1["x.p"])
    assert len(a) == len(c) >= 4
    assert abs(a[0] - b[0]) <= 2e-6 * abs(b[0]), (a[0], b[0])
    assert all(abs(x - y) <= 1e-5 * abs(y) for x, y in zip(a, c)), (a, c)


def test_unequal_frequencies_of_multistate_characters_stay_on_the_host():
    """symdirihyperpr other than fixed(infinity) with characters of more than two states (an eigen-system per character): refused
    with a printed reason, the reference's own kernels run.  (Binary characters under such a prior ARE served: the
    binary_symdir_* cases.)"""
    _build_emu()
    b = os.path.join(REF, "mb_emu_std")
    kw = std_cases.SYNTHETIC["mk_gamma_variable"]
    text = std_cases.synthetic_nexus(beagle="always", **kw).replace(" lset coding", " prset symdirihyperpr=fixed(1.0);\n lset coding")
    out, _ = refrun.run_mb_with_samples(b, text)
    assert "stays on the host kernels" in out and "Analysis completed" in out, out[-1500:]


def _check_beta_category_limit(binary, marker):
    """`lset nbetacat` beyond what one integration of the engine mixes (eight buffer sets): the division stays on the reference's
    kernels with a printed reason (advisor, round 5: it used to be accepted and ended in Die() at the first evaluation); eight
    categories -- the limit itself -- are served.  Either way the scalar build's number."""
    _need(binary, os.path.join(REF, "mb_scalar"))
    kw = dict(std_cases.SYNTHETIC["binary_symdir_exponential"])
    for nbeta, served in ((8, True), (10, False)):
        kw["nbetacat"] = nbeta
        want, _ = _lnl(os.path.join(REF, "mb_scalar"), std_cases.synthetic_nexus(beagle=None, **kw))
        got, out = _lnl(binary, std_cases.synthetic_nexus(beagle="always", **kw))
        if served:
            _served(out, marker)
        else:
            assert "stays on the host kernels: 10 beta categories" in out and "(standard data):" not in out, out[-1500:]
        assert abs(got - want) <= 1e-5 * abs(want), (nbeta, got, want)


def test_beta_category_limit_on_emulated_engine():
    _build_emu()
    _check_beta_category_limit(os.path.join(REF, "mb_emu_std"), "HOST EMULATION")


def test_patch_site_is_pinned():
    """The edits patch_std.py makes (one in likelihood.c, one in mcmc.c) must each apply to the reference exactly once (an upstream change fails here, not at run time)."""
    src = "/root/reference/src/likelihood.c"
    if not os.path.exists(src):
        pytest.skip("reference sources not present")
    import importlib.util
    import sys
    pdir = os.path.join(refrun.ROOT, "integration", "mrbayes", "patches")
    sys.path.insert(0, pdir)
    try:
        spec = importlib.util.spec_from_file_location("patch_std", os.path.join(pdir, "patch_std.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        with open(src) as fh:
            out = mod.patch(fh.read())
        with open("/root/reference/src/mcmc.c") as fh:
            out2 = mod.patch_mcmc(fh.read())
    finally:
        sys.path.remove(pdir)
    assert out.count("MbamdStdServes (m) == YES") == 1 and '#include "mbamd_std_glue.h"' in out
    assert out2.count("MbamdStdMaterialise (coldId, d)") == 1 and '#include "mbamd_std_glue.h"' in out2
    assert out2.count("MbamdStdFinalize ();") == 1


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(std_cases.SYNTHETIC))
def test_standard_data_on_mi355x(case):
    _check_synthetic(case, os.path.join(REF, "mb_amd_std"), "gfx950", scalings=("always", "dynamic"))


@pytest.mark.gpu
def test_beta_category_limit_on_mi355x():
    _check_beta_category_limit(os.path.join(REF, "mb_amd_std"), "gfx950")


@pytest.mark.gpu
def test_cynmix_morphology_on_mi355x():
    _check_cynmix(os.path.join(REF, "mb_amd_std"), "gfx950")


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(std_cases.ANCSTATES))
def test_standard_data_ancestral_states_on_mi355x(case):
    _check_ancstates(case, os.path.join(REF, "mb_amd_std"), "gfx950")


@pytest.mark.gpu
def test_standard_data_100x2000_on_mi355x():
    """100 taxa x 2 000 characters, 2 ... 6 states, gamma-4."""
    b = os.path.join(REF, "mb_amd_std")
    _need(b, os.path.join(REF, "mb_scalar"))
    want, _ = _lnl(os.path.join(REF, "mb_scalar"), std_cases.synthetic_nexus(beagle=None, **std_cases.BIG))
    got, out = _lnl(b, std_cases.synthetic_nexus(beagle="always", **std_cases.BIG))
    _served(out, "gfx950")
    assert abs(got - want) <= 1e-5 * abs(want), (got, want)


@pytest.mark.gpu
def test_standard_data_mcmc_on_mi355x():
    _check_mcmc(os.path.join(REF, "mb_amd_std"), "gfx950")


@pytest.mark.gpu
def test_hymfossil_morphology_on_mi355x():
    _check_hymfossil(os.path.join(REF, "mb_amd_std"), "gfx950")


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(std_cases.SECOND_ANALYSIS))
def test_second_analysis_in_a_session_on_mi355x(case):
    _check_second_analysis(case, os.path.join(REF, "mb_amd_std"), "gfx950")


@pytest.mark.gpu
def test_standard_data_mcmc_with_moving_frequencies_on_mi355x():
    _check_mcmc(os.path.join(REF, "mb_amd_std"), "gfx950", kw=std_cases.MCMC_SYMDIR)
