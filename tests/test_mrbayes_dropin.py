"""The drop-in boundary end to end: the UNMODIFIED reference MrBayes (compiled from /root/reference/src with
-DBEAGLE_ENABLED against include/libhmsbeagle/beagle.h, see oracle/Makefile) driving this engine through
src/mbbeagle.c, compared with the same reference's native CPU kernels on the same fixed state.

  * CPU  (`not gpu`): oracle/_ref/mb_emu -- linked to the TEST-ONLY host-emulation build of the engine; needs
    the reference sources, i.e. runs in the build container only (skipped elsewhere).
  * GPU  (`gpu`): oracle/_ref/mb_amd -- linked to the product library mrbayes_amd/libhmsbeagle.so; the binary
    travels to the GPU box with the repository snapshot.
"""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from mrbayes_amd import data as mbdata
from mrbayes_amd import tree as mbtree
from tools import refrun

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REVMAT, PI, ALPHA = [0.10, 0.30, 0.05, 0.08, 0.40, 0.07], [0.35, 0.25, 0.15, 0.25], 0.6


def _case(ntaxa, nsites, p_gap):
    st = mbdata.synthetic_states(ntaxa, nsites, 4, 11, 0.15, p_gap)
    tr = mbtree.random_tree(ntaxa, 12, brlen=0.05)
    return st, tr


def _lnl(binary, st, tr, beagle):
    out, _ = refrun.run_mb(binary, refrun.known_answer_nexus(st, tr, REVMAT, PI, ALPHA, beagle=beagle))
    return refrun.initial_lnl(out), out


@pytest.mark.parametrize("scaling", ["dynamic", "always"])
def test_unmodified_mrbayes_on_emulated_engine(scaling):
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference sources not present (build container only)")
    from tests.hostemu import build_emu
    build_emu.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/mb", "_ref/mb_emu"],
                          stdout=subprocess.DEVNULL)
    st, tr = _case(30, 600, 0.03)
    native, _ = _lnl(refrun.REF_MB, st, tr, None)
    ours, out = _lnl(refrun.REF_MB_EMU, st, tr, scaling)
    assert "mbamd" in out                                  # MrBayes reports the engine it picked
    assert abs(ours - native) / abs(native) < 1e-5, (ours, native)


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["dynamic", "always"])
def test_unmodified_mrbayes_on_mi355x(scaling):
    if not os.path.exists(refrun.REF_MB_AMD):
        pytest.skip("oracle/_ref/mb_amd was not built (needs the reference sources at build time)")
    st, tr = _case(60, 3000, 0.02)
    ours, out = _lnl(refrun.REF_MB_AMD, st, tr, scaling)
    assert "mbamd HIP gfx950" in out
    if os.path.exists(refrun.REF_MB):
        native, _ = _lnl(refrun.REF_MB, st, tr, None)
        assert abs(ours - native) / abs(native) < 1e-5, (ours, native)
    # and a short MCMC run (default moves, 2 heated chains) must complete on the GPU engine
    out, _ = refrun.run_mb(refrun.REF_MB_AMD, refrun.mcmc_nexus(st, tr, 300, beagle=scaling, nchains=2))
    assert "Analysis completed" in out, out[-1500:]


# ---- rooted (clock) trees: MrBayes integrates at the root with beagleCalculateRootLogLikelihoods (src/mbbeagle.c:1251-1257)
def _clock_nexus(beagle, ngen=1):
    import random
    st, _ = _case(24, 500, 0.02)
    names = ["t%d" % (i + 1) for i in range(st.shape[0])]
    seqs = ["".join("ACGT-"[x] for x in row) for row in st]
    rng = random.Random(5)
    nodes = [(n, 0.0) for n in names]                          # (newick, height): a random ultrametric tree
    while len(nodes) > 1:
        i, j = sorted(rng.sample(range(len(nodes)), 2))
        b = nodes.pop(j)
        a = nodes.pop(i)
        h = max(a[1], b[1]) + 0.03 * (0.5 + rng.random())
        nodes.append(("(%s:%.12f,%s:%.12f)" % (a[0], h - a[1], b[0], h - b[1]), h))
    s = "#NEXUS\nbegin data;\n  dimensions ntax=%d nchar=%d;\n" % (len(names), len(seqs[0]))
    s += "  format datatype=dna interleave=no gap=- missing=?;\n  matrix\n"
    for n, q in zip(names, seqs):
        s += "%s  %s\n" % (n, q)
    s += "  ;\nend;\nbegin mrbayes;\n  set autoclose=yes nowarnings=yes seed=12345 swapseed=12345 precision=15;\n"
    s += "  lset nst=6 rates=gamma ngammacat=4;\n  prset brlenspr=clock:uniform;\n"
    if beagle:
        s += "  set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=%s;\n" % beagle
    s += "end;\nbegin trees;\n  tree t = [&R] %s;\nend;\n" % nodes[0][0]
    s += "begin mrbayes;\n  startvals tau=t V=t Revmat=(0.10,0.30,0.05,0.08,0.40,0.07) Pi=(0.35,0.25,0.15,0.25) Alpha=(0.6);\n"
    s += "  mcmc ngen=%d nchains=1 nruns=1 samplefreq=%d printfreq=%d diagnfreq=%d filename=ck;\nend;\n" % (ngen, ngen, ngen, ngen)
    return s


@pytest.mark.parametrize("scaling", ["dynamic", "always"])
def test_clock_tree_on_emulated_engine(scaling):
    if not os.path.exists(refrun.REF_MB_EMU):
        pytest.skip("oracle/_ref/mb_emu not built (build container only)")
    native = refrun.initial_lnl(refrun.run_mb(refrun.REF_MB, _clock_nexus(None))[0])
    out, _ = refrun.run_mb(refrun.REF_MB_EMU, _clock_nexus(scaling), env={"MBAMD_API_TRACE": "1"})
    assert "beagleCalculateRootLogLikelihoods" in out and "beagleCalculateEdgeLogLikelihoods" not in out
    assert abs(refrun.initial_lnl(out) - native) / abs(native) < 1e-5


@pytest.mark.gpu
def test_clock_tree_on_mi355x():
    if not os.path.exists(refrun.REF_MB_AMD):
        pytest.skip("oracle/_ref/mb_amd was not built (needs the reference sources at build time)")
    out, _ = refrun.run_mb(refrun.REF_MB_AMD, _clock_nexus("dynamic"), env={"MBAMD_API_TRACE": "1"})
    assert "mbamd HIP gfx950" in out and "beagleCalculateRootLogLikelihoods" in out
    ours = refrun.initial_lnl(out)
    if os.path.exists(refrun.REF_MB):
        native = refrun.initial_lnl(refrun.run_mb(refrun.REF_MB, _clock_nexus(None))[0])
        assert abs(ours - native) / abs(native) < 1e-5, (ours, native)
    out, _ = refrun.run_mb(refrun.REF_MB_AMD, _clock_nexus("always", ngen=300))       # clock moves, rooted integration every generation
    assert "Analysis completed" in out, out[-1500:]


# ---- partially ambiguous tips (IUPAC codes): those taxa reach the engine as tip PARTIALS ------------------------
def _ambiguous_nexus(beagle):
    st, tr = _case(20, 500, 0.02)
    text = refrun.known_answer_nexus(st, tr, REVMAT, PI, ALPHA, beagle=beagle)
    rng = np.random.default_rng(3)
    lines = text.split("\n")
    for i, line in enumerate(lines):
        parts = line.split("  ")
        if len(parts) == 2 and parts[0].startswith("t") and set(parts[1]) <= set("ACGT-"):
            if int(parts[0][1:]) % 3 == 0:                     # every third taxon carries IUPAC ambiguity codes
                seq = list(parts[1])
                for j in rng.choice(len(seq), size=len(seq) // 10, replace=False):
                    seq[j] = "RYMKSWBDHVN"[int(rng.integers(0, 11))]
                lines[i] = parts[0] + "  " + "".join(seq)
    return "\n".join(lines)


def test_ambiguity_codes_on_emulated_engine():
    if not os.path.exists(refrun.REF_MB_EMU):
        pytest.skip("oracle/_ref/mb_emu not built (build container only)")
    native = refrun.initial_lnl(refrun.run_mb(refrun.REF_MB, _ambiguous_nexus(None))[0])
    out, _ = refrun.run_mb(refrun.REF_MB_EMU, _ambiguous_nexus("dynamic"))
    assert "mbamd" in out
    assert abs(refrun.initial_lnl(out) - native) / abs(native) < 1e-5


@pytest.mark.gpu
def test_ambiguity_codes_on_mi355x():
    if not os.path.exists(refrun.REF_MB_AMD):
        pytest.skip("oracle/_ref/mb_amd was not built (needs the reference sources at build time)")
    out, _ = refrun.run_mb(refrun.REF_MB_AMD, _ambiguous_nexus("always"))
    assert "mbamd HIP gfx950" in out
    ours = refrun.initial_lnl(out)
    if os.path.exists(refrun.REF_MB):
        native = refrun.initial_lnl(refrun.run_mb(refrun.REF_MB, _ambiguous_nexus(None))[0])
        assert abs(ours - native) / abs(native) < 1e-5, (ours, native)
    out, _ = refrun.run_mb(refrun.REF_MB_AMD, _ambiguous_nexus("dynamic").replace("ngen=1 ", "ngen=300 "))
    assert "Analysis completed" in out, out[-1500:]


# ---- partitioned analysis: one engine instance per division, both alive in the same MrBayes process -------------
def _partitioned_nexus(beagle, ngen=1, same_shape=False, ntaxa=16, nsites=600, first=350):
    """same_shape: both divisions GTR+G4 (same state / category / eigen-part counts) -- what MrBayes' v3 build merges into
    ONE multi-partition instance (reference src/mbbeagle.c:1519-1546); otherwise GTR+G4 next to HKY+I."""
    st, tr = _case(ntaxa, nsites, 0.02)
    names = ["t%d" % (i + 1) for i in range(st.shape[0])]
    seqs = ["".join("ACGT-"[x] for x in row) for row in st]
    s = "#NEXUS\nbegin data;\n  dimensions ntax=%d nchar=%d;\n" % (len(names), len(seqs[0]))
    s += "  format datatype=dna interleave=no gap=- missing=?;\n  matrix\n"
    for n, q in zip(names, seqs):
        s += "%s  %s\n" % (n, q)
    s += "  ;\nend;\nbegin mrbayes;\n  set autoclose=yes nowarnings=yes seed=12345 swapseed=12345 precision=15;\n"
    s += "  charset first = 1-%d;\n  charset second = %d-%d;\n  partition genes = 2: first, second;\n  set partition=genes;\n" % (first, first + 1, nsites)
    if same_shape:
        s += "  lset applyto=(all) nst=6 rates=gamma ngammacat=4;\n"
    else:
        s += "  lset applyto=(1) nst=6 rates=gamma ngammacat=4;\n  lset applyto=(2) nst=2 rates=propinv;\n"
    s += "  unlink revmat=(all) tratio=(all) statefreq=(all) shape=(all) pinvar=(all);\n  prset applyto=(all) ratepr=variable;\n"
    if beagle:
        s += "  set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=%s;\n" % beagle
    s += "end;\nbegin trees;\n  tree t = [&U] %s\nend;\n" % tr.to_newick(names)
    s += "begin mrbayes;\n  startvals tau=t V=t;\n"
    s += "  mcmc ngen=%d nchains=2 nruns=1 samplefreq=%d printfreq=%d diagnfreq=%d filename=part;\nend;\n" % (
        ngen, max(ngen, 1), max(ngen, 1), max(ngen, 1))
    return s


def _check_partitioned(binary, marker):
    native = refrun.initial_lnl(refrun.run_mb(refrun.REF_MB, _partitioned_nexus(None))[0])
    out, _ = refrun.run_mb(binary, _partitioned_nexus("dynamic"))
    assert out.count(marker) >= 2, out[-2500:]             # both divisions run on the engine
    ours = refrun.initial_lnl(out)
    assert abs(ours - native) / abs(native) < 1e-5, (ours, native)
    out, _ = refrun.run_mb(binary, _partitioned_nexus("dynamic", ngen=400))
    assert "Analysis completed" in out, out[-1500:]


def _check_multi_partition(binary):
    """MrBayes' BEAGLE v3 build: resource benchmark, then ONE multi-partition instance for both divisions
    (beagleSetPatternPartitions, *ByPartition calls) -- same lnL as the native kernels, and a default-move run completes
    (per-division partial updates, rejects, dynamic rescaling through the ByPartition scale-factor calls)."""
    native = refrun.initial_lnl(refrun.run_mb(refrun.REF_MB, _partitioned_nexus(None, same_shape=True))[0])
    for scaling in ("dynamic", "always"):
        out, _ = refrun.run_mb(binary, _partitioned_nexus(scaling, same_shape=True), env={"MBAMD_API_TRACE": "1"})
        assert "for 2 divisions" in out, out[-2500:]                 # "Using BEAGLE v... resource 0 for 2 divisions"
        assert "beagleSetPatternPartitions(2 partitions)" in out and "beagleUpdatePartialsByPartition" in out
        assert "beagleCalculateEdgeLogLikelihoodsByPartition" in out
        ours = refrun.initial_lnl(out)
        assert abs(ours - native) / abs(native) < 1e-5, (scaling, ours, native)
    out, _ = refrun.run_mb(binary, _partitioned_nexus("dynamic", ngen=400, same_shape=True))
    assert "Analysis completed" in out, out[-1500:]
    # divisions of different shape cannot be merged: the v3 build falls back to one instance per division
    native = refrun.initial_lnl(refrun.run_mb(refrun.REF_MB, _partitioned_nexus(None))[0])
    out, _ = refrun.run_mb(binary, _partitioned_nexus("dynamic"))
    assert abs(refrun.initial_lnl(out) - native) / abs(native) < 1e-5


def test_multi_partition_v3_on_emulated_engine():
    if not (os.path.exists(refrun.REF_MB_EMU_V3) and os.path.exists(refrun.REF_MB)):
        pytest.skip("oracle/_ref binaries not built (build container only)")
    _check_multi_partition(refrun.REF_MB_EMU_V3)


@pytest.mark.gpu
def test_multi_partition_v3_on_mi355x():
    if not (os.path.exists(refrun.REF_MB_AMD_V3) and os.path.exists(refrun.REF_MB)):
        pytest.skip("oracle/_ref/mb_amd_v3 was not built (needs the reference sources at build time)")
    _check_multi_partition(refrun.REF_MB_AMD_V3)


def test_partitioned_analysis_on_emulated_engine():
    if not (os.path.exists(refrun.REF_MB_EMU) and os.path.exists(refrun.REF_MB)):
        pytest.skip("oracle/_ref binaries not built (build container only)")
    _check_partitioned(refrun.REF_MB_EMU, "Impl Name : mbamd")


@pytest.mark.gpu
def test_partitioned_analysis_on_mi355x():
    if not (os.path.exists(refrun.REF_MB_AMD) and os.path.exists(refrun.REF_MB)):
        pytest.skip("oracle/_ref binaries not built (need the reference sources at build time)")
    _check_partitioned(refrun.REF_MB_AMD, "Impl Name : mbamd HIP gfx950")


# ---- general-state models through the real src/mbbeagle.c ---------------------------------------------------
def _general_case(kind, ntaxa, nsites):
    nstates = {"wag": 20, "m3": 61}[kind]
    st = mbdata.synthetic_states(ntaxa, nsites, nstates, 11, 0.15, 0.03)
    tr = mbtree.random_tree(ntaxa, 12, brlen=0.05)
    return st, tr


def _check_wag(binary):
    """Protein data, fixed WAG + gamma(4): no randomly drawn starting values, so the engine-driven binary must
    print the native kernels' initial lnL."""
    st, tr = _general_case("wag", 14, 300)
    out, _ = refrun.run_mb(binary, refrun.model_nexus("wag", st, tr, beagle="dynamic"))
    assert "mbamd" in out, out[-1500:]
    ours = refrun.initial_lnl(out)
    if os.path.exists(refrun.REF_MB):
        native = refrun.initial_lnl(refrun.run_mb(refrun.REF_MB, refrun.model_nexus("wag", st, tr))[0])
        assert abs(ours - native) / abs(native) < 1e-5, (ours, native)
    return ours


def _check_m3(binary, oracle):
    """Codon M3 (three eigen-systems mixed at the root, nCijkParts = 3, reference src/mbbeagle.c:1194-1206,
    1231-1274).  MrBayes draws the omega-class frequencies at start-up (and draws them differently in its BEAGLE
    and native configurations), so the check is against the CPU oracle evaluated at the values this very run
    started from (generation-0 row of its .p file)."""
    from mrbayes_amd.division import build_division, _tips_from_states
    st, tr = _general_case("m3", 9, 80)
    out, row = refrun.run_mb_with_samples(binary, refrun.model_nexus("m3", st, tr, ngen=1, beagle="dynamic"))
    assert "mbamd" in out, out[-1500:]
    ours = refrun.initial_lnl(out)
    omegas = [row["omega(%d)" % i] for i in (1, 2, 3)]
    freqs = [row["pi(%d)" % i] for i in (1, 2, 3)]
    cols, counts = np.unique(st, axis=1, return_counts=True)
    tip_states, tip_partials = _tips_from_states(cols)
    div = build_division("m3", tr, counts.astype(float), tip_states, tip_partials, omegas=omegas, omega_freqs=freqs, ncat=1)
    want = oracle.tree_loglike(div, use_shortcuts=False)
    assert abs(ours - want) / abs(want) < 2e-6, (ours, want, omegas, freqs)


def test_general_state_models_on_emulated_engine(oracle):
    if not os.path.exists(refrun.REF_MB_EMU):
        pytest.skip("oracle/_ref/mb_emu not built (build container only)")
    _check_wag(refrun.REF_MB_EMU)
    _check_m3(refrun.REF_MB_EMU, oracle)


@pytest.mark.gpu
def test_general_state_models_on_mi355x(oracle):
    if not os.path.exists(refrun.REF_MB_AMD):
        pytest.skip("oracle/_ref/mb_amd was not built (needs the reference sources at build time)")
    _check_wag(refrun.REF_MB_AMD)
    _check_m3(refrun.REF_MB_AMD, oracle)
    # a short protein MCMC run (default moves: partial updates -> spine kernel) must complete on the GPU engine
    st, tr = _general_case("wag", 14, 300)
    out, _ = refrun.run_mb(refrun.REF_MB_AMD, refrun.model_nexus("wag", st, tr, ngen=400, beagle="dynamic"))
    assert "Analysis completed" in out, out[-1500:]


# ---- the reference's own integration check (testing/test1.nex style), engine-driven ------------------------
def _tap_expected():
    with open(os.path.join(ROOT, "tests", "golden", "primates_tap.json")) as fh:
        return json.load(fh)


def test_tap_style_run_on_emulated_engine():
    """Short version on the CPU (host-emulation engine): mixed nst + invgamma (pInvar goes through
    beagleGetSiteLogLikelihoods), 2 runs x 4 chains, swaps: completes and climbs to a sane likelihood."""
    if not os.path.exists(refrun.REF_MB_EMU):
        pytest.skip("oracle/_ref/mb_emu not built (build container only)")
    from tools import gen_tap_fixture as tap
    names, seqs = tap.alignment_from_fixture()
    out, _ = refrun.run_mb(refrun.REF_MB_EMU, tap.tap_nexus(names, seqs, 1500, beagle="dynamic"))
    res = tap.parse(out)
    assert res["completed"] == 1, out[-1500:]
    assert -6200.0 < res["best_cold_lnL_run1"] < -5690.0, res


@pytest.mark.gpu
def test_tap_style_run_on_mi355x():
    """The full 20 000-generation check on the GPU against the statistics the reference's native kernels produced
    for the same analysis (tests/golden/primates_tap.json, tools/gen_tap_fixture.py): the reference's own test
    accepts a 15-unit window for the best cold-chain lnL (testing/runtests.sh.in:82-100)."""
    if not os.path.exists(refrun.REF_MB_AMD):
        pytest.skip("oracle/_ref/mb_amd was not built (needs the reference sources at build time)")
    from tools import gen_tap_fixture as tap
    want = _tap_expected()
    names, seqs = tap.alignment_from_fixture()
    out, _ = refrun.run_mb(refrun.REF_MB_AMD, tap.tap_nexus(names, seqs, 20000, beagle="dynamic"), timeout=1200)
    res = tap.parse(out)
    assert res["completed"] == 1, out[-1500:]
    assert abs(res["best_cold_lnL_run1"] - want["best_cold_lnL_run1"]) < 15.0, (res, want)
    assert abs(res["TL_mean"] - want["TL_mean"]) < 0.35 * want["TL_mean"], (res, want)


# ---- device parsimony: the reference + integration/mrbayes/mbamd_pars_glue.c ----------------------
def _trajectory(out):
    """the per-generation chain lines of MrBayes' screen output, without the time-remaining column"""
    return [re.sub(r" -- \d+:\d\d:\d\d\s*$", "", l) for l in out.split("\n") if re.match(r"\s+\d+ -- [\[\(-]", l)]


def _pars_nexus(ngen, nchains=2, kind="dna"):
    if kind == "dna":
        st, _ = _case(20, 400, 0.03)
        nex = refrun.mcmc_nexus(st, None, ngen, beagle="dynamic", nchains=nchains)
    elif kind == "clock":                              # rooted clock tree: Move_ParsSPRClock (the node-length shape of the binding)
        nex = _clock_nexus("dynamic", ngen=ngen)
    else:
        st, tr = _general_case("wag", 12, 120)
        nex = refrun.model_nexus("wag", st, tr, ngen=ngen, beagle="dynamic")
    return nex.replace("printfreq=%d" % ngen, "printfreq=%d" % max(ngen // 20, 1))


def _check_device_parsimony(plain, patched, ngen, kind="dna"):
    nex = _pars_nexus(ngen, kind=kind)
    ref_out, _ = refrun.run_mb(plain, nex)
    # 1. every GetParsDP / GetParsFP / candidate loop also runs in the reference's own host functions, inside the
    #    same process, and is compared word for word (the glue exits non-zero on the first difference)
    chk_out, _ = refrun.run_mb(patched, nex, env={"MBAMD_PARS_CHECK": "1", "MBAMD_API_TRACE": "1"})
    text = chk_out
    assert "Analysis completed" in chk_out, text[-2000:]
    m = re.search(r"mbamd parsimony check: (\d+) comparisons against the host functions, all equal", text)
    assert m and int(m.group(1)) > 1000, text[-2000:]
    assert "mbamdParsDownPass" in text and "mbamdParsScore" in text and (kind == "clock" or "mbamdParsFinalPass" in text)
    # 2. device only: the same proposals, hence the same chain, generation for generation
    dev_out, _ = refrun.run_mb(patched, nex)
    assert "Analysis completed" in dev_out
    a, b, c = _trajectory(ref_out), _trajectory(chk_out), _trajectory(dev_out)
    assert len(a) > 10 and a == b and a == c, (a[-1], b[-1], c[-1])
    # 3. switched off (MBAMD_DEVICE_PARSIMONY=0) the patched binary is the reference
    off_out, _ = refrun.run_mb(patched, nex, env={"MBAMD_DEVICE_PARSIMONY": "0", "MBAMD_API_TRACE": "1"})
    assert "mbamdPars" not in off_out and _trajectory(off_out) == a


@pytest.mark.parametrize("kind", ["dna", "wag", "clock"])
def test_device_parsimony_on_emulated_engine(kind):
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference sources not present (build container only)")
    from tests.hostemu import build_emu
    build_emu.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/mb_emu", "_ref/mb_emu_pars"], stdout=subprocess.DEVNULL)
    _check_device_parsimony(refrun.REF_MB_EMU, refrun.REF_MB_EMU_PARS, 1500 if kind == "dna" else 300, kind)


@pytest.mark.gpu
def test_device_parsimony_on_mi355x():
    if not os.path.exists(refrun.REF_MB_AMD_PARS):
        pytest.skip("oracle/_ref/mb_amd_pars was not built (needs the reference sources at build time)")
    _check_device_parsimony(refrun.REF_MB_AMD, refrun.REF_MB_AMD_PARS, 3000)
    _check_device_parsimony(refrun.REF_MB_AMD, refrun.REF_MB_AMD_PARS, 1500, "clock")


# ---- `set beagleprecision=double`: the fp64 engine (mbamd_f64.h) behind the unmodified binary -------------------------
def _check_double_precision(binary, marker):
    st, tr = _case(30, 600, 0.03)
    want, _ = _lnl(refrun.REF_MB_FP64, st, tr, None)                   # the reference's own double build, native kernels
    native32, _ = _lnl(refrun.REF_MB, st, tr, None)
    for scaling in ("dynamic", "always"):
        nex = refrun.known_answer_nexus(st, tr, REVMAT, PI, ALPHA, beagle=scaling).replace("beagleprecision=single", "beagleprecision=double")
        out, _ = refrun.run_mb(binary, nex)
        assert marker in out and "double-precision" in out, out[-1500:]
        ours = refrun.initial_lnl(out)
        assert abs(ours - want) <= 2e-6 + 1e-11 * abs(want), (scaling, ours, want)     # same parameters in the same process: to the printed digits
        assert abs(ours - want) < abs(native32 - want)                                 # and closer than the reference's fp32 build
    # the v3 build puts both divisions of a partitioned analysis into ONE multi-partition instance: the *ByPartition surface in fp64
    if "v3" in os.path.basename(binary):
        pn = _partitioned_nexus(None, same_shape=True)
        want2 = refrun.initial_lnl(refrun.run_mb(refrun.REF_MB_FP64, pn)[0])
        for scaling in ("dynamic", "always"):
            out, _ = refrun.run_mb(binary, _partitioned_nexus(scaling, same_shape=True).replace("beagleprecision=single", "beagleprecision=double"),
                                   env={"MBAMD_API_TRACE": "1", "MBAMD_VERBOSE": "1"})
            assert "for 2 divisions" in out and "double-precision" in out and "[mbamd] error" not in out, out[-2500:]
            assert abs(refrun.initial_lnl(out) - want2) <= 2e-6 + 1e-11 * abs(want2), (scaling, refrun.initial_lnl(out), want2)
        out, _ = refrun.run_mb(binary, _partitioned_nexus("dynamic", ngen=200, same_shape=True).replace("beagleprecision=single", "beagleprecision=double"),
                               env={"MBAMD_VERBOSE": "1"})
        assert "Analysis completed" in out and "[mbamd] error" not in out, out[-1500:]
        return
    # a short default-mix run (dynamic rescaling at the double-precision frequency, src/mcmc.c:6226-6228) completes
    nex = refrun.mcmc_nexus(st, tr, 200, beagle="dynamic", nchains=2).replace("beagleprecision=single", "beagleprecision=double")
    out, _ = refrun.run_mb(binary, nex)
    assert "Analysis completed" in out, out[-1500:]


def test_double_precision_on_emulated_engine():
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference sources not present (build container only)")
    from tests.hostemu import build_emu
    build_emu.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/mb", "_ref/mb_fp64", "_ref/mb_emu"], stdout=subprocess.DEVNULL)
    _check_double_precision(refrun.REF_MB_EMU, "mbamd")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/mb_emu_v3"], stdout=subprocess.DEVNULL)
    _check_double_precision(refrun.REF_MB_EMU_V3, "mbamd")


@pytest.mark.gpu
def test_double_precision_on_mi355x():
    if not (os.path.exists(refrun.REF_MB_AMD) and os.path.exists(refrun.REF_MB_FP64)):
        pytest.skip("oracle/_ref/mb_amd / mb_fp64 were not built (need the reference sources at build time)")
    _check_double_precision(refrun.REF_MB_AMD, "mbamd HIP gfx950")
    if os.path.exists(refrun.REF_MB_AMD_V3):
        _check_double_precision(refrun.REF_MB_AMD_V3, "mbamd HIP gfx950")


# ---- pattern compression (second half): CompressData's O(columns^2) search as a hash-table lookup ------
def test_pattern_compression_binding():
    """integration/mrbayes/mbamd_compress_glue.c inside the patched binary: the same site patterns, the same counts, the
    same log-likelihood as the reference's own search -- which runs beside it under MBAMD_COMPRESS_CHECK=1 and must agree on
    every column (the binding exits non-zero otherwise)."""
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference sources not present (build container only)")
    from tests.hostemu import build_emu
    build_emu.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/mb_emu", "_ref/mb_emu_pars"], stdout=subprocess.DEVNULL)
    st, tr = _case(20, 300, 0.03)
    st = np.concatenate([st, st[:, ::3], st[:, 5:200]], axis=1)               # every pattern several times, out of order
    nex = refrun.known_answer_nexus(st, tr, REVMAT, PI, ALPHA, beagle="dynamic")
    plain, _ = refrun.run_mb(refrun.REF_MB_EMU, nex)
    checked, _ = refrun.run_mb(refrun.REF_MB_EMU_PARS, nex, env={"MBAMD_COMPRESS_CHECK": "1"})
    hashed, _ = refrun.run_mb(refrun.REF_MB_EMU_PARS, nex)
    off, _ = refrun.run_mb(refrun.REF_MB_EMU_PARS, nex, env={"MBAMD_HASH_COMPRESS": "0"})
    m = re.search(r"mbamd compression check: (\d+) columns, the reference's search agreed on all of them", checked)
    assert m and int(m.group(1)) >= st.shape[1], checked[-1500:]
    pat = [re.findall(r"Division 1 has (\d+) unique site patterns", o) for o in (plain, checked, hashed, off)]
    assert pat[0] and all(p == pat[0] for p in pat) and int(pat[0][0]) < st.shape[1] // 2, pat
    lnl = [refrun.initial_lnl(o) for o in (plain, checked, hashed, off)]
    assert all(v == lnl[0] for v in lnl), lnl
    # a codon-width case exercises nCharsPerSite > 1
    stc, trc = _general_case("m3", 10, 60)
    nexc = refrun.model_nexus("m3", np.concatenate([stc, stc[:, :25]], axis=1), trc, ngen=1, beagle="dynamic")
    a, _ = refrun.run_mb(refrun.REF_MB_EMU, nexc)
    b, _ = refrun.run_mb(refrun.REF_MB_EMU_PARS, nexc, env={"MBAMD_COMPRESS_CHECK": "1"})
    assert "agreed on all of them" in b and refrun.initial_lnl(a) == refrun.initial_lnl(b)
    assert re.findall(r"has (\d+) unique site patterns", a) == re.findall(r"has (\d+) unique site patterns", b)
