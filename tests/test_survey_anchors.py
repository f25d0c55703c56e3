"""The survey's own anchor vectors, asserted through the drop-in boundary.

The four numbers were printed by the real reference on its example data sets (seed = swapseed = 12345, one chain, one
run): primates at the default start, primates at the generation-2000 state of that run, avian_ovomucoids (WAG+G4),
replicase (codon M3).  /root/reference does not exist on the GPU box, so the alignments are rebuilt from the committed
pattern fixtures (tests/golden/*.npz: the compressed columns re-expanded by their weights; the default start depends on
the seeds and the taxon count only).  First the rebuilt files must give the survey's digits on the reference's NATIVE
kernels (oracle/_ref/mb; live, also on the GPU box -- the binary travels).  A BEAGLE build of the reference draws a
different default start from the same seeds, so the native run's start state (generation-0 rows of its .t / .p files,
precision=15) is handed to the engine-backed binary as a trees block + startvals, and that must print the anchor too:

  * CPU (`not gpu`): oracle/_ref/mb (native) and oracle/_ref/mb_emu (unmodified reference + TEST-ONLY host emulation)
  * GPU (`gpu`):     oracle/_ref/mb_amd (unmodified reference + mrbayes_amd/libhmsbeagle.so)
"""
import os
import re

import numpy as np
import pytest

from mrbayes_amd import data as mbdata
from mrbayes_amd.model import AA_ORDER, NUC, sense_codons
from tools import refrun

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# FMA/AVX build of the reference, screen line "Chain 1 -- <lnL>"
ANCHOR_PRIMATES_START = -8955.861383
ANCHOR_PRIMATES_GEN2000 = -5787.988379
ANCHOR_AVIAN = -6831.225867
ANCHOR_REPLICASE = -9418.182608
ABS_TOL_NATIVE = 2e-5            # the reference's own kernels on the rebuilt file: the printed digits up to the fp32 summation
                                 # order of the re-expanded columns (3e-9 relative; avian differs by 1e-5, the others by 0)
ABS_TOL_ENGINE = 1e-3            # |dlnL| <= 1e-3 on P < 1000 cases (and rel <= 1e-5)


def _alignment(case, datatype):
    z = np.load(os.path.join(GOLD, case + ".npz"))
    bits, w = z["bits"], z["weights"].astype(int)
    if datatype == "dna":
        inv = {}
        for ch in "ACGTRYMKSWHBVDN":
            inv.setdefault(mbdata.dna_bits(ch), ch)
        sym = lambda b: inv[int(b)]                          # noqa: E731
    elif datatype == "protein":
        one = {1 << i: AA_ORDER[i] for i in range(20)}
        sym = lambda b: one.get(int(b), "-")                 # noqa: E731
    else:
        nucs, _ = sense_codons()
        one = {1 << i: "".join(NUC[n] for n in c) for i, c in enumerate(nucs)}
        sym = lambda b: one.get(int(b), "---")               # noqa: E731
    seqs = ["".join(sym(b) * int(n) for b, n in zip(bits[t], w)) for t in range(bits.shape[0])]
    return ["t%d" % (i + 1) for i in range(len(seqs))], seqs


def _nexus(names, seqs, datatype, model, beagle, ngen=1, extra_mcmc="", trees="", startvals=""):
    # (seeds are set BEFORE the data block, as the survey's runs did: `set seed` then `execute <file>`)
    s = "#NEXUS\nbegin mrbayes;\n  set autoclose=yes nowarnings=yes seed=12345 swapseed=12345;\nend;\n"
    s += "begin data;\n  dimensions ntax=%d nchar=%d;\n  format datatype=%s gap=- missing=?;\n  matrix\n" % (
        len(names), len(seqs[0]), "dna" if datatype == "codon" else datatype)
    for n, q in zip(names, seqs):
        s += "%s  %s\n" % (n, q)
    s += "  ;\nend;\n" + trees + "begin mrbayes;\n  %s\n" % model
    if beagle:
        s += "  set usebeagle=yes beagledevice=gpu beagleprecision=single beaglescaling=%s;\n" % beagle
    s += startvals
    s += "  mcmc ngen=%d nchains=1 nruns=1 %s filename=anchor;\nend;\n" % (ngen, extra_mcmc)
    return s


CASES = {
    "primates": ("primates_gtr_g4", "dna", "lset nst=6 rates=gamma ngammacat=4;", ANCHOR_PRIMATES_START),
    "avian": ("avian_wag_g4", "protein", "prset aamodelpr=fixed(wag); lset rates=gamma ngammacat=4;", ANCHOR_AVIAN),
    "replicase": ("replicase_m3", "codon", "lset nucmodel=codon omegavar=M3;", ANCHOR_REPLICASE),
}


def _start_lnl(binary, key, beagle, env=None):
    case, datatype, model, _ = CASES[key]
    names, seqs = _alignment(case, datatype)
    out, _ = refrun.run_mb(binary, _nexus(names, seqs, datatype, model, beagle), env=env)
    return refrun.initial_lnl(out), out


def _named(newick):
    return re.sub(r"(?<=[(,])(\d+):", r"t\1:", newick)


def _start_state(key):
    """The default start state of the survey's run -- tree, branch lengths and every parameter MrBayes drew -- read from the
    generation-0 rows of the NATIVE run's .t / .p files (precision=15), as a trees block + startvals.  A BEAGLE build of the
    reference draws a DIFFERENT default start from the same seeds (its random-number stream differs: start tree, M3 class
    frequencies), so the engine is handed the native run's start explicitly instead of being asked to draw its own."""
    case, datatype, model, _ = CASES[key]
    names, seqs = _alignment(case, datatype)
    nex = _nexus(names, seqs, datatype, model, None, extra_mcmc="samplefreq=1").replace("swapseed=12345;", "swapseed=12345 precision=15;")
    out, _, files = refrun.run_mb(refrun.REF_MB, nex, keep=("anchor.t", "anchor.p"))
    tree = re.search(r"tree gen\.0 = \[&U\] ([^;]+);", files["anchor.t"])
    assert tree, files["anchor.t"][-800:]
    lines = [l for l in files["anchor.p"].splitlines() if l and not l.startswith("[")]
    row = dict(zip(lines[0].split("\t"), lines[1].split("\t")))
    assert row["Gen"] == "0"
    vals = []
    if key == "primates":
        vals.append("Revmat=(%s)" % ",".join(row["r(%s)" % x] for x in ("A<->C", "A<->G", "A<->T", "C<->G", "C<->T", "G<->T")))
        vals.append("Pi=(%s)" % ",".join(row["pi(%s)" % x] for x in "ACGT"))
        vals.append("Alpha=(%s)" % row["alpha"])
    elif key == "avian":
        vals.append("Alpha=(%s)" % row["alpha"])
    else:
        vals.append("Omega=(%s)" % ",".join([row["omega(%d)" % i] for i in (1, 2, 3)] + [row["pi(%d)" % i] for i in (1, 2, 3)]))
        vals.append("Pi=(%s)" % ",".join(v for k, v in zip(lines[0].split("\t"), lines[1].split("\t")) if re.fullmatch(r"pi\([ACGT]{3}\)", k)))
    return refrun.initial_lnl(out), (names, seqs, "t", _named(tree.group(1)), "tau=t V=t " + " ".join(vals))


def _state_lnl(binary, key, beagle, state, env=None):
    case, datatype, model, _ = CASES[key]
    names, seqs, tname, newick, vals = state
    trees = "begin trees;\n  tree %s = [&U] %s;\nend;\n" % (tname, newick)
    out, _ = refrun.run_mb(binary, _nexus(names, seqs, datatype, model, beagle, trees=trees, startvals="  startvals %s;\n" % vals), env=env)
    return refrun.initial_lnl(out), out


def _gen2000_state():
    """The generation-2000 state of the primates run on the reference's NATIVE kernels, as a trees block + startvals
    (read from the run's checkpoint file, the format `startvals` takes: reference src/mcmc.c:11192-11250)."""
    names, seqs = _alignment("primates_gtr_g4", "dna")
    nex = _nexus(names, seqs, "dna", CASES["primates"][2], None, ngen=2000, extra_mcmc="checkfreq=2000 samplefreq=100")
    out, _, files = refrun.run_mb(refrun.REF_MB, nex, keep=("anchor.ckp",))
    ckp = files["anchor.ckp"]
    tree = re.search(r"tree\s+(\S+)\s*=\s*(\[&U\]\s*)?([^;]+);", ckp)
    assert tree, ckp[:500]
    vals = re.search(r"startvals\s+(.*?);\s*\n", ckp, re.S)
    assert vals, ckp[:2000]
    return names, seqs, tree.group(1), tree.group(3), " ".join(vals.group(1).split())


def _need(path):
    if not os.path.exists(path):
        pytest.skip("%s was not built (needs the reference sources at build time)" % os.path.relpath(path, ROOT))


def _check_start(key, binary, marker):
    want = CASES[key][3]
    native, state = _start_state(key)
    assert abs(native - want) <= ABS_TOL_NATIVE, (key, native, want)            # the rebuilt alignment IS the survey's run
    again, _ = _state_lnl(refrun.REF_MB, key, None, state)
    assert abs(again - want) <= ABS_TOL_NATIVE, (key, again, want)              # ... and the captured state IS its start
    for scaling in ("dynamic", "always"):
        ours, out = _state_lnl(binary, key, scaling, state)
        assert marker in out, out[-1500:]
        assert abs(ours - want) <= ABS_TOL_ENGINE and abs(ours - want) <= 1e-5 * abs(want), (key, scaling, ours, want)


def _check_gen2000(binary, marker):
    state = _gen2000_state()
    native, _ = _state_lnl(refrun.REF_MB, "primates", None, state)
    assert abs(native - ANCHOR_PRIMATES_GEN2000) <= ABS_TOL_NATIVE, native
    for scaling in ("dynamic", "always"):
        ours, out = _state_lnl(binary, "primates", scaling, state)
        assert marker in out, out[-1500:]
        assert abs(ours - ANCHOR_PRIMATES_GEN2000) <= ABS_TOL_ENGINE, (scaling, ours)


@pytest.mark.parametrize("key", sorted(CASES))
def test_anchor_start_states_native_and_emulated(key):
    _need(refrun.REF_MB)
    _need(refrun.REF_MB_EMU)
    _check_start(key, refrun.REF_MB_EMU, "mbamd")


def test_anchor_primates_generation_2000_native_and_emulated():
    _need(refrun.REF_MB)
    _need(refrun.REF_MB_EMU)
    _check_gen2000(refrun.REF_MB_EMU, "mbamd")


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(CASES))
def test_anchor_start_states_on_mi355x(key):
    _need(refrun.REF_MB)
    _need(refrun.REF_MB_AMD)
    _check_start(key, refrun.REF_MB_AMD, "mbamd HIP gfx950")


@pytest.mark.gpu
def test_anchor_primates_generation_2000_on_mi355x():
    _need(refrun.REF_MB)
    _need(refrun.REF_MB_AMD)
    _check_gen2000(refrun.REF_MB_AMD, "mbamd HIP gfx950")
