"""Parity tests proper: the PRODUCT library (mrbayes_amd/libhmsbeagle.so, HIP, gfx950) on a real MI355X,
called through its C ABI and compared with the CPU oracle, the golden fixtures written by the real
reference binaries, and -- at the BASELINE sizes -- size-independent properties of the recursion.

Every test is `@pytest.mark.gpu`; the library is loaded from the tree (never a CPU build) and the
fixture fails loudly when it or the GPU is missing.
"""
import os

import numpy as np
import pytest

from mrbayes_amd import beagle as bg
from mrbayes_amd import likelihood as lk
from mrbayes_amd import tree as mbtree
from mrbayes_amd.division import division_from_golden, synthetic_division
from tests import engine_checks as ec

pytestmark = pytest.mark.gpu

SMALL = ["primates_gtr_g4", "primates_gtr_ig4", "primates_gtr_equal", "avian_wag_g4", "replicase_m3",
         "synth_dna_gaps", "synth_aa_wag", "synth_codon_m3"]


@pytest.fixture(scope="module")
def gpu():
    lib = bg.library()                       # raises if libhmsbeagle.so was not built
    assert "hostemu" not in lib.path
    res = lib.resources()
    assert res, "no HIP device visible: the engine has no CPU path"
    assert res[0][2] & bg.BEAGLE_FLAG_PROCESSOR_GPU
    return lib


def test_resource_is_a_gpu(gpu):
    name, desc, flags = gpu.resources()[0]
    assert "gfx" in desc


@pytest.mark.parametrize("case", ["primates_gtr_g4", "avian_wag_g4", "replicase_m3"])
def test_transition_matrices(gpu, oracle, golden_dir, case):
    ec.check_transition_matrices(gpu, oracle, division_from_golden(golden_dir, case))


@pytest.mark.parametrize("nstates,ncat,npat", [(4, 4, 100), (4, 1, 64), (4, 3, 65), (4, 8, 130), (20, 4, 70),
                                                 (20, 1, 200), (61, 1, 33), (61, 1, 129), (61, 3, 40),
                                                 (16, 2, 10), (2, 4, 5), (20, 4, 1), (4, 4, 1),
                                                 (20, 3, 95), (20, 2, 64), (61, 2, 70), (33, 1, 50), (5, 4, 40), (64, 1, 31),
                                                 (60, 1, 45), (62, 2, 33), (63, 1, 70), (16, 4, 97), (8, 4, 50), (8, 1, 200), (8, 2, 33), (2, 1, 100), (2, 4, 70), (40, 4, 70), (40, 1, 33)])
def test_single_operations(gpu, oracle, nstates, ncat, npat):
    ec.check_single_operations(gpu, oracle, nstates, ncat, npat)


@pytest.mark.parametrize("nstates,ncat,npat", [(4, 4, 1000), (4, 1, 65), (4, 3, 129), (20, 4, 700), (61, 1, 330), (61, 3, 40), (8, 4, 500),
                                                 (40, 4, 90), (2, 3, 9), (16, 2, 100)])
def test_final_pass_and_scaled_readout(gpu, nstates, ncat, npat):
    ec.check_final_pass(gpu, nstates, ncat, npat)


@pytest.mark.parametrize("nstates,ntaxa,npat", [(2, 40, 700), (8, 60, 1500), (16, 30, 300), (40, 30, 400), (5, 20, 200), (33, 12, 100),
                                                (3, 50, 400), (6, 40, 700), (7, 20, 200), (9, 30, 300), (10, 45, 500)])
def test_other_state_counts_on_the_tree_walk(gpu, oracle, nstates, ntaxa, npat):
    """Restriction sites (2 states, CondLikeDown_Bin / Likelihood_Res), covarion nucleotides and amino acids (8 / 40 states,
    CondLikeDown_Gen with TiProbs_GenCov) and the state counts of standard characters (3, 5 ... 10) have their own instantiations of
    the 20/61-state tree-walk kernel; 33 states stay on the level kernels."""
    ec.check_generic_states(gpu, oracle, nstates, ntaxa, npat)


@pytest.mark.parametrize("case", SMALL)
def test_golden_always_rescale(gpu, oracle, golden_dir, case):
    ec.check_golden_case(gpu, oracle, golden_dir, case, lk.MB_BEAGLE_SCALE_ALWAYS)


@pytest.mark.parametrize("case", ["primates_gtr_g4", "synth_dna_gaps", "synth_aa_wag", "replicase_m3"])
def test_golden_dynamic_rescale(gpu, oracle, golden_dir, case):
    ec.check_golden_case(gpu, oracle, golden_dir, case, lk.MB_BEAGLE_SCALE_DYNAMIC)


def test_site_likelihoods(gpu, oracle, golden_dir):
    ec.check_site_likelihoods(gpu, oracle, division_from_golden(golden_dir, "primates_gtr_g4"))
    ec.check_site_likelihoods(gpu, oracle, division_from_golden(golden_dir, "replicase_m3"))
    ec.check_site_likelihoods(gpu, oracle, division_from_golden(golden_dir, "avian_wag_g4"))


@pytest.mark.parametrize("case", ["primates_gtr_g4", "synth_dna_gaps", "avian_wag_g4", "replicase_m3"])
@pytest.mark.parametrize("scaling", [lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC])
def test_partial_update_and_reject(gpu, oracle, golden_dir, case, scaling):
    ec.check_partial_update_and_reject(gpu, oracle, division_from_golden(golden_dir, case), scaling)


def test_multi_chain(gpu, oracle, golden_dir):
    ec.check_multi_chain(gpu, oracle, division_from_golden(golden_dir, "primates_gtr_g4"))
    ec.check_multi_chain(gpu, oracle, division_from_golden(golden_dir, "avian_wag_g4"), nchains=2)


def test_dynamic_rescaling_state_machine(gpu, oracle):
    div = synthetic_division("gtr", 300, 96, seed=31, tree_seed=32)
    ec.check_dynamic_rescaling_state_machine(gpu, oracle, div)


def test_generic_kernels_on_dna(gpu, oracle, golden_dir, monkeypatch):
    monkeypatch.setenv("MBAMD_FORCE_GENERIC", "1")
    ec.check_golden_case(gpu, oracle, golden_dir, "primates_gtr_g4")
    ec.check_single_operations(gpu, oracle, 4, 4, 70)
    ec.check_single_operations(gpu, oracle, 4, 3, 70)


def test_vector_kernels_on_protein_and_codon(gpu, oracle, golden_dir, monkeypatch):
    """MBAMD_NO_MFMA routes 20/61-state data through the VALU general-state kernels: both paths must agree
    with the oracle (and therefore with each other)."""
    monkeypatch.setenv("MBAMD_NO_MFMA", "1")
    ec.check_golden_case(gpu, oracle, golden_dir, "avian_wag_g4")
    ec.check_golden_case(gpu, oracle, golden_dir, "replicase_m3")
    ec.check_single_operations(gpu, oracle, 20, 4, 70)
    ec.check_single_operations(gpu, oracle, 61, 1, 33)


def test_lds_stack_eviction(gpu, oracle, monkeypatch):
    monkeypatch.setenv("MBAMD_MAX_LDS_SLOTS", "3")
    div = synthetic_division("gtr", 60, 130, seed=41, tree_seed=42)
    lnl = ec.engine_lnl(gpu, div)
    want = oracle.tree_loglike(div, use_shortcuts=False)
    assert abs(lnl - want) / abs(want) < ec.REL_FP64
    cat = synthetic_division("gtr", 40, 70, seed=43, tree_seed=44)
    cat.tree = mbtree.caterpillar_tree(40)
    lnl = ec.engine_lnl(gpu, cat)
    want = oracle.tree_loglike(cat, use_shortcuts=False)
    assert abs(lnl - want) / abs(want) < ec.REL_FP64


@pytest.mark.parametrize("waves,slots", [(2, 16), (4, 7), (8, 4), (3, 3)])
def test_walk_waves_agree(gpu, oracle, monkeypatch, waves, slots):
    """The tree walk compiled for 1..8 waves per (pattern block, category) workgroup gives bit-identical partials (the
    arithmetic per node does not depend on which wave executes it or where its inputs were staged), hence identical lnL;
    so do waiting for every memory operation instead of counting (MBAMD_WALK_SAFE) and no look-ahead prefetching."""
    div = synthetic_division("gtr", 150, 1000, seed=51, tree_seed=52, p_gap=0.05)
    monkeypatch.setenv("MBAMD_WALK_WAVES", "1")
    base = ec.engine_lnl(gpu, div)
    monkeypatch.setenv("MBAMD_WALK_WAVES", str(waves))
    monkeypatch.setenv("MBAMD_MAX_LDS_SLOTS", str(slots))
    assert ec.engine_lnl(gpu, div) == base
    monkeypatch.setenv("MBAMD_WALK_PREFETCH", "0")
    assert ec.engine_lnl(gpu, div) == base
    monkeypatch.setenv("MBAMD_WALK_SAFE", "1")
    assert ec.engine_lnl(gpu, div) == base
    want = oracle.tree_loglike(div, use_shortcuts=False)
    assert abs(base - want) / abs(want) < ec.REL_FP64


def test_closed_form_matrices(gpu, oracle):
    ec.check_closed_form_matrices(gpu, oracle)


@pytest.mark.parametrize("case", ["primates_gtr_g4", "synth_dna_gaps", "avian_wag_g4", "replicase_m3"])
@pytest.mark.parametrize("scaling", [lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC])
def test_root_integration_equals_edge_integration(gpu, oracle, golden_dir, case, scaling):
    ec.check_root_equals_edge(gpu, oracle, division_from_golden(golden_dir, case), scaling)


def test_error_codes(gpu):
    inst = bg.BeagleInstance(gpu, 2, 4, 2, 4, 10, 1, 2, 4, 2)
    with pytest.raises(bg.BeagleError) as e:
        inst.update_partials(np.array([[9, -1, -1, 0, 0, 1, 1]], dtype=np.int32))
    assert e.value.code == bg.BEAGLE_ERROR_OUT_OF_RANGE
    with pytest.raises(bg.BeagleError):
        bg.BeagleInstance(gpu, 2, 4, 2, 65, 10, 1, 2, 4, 2)
    inst.finalize()


# ---- BASELINE-size cases ---------------------------------------------------------------------------
def test_config2_dna_500x20k_against_reference(gpu, golden_dir):
    """configs[1]: the engine's lnL on the 500 x 20 000 GTR+G4 alignment vs the lnL the real reference
    printed for the same alignment/tree/parameters (fp64 build and FMA build)."""
    import json
    with open(os.path.join(golden_dir, "synth_dna_500x20k.json")) as fh:
        g = json.load(fh)
    div = division_from_golden(golden_dir, "synth_dna_500x20k")
    assert div.npatterns == g["npatterns"]
    for scaling in (lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC):
        lnl = ec.engine_lnl(gpu, div, scaling)
        assert abs(lnl - g["lnL"]["fp64"]) / abs(g["lnL"]["fp64"]) < ec.REL_FP64, (scaling, lnl)
        assert abs(lnl - g["lnL"]["fma"]) / abs(g["lnL"]["fma"]) < ec.REL_FMA, (scaling, lnl)


@pytest.mark.parametrize("case", ["bench_c2", "bench_c3", "bench_c5", "bench_c4", "bench_c2_gaps", "bench_c3_gaps", "bench_c2_ig_gaps"])
def test_bench_workloads_against_reference(gpu, golden_dir, case):
    """configs[1..4] at FULL size (the bench.py workloads themselves): the engine's lnL vs what the real reference printed
    for the same alignment, tree and parameters -- its fp64 build (2e-6) and its FMA/SSE build (1e-5); both scaling schemes.
    The *_gaps cases are the same shapes with 5 % missing data (the missing-state tip path at full size),
    bench_c2_ig_gaps adds invariable sites (the +I path: per-site values read back and mixed on the host, src/mbbeagle.c:1322-1358).
    Goldens: tools/gen_golden.py bench_c2 bench_c3 bench_c5 bench_c4 bench_c2_gaps bench_c3_gaps bench_c2_ig_gaps."""
    import json
    with open(os.path.join(golden_dir, case + ".json")) as fh:
        g = json.load(fh)
    div = division_from_golden(golden_dir, case)
    assert div.npatterns == g["npatterns"] and div.ntaxa == g["ntaxa"]
    for scaling in (lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC):
        lnl = ec.engine_lnl(gpu, div, scaling)
        assert abs(lnl - g["lnL"]["fp64"]) / abs(g["lnL"]["fp64"]) < ec.REL_FP64, (scaling, lnl, g["lnL"])
        assert abs(lnl - g["lnL"]["fma"]) / abs(g["lnL"]["fma"]) < ec.REL_FMA, (scaling, lnl, g["lnL"])
        # about as close to the fp64 reference as the reference's own fp32 build is (floor: 1e-3 or 2e-8 relative -- the
        # parameter values themselves travel through the .p file with 15 digits and a different eigen-solver)
        assert abs(lnl - g["lnL"]["fp64"]) <= 2.0 * abs(g["lnL"]["fma"] - g["lnL"]["fp64"]) + max(1e-3, 2e-8 * abs(lnl))


def _additivity(gpu, kind, ntaxa, npat, cut, **kw):
    """Site patterns are independent: lnL(all patterns) == lnL(first block) + lnL(rest), and the per-site
    values of the blocks are the corresponding slices (a checksum of checksums at any size)."""
    div = synthetic_division(kind, ntaxa, npat, **kw)
    bd = lk.BeagleDivision(div, gpu)
    whole = bd.LogLike(0)
    site = bd.inst.get_site_log_likelihoods()
    bd.finalize()
    assert np.isfinite(whole)
    assert abs(site.sum() - whole) <= 1e-9 * abs(whole)
    parts = []
    for lo, hi in ((0, cut), (cut, npat)):
        sub = synthetic_division(kind, ntaxa, npat, **kw)
        sub.weights = sub.weights[lo:hi]
        sub.tip_states = [s[lo:hi].copy() for s in sub.tip_states]
        bs = lk.BeagleDivision(sub, gpu)
        parts.append(bs.LogLike(0))
        s2 = bs.inst.get_site_log_likelihoods()
        bs.finalize()
        assert np.array_equal(s2, site[lo:hi])
    assert abs(sum(parts) - whole) <= 1e-9 * abs(whole)
    return div, whole


def test_config2_additivity_and_sample(gpu, oracle):
    div, whole = _additivity(gpu, "gtr", 500, 20000, 7777)
    # oracle on a sample of the patterns (the full problem takes the scalar oracle about a minute)
    sub = synthetic_division("gtr", 500, 20000)
    sub.weights = sub.weights[:1500]
    sub.tip_states = [s[:1500].copy() for s in sub.tip_states]
    bs = lk.BeagleDivision(sub, gpu)
    got = bs.LogLike(0)
    bs.finalize()
    want = oracle.tree_loglike(sub, use_shortcuts=False)
    assert abs(got - want) / abs(want) < ec.REL_FP64


def test_config3_aa_200x10k(gpu, oracle, golden_dir):
    div, whole = _additivity(gpu, "wag", 200, 10000, 3333, seed=5, tree_seed=9, golden_dir=golden_dir)
    sub = synthetic_division("wag", 200, 10000, seed=5, tree_seed=9, golden_dir=golden_dir)
    sub.weights = sub.weights[:300]
    sub.tip_states = [s[:300].copy() for s in sub.tip_states]
    bs = lk.BeagleDivision(sub, gpu)
    got = bs.LogLike(0)
    bs.finalize()
    want = oracle.tree_loglike(sub, use_shortcuts=False)
    assert abs(got - want) / abs(want) < ec.REL_FP64


def test_config5_codon_100x5k(gpu, oracle):
    div, whole = _additivity(gpu, "m3", 100, 5000, 1234, seed=6, tree_seed=10)
    sub = synthetic_division("m3", 100, 5000, seed=6, tree_seed=10)
    sub.weights = sub.weights[:100]
    sub.tip_states = [s[:100].copy() for s in sub.tip_states]
    bs = lk.BeagleDivision(sub, gpu)
    got = bs.LogLike(0)
    bs.finalize()
    want = oracle.tree_loglike(sub, use_shortcuts=False)
    assert abs(got - want) / abs(want) < ec.REL_FP64


def test_config4_dna_1000x50k_two_chains(gpu):
    """configs[3] shape on one GPU: 1000 taxa x 50 000 patterns, two chains in one instance (6.4 GB of
    partials each): the chains agree with each other, a reject restores the previous value exactly, and
    pattern-block additivity holds."""
    div = synthetic_division("gtr", 1000, 50000, seed=6, tree_seed=10)
    bd = lk.BeagleDivision(div, gpu, nchains=2)
    a = bd.LogLike(0)
    b = bd.LogLike(1)
    assert np.isfinite(a) and a == b
    bd.AcceptMove(0)
    t = div.tree
    old = t.length[5]
    t.length[5] = 0.31
    bd.TouchBranch(0, 5)
    c = bd.LogLike(0)
    assert c != a
    t.length[5] = old
    bd.ResetFlips(0)
    assert bd.LogLike(0) == a
    site = bd.inst.get_site_log_likelihoods()
    assert abs(site.sum() - a) <= 1e-9 * abs(a)
    bd.finalize()


# ---- execution-strategy invariance: every alternative way the engine can run a list gives the same bits ----
def _lnl_and_sites(gpu, div, scaling=lk.MB_BEAGLE_SCALE_ALWAYS):
    bd = lk.BeagleDivision(div, gpu, scaling=scaling)
    lnl = bd.LogLike(0)
    site = bd.inst.get_site_log_likelihoods()
    bd.finalize()
    return lnl, site


@pytest.mark.parametrize("case", ["replicase_m3", "avian_wag_g4"])
def test_deferred_lists_equal_immediate(gpu, golden_dir, monkeypatch, case):
    """Lists of several eigen-system parts run merged (one launch per level over all parts) or one by one
    (MBAMD_NO_DEFER): identical results."""
    div = division_from_golden(golden_dir, case)
    a, sa = _lnl_and_sites(gpu, div)
    monkeypatch.setenv("MBAMD_NO_DEFER", "1")
    b, sb = _lnl_and_sites(gpu, div)
    assert a == b and np.array_equal(sa, sb)


@pytest.mark.parametrize("case", ["replicase_m3", "avian_wag_g4", "synth_aa_wag"])
def test_mfma_kernel_variants_agree(gpu, golden_dir, monkeypatch, case):
    """wave-per-factor-tile (default) vs wave-per-tile-column MFMA kernels: same k-ordered fp32 FMA chains."""
    div = division_from_golden(golden_dir, case)
    a, sa = _lnl_and_sites(gpu, div)
    monkeypatch.setenv("MBAMD_MFMA_WHOLE", "1")
    b, sb = _lnl_and_sites(gpu, div)
    assert a == b and np.array_equal(sa, sb)


@pytest.mark.parametrize("case", ["replicase_m3", "avian_wag_g4", "synth_aa_wag"])
@pytest.mark.parametrize("scaling", [lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC])
def test_general_state_schedules_agree(gpu, golden_dir, monkeypatch, case, scaling):
    """The general-state path picks per list between level launches (tip-pair kernel + one wave per factor
    tile), the software-pipelined spine kernel (narrow lists, trailing single-operation levels) and the plain
    serial kernel.  Every choice runs the same k-ordered fp32 chains: bit-identical lnL and site lnLs."""
    div = division_from_golden(golden_dir, case)
    base = _lnl_and_sites(gpu, div, scaling)
    for env in ({"MBAMD_MFMA_SERIAL": "0"},                        # level launches only
                {"MBAMD_MFMA_SERIAL": "100000"},                   # the whole list through the spine kernel
                {"MBAMD_MFMA_SERIAL": "100000", "MBAMD_NO_SPINE": "1"},   # ... through the plain serial kernel
                {"MBAMD_SPINE_WIDTH": "8"}):                       # wider trailing levels join the spine launch
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = _lnl_and_sites(gpu, div, scaling)
        for k in env:
            monkeypatch.delenv(k)
        assert got[0] == base[0] and np.array_equal(got[1], base[1]), env


@pytest.mark.parametrize("kind,ntaxa,npat", [("wag", 60, 700), ("m3", 40, 330)])
def test_general_state_partial_updates_track_full_evaluation(gpu, oracle, kind, ntaxa, npat):
    """A run of single-branch moves (the root-ward path of each is one spine launch; accepted and rejected
    ones mixed) leaves the chain at exactly the lnL a fresh instance computes for the final tree."""
    div = synthetic_division(kind, ntaxa, npat, seed=31, tree_seed=32, p_gap=0.03)
    bd = lk.BeagleDivision(div, gpu, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
    bd.LogLike(0)
    bd.AcceptMove(0)
    t = div.tree
    rng = np.random.default_rng(9)
    nodes = [i for i in range(len(t.anc)) if t.anc[i] != -1 and i != t.root]
    lnl = None
    for rep in range(40):
        b = int(rng.choice(nodes))
        old = t.length[b]
        t.length[b] = old * float(np.exp(0.6 * (rng.random() - 0.5)))
        bd.TouchBranch(0, b)
        lnl = bd.LogLike(0)
        if rep % 3 == 2:                                           # reject: restore the branch, flip back
            t.length[b] = old
            bd.ResetFlips(0)
            lnl = None
        else:
            bd.AcceptMove(0)
    if lnl is None:
        bd.TouchBranch(0, nodes[0])
        lnl = bd.LogLike(0)
        bd.AcceptMove(0)
    bd.finalize()
    fresh = lk.BeagleDivision(div, gpu, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
    want = fresh.LogLike(0)
    fresh.finalize()
    assert lnl == pytest.approx(want, rel=2e-7)
    ref = oracle.tree_loglike(div, use_shortcuts=False)
    assert abs(lnl - ref) / abs(ref) < 2e-6


@pytest.mark.parametrize("ncat", [1, 2, 3])
def test_protein_other_category_counts(gpu, oracle, ncat):
    """WAG with 1-3 gamma categories: every category count takes its own mix of kernels (3 has no serial /
    tip-pair / per-factor-tile instance and runs on the wave-per-tile kernel); full evaluation + a partial update."""
    div = synthetic_division("wag", 40, 330, seed=41, tree_seed=42, p_gap=0.03, ncat=ncat)
    ec.check_partial_update_and_reject(gpu, oracle, div, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
    ec.check_partial_update_and_reject(gpu, oracle, div, scaling=lk.MB_BEAGLE_SCALE_ALWAYS)


@pytest.mark.parametrize("kind", ["gtr", "wag", "m3"])
def test_device_eigen(gpu, kind):
    """Eigen-systems from rate matrices on the device (mbamdSetRateMatrices, k_eigen_reversible)."""
    ec.check_device_eigen(gpu, kind)


@pytest.mark.parametrize("nstates", [4, 20, 61])
def test_device_eigen_warm_start(gpu, nstates):
    ec.check_device_eigen_warm_start(gpu, nstates)


@pytest.mark.parametrize("case", ["replicase_m3", "avian_wag_g4", "synth_aa_wag"])
def test_general_state_tree_walk_schedules_agree(gpu, golden_dir, monkeypatch, case):
    """20/61-state tree walk (k_walkg): waves per workgroup, LDS slots (children re-read from HBM instead), lists run one by
    one instead of merged -- every schedule gives the same bits; the level kernels (MBAMD_NO_WALKG=1: a scaler per pattern
    instead of per pattern and category, another summation order) agree to rounding."""
    div = division_from_golden(golden_dir, case)
    a, sa = _lnl_and_sites(gpu, div)
    for env in ({"MBAMD_WALK_WAVES": "1"}, {"MBAMD_WALK_WAVES": "4"}, {"MBAMD_WALK_WAVES": "2", "MBAMD_MAX_LDS_SLOTS": "3"},
                {"MBAMD_NO_DEFER": "1"}, {"MBAMD_NO_INLINE_PROGRAMS": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        b, sb = _lnl_and_sites(gpu, div)
        assert a == b and np.array_equal(sa, sb), env
        b, sb = _lnl_and_sites(gpu, div, lk.MB_BEAGLE_SCALE_DYNAMIC)
        assert abs(a - b) <= 1e-9 * abs(a)
        for k in env:
            monkeypatch.delenv(k)
    monkeypatch.setenv("MBAMD_NO_WALKG", "1")
    b, sb = _lnl_and_sites(gpu, div)
    assert abs(a - b) <= 2e-7 * abs(a)
    assert np.allclose(sa, sb, rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("nstates,ntaxa,npat", [(2, 60, 900), (8, 50, 700), (40, 24, 300)])
def test_new_state_counts_schedules_agree(gpu, monkeypatch, nstates, ntaxa, npat):
    """The 2-, 8- and 40-state instantiations of k_walkg: every schedule (waves per workgroup, slot budget) gives the same bits, and
    the level kernels (MBAMD_NO_WALKG=1) agree to rounding -- per site."""
    div = synthetic_division("gen%d" % nstates, ntaxa, npat, seed=13, tree_seed=6, alpha=0.6, ncat=4, p_gap=0.03)
    a, sa = _lnl_and_sites(gpu, div)
    for env in ({"MBAMD_WALK_WAVES": "1"}, {"MBAMD_WALK_WAVES": "4"}, {"MBAMD_WALK_WAVES": "2", "MBAMD_MAX_LDS_SLOTS": "3"}, {"MBAMD_NO_INLINE_PROGRAMS": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        b, sb = _lnl_and_sites(gpu, div)
        assert a == b and np.array_equal(sa, sb), (nstates, env)
        for k in env:
            monkeypatch.delenv(k)
    monkeypatch.setenv("MBAMD_NO_WALKG", "1")
    b, sb = _lnl_and_sites(gpu, div)
    assert abs(a - b) <= 2e-7 * abs(a)
    assert np.allclose(sa, sb, rtol=2e-6, atol=2e-5)


def test_deferred_lists_without_a_merge_kernel(gpu, oracle):
    """Two independent operation lists back to back on a shape that has no merged-launch kernel (20 states, three
    categories): the deferred lists must run one after the other, not be dropped."""
    rng = np.random.default_rng(11)
    S, K, P = 20, 3, 70
    inst = bg.BeagleInstance(gpu, 2, 8, 2, S, P, 1, 4, K, 4)
    try:
        ti = rng.random((K, S, S)) + 0.05
        ti /= ti.sum(axis=2, keepdims=True)
        for m in range(2):
            inst.set_transition_matrix(m, ti)
        st1 = rng.integers(0, S + 1, size=P).astype(np.int32)
        st2 = rng.integers(0, S + 1, size=P).astype(np.int32)
        inst.set_tip_states(0, st1)
        inst.set_tip_states(1, st2)
        t = np.ascontiguousarray(ti, dtype=np.float32)
        want = oracle.condlike_down(S, K, P, None, st1, t, None, st2, t)
        inst.update_partials(np.array([[4, -1, -1, 0, 0, 1, 1]], dtype=np.int32), bg.BEAGLE_OP_NONE)
        inst.update_partials(np.array([[5, -1, -1, 1, 1, 0, 0]], dtype=np.int32), bg.BEAGLE_OP_NONE)   # independent of the first
        assert ec.close_partials(inst.get_partials(4), want)
        assert ec.close_partials(inst.get_partials(5), oracle.condlike_down(S, K, P, None, st2, t, None, st1, t))
    finally:
        inst.finalize()


def test_walk_counted_waits_agree_on_partial_updates(gpu, oracle, monkeypatch):
    """Root-ward paths (every operation prefetches its sibling from HBM by LDS-DMA, waits are by count): same bits as
    with every wait draining the memory queue, for several prefetch distances."""
    div = synthetic_division("gtr", 120, 700, seed=71, tree_seed=72, p_gap=0.04)

    def path_values():
        bd = lk.BeagleDivision(div, gpu)
        vals = [bd.LogLike(0)]
        bd.AcceptMove(0)
        for node in (3, 17, 60, 101):
            div.tree.length[node] *= 1.7
            bd.TouchBranch(0, node)
            vals.append(bd.LogLike(0))
            bd.AcceptMove(0)
        for node in (3, 17, 60, 101):
            div.tree.length[node] /= 1.7
        bd.finalize()
        return vals

    base = path_values()
    # ... and with the short programs in a device buffer instead of the kernel arguments (k_walk4_t<Walk4ArgsInline>)
    # ... and with the result waited for by hipStreamSynchronize instead of the polled word (MBAMD_NO_POLL)
    for env in ({"MBAMD_WALK_SAFE": "1"}, {"MBAMD_WALK_PREFETCH": "0"}, {"MBAMD_WALK_PREFETCH": "9"}, {"MBAMD_MAX_LDS_SLOTS": "3"},
                {"MBAMD_NO_INLINE_PROGRAMS": "1"}, {"MBAMD_NO_POLL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert path_values() == base
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("ncat", [1, 2, 5, 16])
def test_dna_other_category_counts(gpu, oracle, ncat):
    div = synthetic_division("gtr", 30, 140, seed=45, tree_seed=46, p_gap=0.03, ncat=ncat)
    ec.check_partial_update_and_reject(gpu, oracle, div, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
    ec.check_partial_update_and_reject(gpu, oracle, div, scaling=lk.MB_BEAGLE_SCALE_ALWAYS)


def test_lists_with_hazards_are_cut_into_segments(gpu, oracle):
    ec.check_hazard_lists(gpu, 4, 4, 100)
    ec.check_hazard_lists(gpu, 20, 4, 40)
    ec.check_hazard_lists(gpu, 8, 4, 70)
    ec.check_hazard_lists(gpu, 2, 2, 40)
    ec.check_hazard_lists(gpu, 40, 2, 40)
    ec.check_hazard_lists(gpu, 4, 4, 100, double_precision=True)      # (the fp64 walk hands lists with hazards to the level kernels)
    ec.check_hazard_lists(gpu, 20, 2, 40, double_precision=True)


def test_instances_created_and_destroyed_repeatedly(gpu):
    """Buffers land at different device addresses every time (address-dependent bugs, e.g. a pointer whose low
    half has bit 31 set, show up here); results must not move."""
    div = synthetic_division("gtr", 150, 1000, seed=51, tree_seed=52, p_gap=0.05)
    vals = set()
    keep = []
    for rep in range(6):
        bd = lk.BeagleDivision(div, gpu, nchains=1 + rep % 3)
        vals.add(bd.LogLike(0))
        if rep % 2:
            keep.append(bd)                  # leave some alive so the allocator moves on
        else:
            bd.finalize()
    for bd in keep:
        bd.finalize()
    assert len(vals) == 1


@pytest.mark.parametrize("kind,ntaxa,npat", [("gtr", 40, 700), ("wag", 20, 300)])
def test_patterns_sharded_inside_one_instance(gpu, oracle, golden_dir, monkeypatch, kind, ntaxa, npat):
    div = synthetic_division(kind, ntaxa, npat, seed=81, tree_seed=82, p_gap=0.03, golden_dir=golden_dir)
    ec.check_sharded_instance(gpu, oracle, div, monkeypatch, shards=3)


@pytest.mark.parametrize("double_precision", [False, True])
@pytest.mark.parametrize("kind", ["gtr", "wag"])
def test_multi_partition_instance(gpu, oracle, golden_dir, kind, double_precision):
    a = synthetic_division(kind, 24, 330, seed=91, tree_seed=92, p_gap=0.02, golden_dir=golden_dir, alpha=0.5)
    b = synthetic_division(kind, 24, 150, seed=93, tree_seed=94, p_gap=0.02, golden_dir=golden_dir, alpha=1.7, brlen=0.11)
    ec.check_multi_partition_instance(gpu, oracle, a, b, double_precision)


@pytest.mark.parametrize("ntaxa,npat,nstates,words", [(12, 100, 4, 1), (200, 5000, 4, 1), (50, 1000, 20, 1), (30, 700, 61, 1), (7, 65, 64, 1),
                                                       (10, 50, 10, 1), (16, 300, 70, 2), (500, 20000, 4, 1)])
def test_parsimony(gpu, ntaxa, npat, nstates, words):
    """Device Fitch parsimony (mbamdPars*) against the oracle: u8 / u16 / u32 / u64 / 2 x u64 sets."""
    ec.check_parsimony(gpu, ntaxa, npat, nstates, words=words)


@pytest.mark.parametrize("waves", [1, 2, 4, 8])
def test_parsimony_waves_per_workgroup(gpu, monkeypatch, waves):
    """The parsimony walk with the tree cut over 1 / 2 / 4 / 8 waves of a workgroup (phases separated by barriers, ParsInstance::flush):
    every setting gives the oracle's sets and lengths exactly."""
    monkeypatch.setenv("MBAMD_PARS_WAVES", str(waves))
    ec.check_parsimony(gpu, 70, 130, 4, seed=5)
    if waves == 8:                       # ... and with a program cut into launches of two phases each
        monkeypatch.setenv("MBAMD_PARS_PHASE_LIMIT", "2")
        ec.check_parsimony(gpu, 70, 130, 4, seed=8)
        monkeypatch.delenv("MBAMD_PARS_PHASE_LIMIT")
    ec.check_parsimony(gpu, 33, 64, 20, seed=6)
    ec.check_parsimony(gpu, 150, 2000, 61, seed=7)


@pytest.mark.parametrize("case", ["primates_gtr_g4", "primates_gtr_ig4", "avian_wag_g4", "replicase_m3", "synth_dna_gaps", "synth_aa_wag",
                                  "synth_codon_m3", "bench_c2", "bench_c3", "bench_c5"])
def test_double_precision(gpu, golden_dir, case):
    """BEAGLE_FLAG_PRECISION_DOUBLE: the fp64 engine (mbamd_f64.h) against the reference's double build."""
    ec.check_double_precision(gpu, golden_dir, case)


@pytest.mark.parametrize("case", ["primates_gtr_g4", "synth_dna_gaps", "bench_c2"])
def test_double_precision_walk_equals_levels(gpu, golden_dir, case, monkeypatch):
    """fp64, four states: k64_walk4 (one launch per operation list) against the level kernels, bit for bit."""
    ec.check_double_precision_walk(gpu, golden_dir, case, monkeypatch)


@pytest.mark.parametrize("case", ["synth_codon_m3", "replicase_m3", "bench_c5", "synth_aa_wag", "avian_wag_g4"])
def test_double_precision_general_state_paths(gpu, golden_dir, case, monkeypatch):
    """fp64, 16 ... 64 states: LDS-staged kernels, queued matrix updates and the staging ring against the plain level kernels, bit for bit."""
    ec.check_double_precision_general_paths(gpu, golden_dir, case, monkeypatch)


def test_double_precision_walk_category_counts(gpu, monkeypatch):
    """fp64, four states: every category count 1 ... 8 on the walk (categories in the lanes of one wave) against the level kernels."""
    ec.check_double_precision_walk_categories(gpu, monkeypatch)


@pytest.mark.parametrize("nstates", [4, 20])
def test_double_precision_queued_lists(gpu, nstates):
    """fp64: operation lists are queued and run together; errors are reported by the call that brought the list."""
    ec.check_double_precision_queue(gpu, nstates=nstates)


@pytest.mark.parametrize("nstates", [4, 20, 61])
def test_double_precision_queued_matrix_updates(gpu, monkeypatch, nstates):
    """fp64: transition-matrix updates of several calls run as one launch; staging ring; unchanged weights / frequencies are not re-sent."""
    ec.check_double_precision_matrix_queue(gpu, monkeypatch, nstates=nstates)


def test_parsimony_model_golden(gpu, golden_dir):
    """device Fitch lengths == the reference's own parsimony-model likelihood (golden vectors from oracle/_ref/mb)"""
    ec.check_parsimony_model_golden(gpu, golden_dir)


@pytest.mark.parametrize("kind,ntaxa,npat", [("gtr", 120, 700), ("wag", 60, 300), ("m3", 30, 120)])
def test_short_walk_programs_travel_in_the_kernel_arguments(gpu, monkeypatch, golden_dir, kind, ntaxa, npat):
    """Partial updates (root-ward paths) compile into programs of a few dozen entries: they are handed to the kernel in its
    arguments (k_walk4_t<Walk4ArgsInline>, k_walkg<..., WalkGArgsInline>) instead of a device buffer -- same bits either way."""
    div = synthetic_division(kind, ntaxa, npat, seed=71, tree_seed=72, p_gap=0.04, golden_dir=golden_dir)

    def path_values():
        bd = lk.BeagleDivision(div, gpu, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
        vals = [bd.LogLike(0)]
        bd.AcceptMove(0)
        nodes = [n for n in (3, 7, 17, 44, 70) if n < div.tree.n_nodes and div.tree.anc[n] >= 0]
        for node in nodes:
            div.tree.length[node] *= 1.7
            bd.TouchBranch(0, node)
            vals.append(bd.LogLike(0))
            bd.AcceptMove(0)
        for node in nodes:
            div.tree.length[node] /= 1.7
        bd.finalize()
        return vals

    base = path_values()
    monkeypatch.setenv("MBAMD_NO_INLINE_PROGRAMS", "1")
    assert path_values() == base


def test_general_state_path_kernel(gpu, oracle, golden_dir, monkeypatch):
    """A move's root-ward path at 20 / 61 states: k_pathg == k_walkg bit for bit, both == the oracle."""
    ec.check_general_state_path_kernel(gpu, oracle, golden_dir, monkeypatch, cases=("avian_wag_g4", "synth_aa_wag", "replicase_m3", "synth_codon_m3"))


@pytest.mark.parametrize("case,switch", [("bench_c2", "MBAMD_NO_PATH4"), ("bench_c3", "MBAMD_NO_PATHG"), ("bench_c5", "MBAMD_NO_PATHG")])
def test_path_kernels_at_baseline_shapes(gpu, golden_dir, monkeypatch, case, switch):
    """k_path4 / k_pathg pinned where they were built to run: DNA 500 x 20 000, protein 200 x 10 000, codon M3 100 x 5 000 --
    bit-equal to the whole-tree walk kernels on the same lists and within REL_FP64 of the fp64 engine after every move
    (reference: the partial updates of src/mbbeagle.c:783-880)."""
    ec.check_path_kernels_at_bench_shape(gpu, golden_dir, monkeypatch, case, switch)


def test_path_and_log_likelihood_in_one_launch(gpu, oracle, monkeypatch, golden_dir):
    """k_path4_lnl == k_path4 + k_integrate_lnl_s4 bit for bit (a fixed-topology generation is one launch behind its matrices)."""
    ec.check_fused_path_and_likelihood(gpu, oracle, monkeypatch)
    ec.check_fused_path_and_likelihood(gpu, None, monkeypatch, golden_dir=golden_dir, case="bench_c2")


def test_paths_that_join_run_as_arms(gpu, oracle, monkeypatch, golden_dir):
    """The lists of topology moves (two dirty branches: two root-ward paths and their common stem) on the path kernel == the same
    lists on the tree-walk kernel, bit for bit -- small against the oracle, at DNA 500 x 20 000 against the fp64 engine."""
    ec.check_forked_paths(gpu, oracle, monkeypatch)
    ec.check_forked_paths(gpu, None, monkeypatch, golden_dir=golden_dir, case="bench_c2")

