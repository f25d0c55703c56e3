"""Host logic of the engine, run on the CPU through the TEST-ONLY host-emulation build
(tests/hostemu/): BEAGLE call protocol, buffer/index bookkeeping, operation scheduling, LDS-slot
allocation of the tree-walk path, layout conversions, scaling arithmetic, dynamic-rescale protocol.
The GPU twin of these checks is tests/test_engine_gpu.py (-m gpu), which runs the product library."""
import os

import numpy as np
import pytest

from mrbayes_amd import beagle as bg
from mrbayes_amd import likelihood as lk
from mrbayes_amd import tree as mbtree
from mrbayes_amd.division import division_from_golden, synthetic_division
from tests import engine_checks as ec
from tests.hostemu import build_emu

SMALL = ["primates_gtr_g4", "primates_gtr_ig4", "primates_gtr_equal", "avian_wag_g4", "replicase_m3",
         "synth_dna_gaps", "synth_aa_wag", "synth_codon_m3"]


@pytest.fixture(scope="module")
def emu():
    return bg.library(build_emu.build())


def test_emu_is_not_the_product(emu):
    assert "TESTONLY" in emu.path
    assert emu.resources()[0][0].startswith("host emulation")


def test_product_sources_have_no_emulation_fork():
    """The shipped sources carry no host-emulation switch: the emulation is a set of same-named device-primitive headers
    (tests/hostemu/mbamd_dev_*.h) that the TEST build puts in front on its include path; every product header of that family
    (mrbayes_amd/csrc/device/) has a twin, and nothing under mrbayes_amd/, include/ or integration/ mentions the emulation macro."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for top in ("mrbayes_amd", "include", "integration"):
        for d, _, files in os.walk(os.path.join(root, top)):
            for f in files:
                if f.endswith((".h", ".cpp", ".c", ".py")):
                    text = open(os.path.join(d, f), errors="replace").read()
                    assert "MBAMD_HOST_EMU" not in text and "hip_emu.h" not in text, os.path.join(d, f)
    dev = os.path.join(root, "mrbayes_amd", "csrc", "device")
    for f in os.listdir(dev):
        assert os.path.exists(os.path.join(root, "tests", "hostemu", f)), "no emulation twin for " + f


@pytest.mark.parametrize("case", ["primates_gtr_g4", "avian_wag_g4", "replicase_m3"])
def test_transition_matrices(emu, oracle, golden_dir, case):
    ec.check_transition_matrices(emu, oracle, division_from_golden(golden_dir, case))


@pytest.mark.parametrize("nstates,ncat,npat", [(4, 4, 100), (4, 1, 64), (4, 3, 65), (20, 4, 70), (61, 1, 33),
                                                 (16, 2, 10), (2, 4, 5), (60, 1, 45), (63, 2, 33), (16, 4, 40)])
def test_single_operations(emu, oracle, nstates, ncat, npat):
    ec.check_single_operations(emu, oracle, nstates, ncat, npat)


@pytest.mark.parametrize("nstates,ncat,npat", [(4, 4, 100), (4, 1, 65), (20, 4, 70), (61, 1, 33), (8, 4, 50), (2, 3, 9)])
def test_final_pass_and_scaled_readout(emu, nstates, ncat, npat):
    ec.check_final_pass(emu, nstates, ncat, npat)


@pytest.mark.parametrize("nstates,ntaxa,npat", [(2, 12, 70), (8, 10, 45), (40, 8, 40), (5, 8, 33), (3, 9, 40), (6, 10, 70), (7, 8, 33), (9, 8, 40),
                                                (10, 9, 65), (33, 8, 33)])
def test_other_state_counts_on_the_tree_walk(emu, oracle, nstates, ntaxa, npat):
    """Restriction sites (2 states, CondLikeDown_Bin / Likelihood_Res), covarion nucleotides and amino acids (8 / 40 states,
    CondLikeDown_Gen with TiProbs_GenCov) and the state counts of standard characters (3, 5 ... 10) have their own instantiations of
    the 20/61-state tree-walk kernel; 33 states stay on the level kernels."""
    ec.check_generic_states(emu, oracle, nstates, ntaxa, npat)


@pytest.mark.parametrize("case", SMALL)
def test_golden_always_rescale(emu, oracle, golden_dir, case):
    ec.check_golden_case(emu, oracle, golden_dir, case, lk.MB_BEAGLE_SCALE_ALWAYS)


@pytest.mark.parametrize("case", ["primates_gtr_g4", "synth_dna_gaps", "synth_aa_wag"])
def test_golden_dynamic_rescale(emu, oracle, golden_dir, case):
    ec.check_golden_case(emu, oracle, golden_dir, case, lk.MB_BEAGLE_SCALE_DYNAMIC)


def test_site_likelihoods(emu, oracle, golden_dir):
    ec.check_site_likelihoods(emu, oracle, division_from_golden(golden_dir, "primates_gtr_g4"))
    ec.check_site_likelihoods(emu, oracle, division_from_golden(golden_dir, "replicase_m3"))


@pytest.mark.parametrize("case", ["primates_gtr_g4", "synth_dna_gaps", "avian_wag_g4"])
@pytest.mark.parametrize("scaling", [lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC])
def test_partial_update_and_reject(emu, oracle, golden_dir, case, scaling):
    ec.check_partial_update_and_reject(emu, oracle, division_from_golden(golden_dir, case), scaling)


def test_multi_chain(emu, oracle, golden_dir):
    ec.check_multi_chain(emu, oracle, division_from_golden(golden_dir, "primates_gtr_g4"))


def test_dynamic_rescaling_state_machine(emu, oracle):
    div = synthetic_division("gtr", 300, 96, seed=31, tree_seed=32)
    ec.check_dynamic_rescaling_state_machine(emu, oracle, div)


def test_generic_kernels_on_dna(emu, oracle, golden_dir, monkeypatch):
    """MBAMD_FORCE_GENERIC routes 4-state data through the general-state kernels (tile-major layout)."""
    monkeypatch.setenv("MBAMD_FORCE_GENERIC", "1")
    ec.check_golden_case(emu, oracle, golden_dir, "primates_gtr_g4")
    ec.check_golden_case(emu, oracle, golden_dir, "primates_gtr_equal")
    ec.check_single_operations(emu, oracle, 4, 4, 70)
    ec.check_single_operations(emu, oracle, 4, 3, 70)      # unfused (two-pass) variant


def test_lds_stack_eviction(emu, oracle, monkeypatch):
    """With only three LDS slots per wave the tree-walk scheduler must evict results and prefetch its own stores back."""
    monkeypatch.setenv("MBAMD_MAX_LDS_SLOTS", "3")
    div = synthetic_division("gtr", 60, 130, seed=41, tree_seed=42)
    lnl = ec.engine_lnl(emu, div)
    want = oracle.tree_loglike(div, use_shortcuts=False)
    assert abs(lnl - want) / abs(want) < ec.REL_FP64
    cat = synthetic_division("gtr", 40, 70, seed=43, tree_seed=44)
    cat.tree = mbtree.caterpillar_tree(40)
    lnl = ec.engine_lnl(emu, cat)
    want = oracle.tree_loglike(cat, use_shortcuts=False)
    assert abs(lnl - want) / abs(want) < ec.REL_FP64


def test_error_codes(emu):
    inst = bg.BeagleInstance(emu, 2, 4, 2, 4, 10, 1, 2, 4, 2)
    with pytest.raises(bg.BeagleError) as e:
        inst.update_partials(np.array([[9, -1, -1, 0, 0, 1, 1]], dtype=np.int32))
    assert e.value.code == bg.BEAGLE_ERROR_OUT_OF_RANGE
    with pytest.raises(bg.BeagleError) as e:
        inst.update_partials(np.array([[3, -1, -1, 0, 0, 1, 1]], dtype=np.int32))   # children never written
    assert e.value.code == bg.BEAGLE_ERROR_OUT_OF_RANGE
    with pytest.raises(bg.BeagleError):
        bg.BeagleInstance(emu, 2, 4, 2, 65, 10, 1, 2, 4, 2)                         # > 64 states
    inst.finalize()


@pytest.mark.parametrize("waves", [2, 3, 4, 8])
@pytest.mark.parametrize("slots", [3, 4, 7, 16])
@pytest.mark.parametrize("prefetch", [0, 4])
def test_walk_schedule(emu, oracle, monkeypatch, waves, slots, prefetch):
    """Per-wave programs of the tree-walk path (host logic: subtree bins, barrier phases, slot allocation, prefetches of
    values that cross waves): the result is bit-identical for every W / slot budget / prefetch distance, because each
    node's arithmetic does not depend on which wave runs it or where its inputs were staged.  (The host emulation
    runs the waves of a workgroup as fibers that meet at the barriers.)"""
    div = synthetic_division("gtr", 90, 130, seed=61, tree_seed=62, p_gap=0.03)
    monkeypatch.setenv("MBAMD_WALK_WAVES", "1")
    base = ec.engine_lnl(emu, div)
    monkeypatch.setenv("MBAMD_WALK_WAVES", str(waves))
    monkeypatch.setenv("MBAMD_MAX_LDS_SLOTS", str(slots))
    monkeypatch.setenv("MBAMD_WALK_PREFETCH", str(prefetch))
    monkeypatch.setenv("MBAMD_WALK_SMALL_PHASE", "4")
    assert ec.engine_lnl(emu, div) == base
    bd = lk.BeagleDivision(div, emu)
    bd.LogLike(0)
    bd.AcceptMove(0)
    t = div.tree
    t.length[3] *= 2.0
    bd.TouchBranch(0, 3)
    got = bd.LogLike(0)
    bd.finalize()
    want = oracle.tree_loglike(div, use_shortcuts=False)
    t.length[3] /= 2.0
    assert abs(got - want) / abs(want) < ec.REL_FP64


@pytest.mark.parametrize("kind,ntaxa,npat", [("gtr", 60, 200), ("wag", 30, 70), ("m3", 16, 40)])
def test_short_walk_programs_travel_in_the_kernel_arguments(emu, monkeypatch, golden_dir, kind, ntaxa, npat):
    """Partial updates (root-ward paths) compile into programs of a few dozen entries: they are handed to the kernel in its
    arguments (k_walk4_t<Walk4ArgsInline>, k_walkg<..., WalkGArgsInline>) instead of a device buffer -- same bits either way."""
    div = synthetic_division(kind, ntaxa, npat, seed=71, tree_seed=72, p_gap=0.04, golden_dir=golden_dir)

    def path_values():
        bd = lk.BeagleDivision(div, emu, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
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
    monkeypatch.setenv("MBAMD_NO_POLL", "1")                 # the result waited for by the runtime instead of the polled word
    assert path_values() == base


@pytest.mark.parametrize("kind", ["gtr", "wag", "m3"])
def test_device_eigen(emu, kind):
    ec.check_device_eigen(emu, kind)


@pytest.mark.parametrize("nstates", [4, 20, 61])
def test_device_eigen_warm_start(emu, nstates):
    ec.check_device_eigen_warm_start(emu, nstates)


@pytest.mark.parametrize("model", ["wag", "m3"])
@pytest.mark.parametrize("waves,slots", [(2, 3), (4, 3), (8, 4), (2, 12)])
def test_general_walk_schedule(emu, oracle, monkeypatch, model, waves, slots):
    """The same for the 20/61-state tree walk (k_walkg): children that are not in an LDS slot are re-read from HBM by the
    kernel's operand pipeline; the eigen-system parts of the codon model run as separate workgroups or -- with
    MBAMD_NO_DEFER -- one list at a time.  Every schedule gives the same bits."""
    div = synthetic_division(model, 40, 70, seed=63, tree_seed=64, p_gap=0.03)
    monkeypatch.setenv("MBAMD_WALK_WAVES", "1")
    base = ec.engine_lnl(emu, div)
    want = oracle.tree_loglike(div, use_shortcuts=False)
    assert abs(base - want) / abs(want) < ec.REL_FP64
    monkeypatch.setenv("MBAMD_WALK_WAVES", str(waves))
    monkeypatch.setenv("MBAMD_MAX_LDS_SLOTS", str(slots))
    monkeypatch.setenv("MBAMD_WALK_SMALL_PHASE", "4")
    assert ec.engine_lnl(emu, div) == base
    monkeypatch.setenv("MBAMD_NO_DEFER", "1")
    assert ec.engine_lnl(emu, div) == base
    monkeypatch.delenv("MBAMD_NO_DEFER")
    bd = lk.BeagleDivision(div, emu, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
    bd.LogLike(0)
    bd.AcceptMove(0)
    t = div.tree
    t.length[3] *= 2.0
    bd.TouchBranch(0, 3)
    got = bd.LogLike(0)
    bd.finalize()
    want = oracle.tree_loglike(div, use_shortcuts=False)
    t.length[3] /= 2.0
    assert abs(got - want) / abs(want) < ec.REL_FP64


@pytest.mark.parametrize("ncat", [1, 2, 5, 16])
def test_dna_other_category_counts(emu, oracle, ncat):
    """The tree walk runs one wave per (pattern block, category): any category count."""
    div = synthetic_division("gtr", 30, 140, seed=45, tree_seed=46, p_gap=0.03, ncat=ncat)
    ec.check_partial_update_and_reject(emu, oracle, div, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
    ec.check_partial_update_and_reject(emu, oracle, div, scaling=lk.MB_BEAGLE_SCALE_ALWAYS)


def test_lists_with_hazards_are_cut_into_segments(emu, oracle):
    ec.check_hazard_lists(emu, 4, 4, 100)
    ec.check_hazard_lists(emu, 20, 4, 40)
    ec.check_hazard_lists(emu, 8, 4, 70)
    ec.check_hazard_lists(emu, 2, 2, 40)
    ec.check_hazard_lists(emu, 40, 2, 40)
    ec.check_hazard_lists(emu, 4, 4, 100, double_precision=True)      # (the fp64 walk hands lists with hazards to the level kernels)
    ec.check_hazard_lists(emu, 20, 2, 40, double_precision=True)


def test_closed_form_matrices(emu, oracle):
    ec.check_closed_form_matrices(emu, oracle)


@pytest.mark.parametrize("case", ["primates_gtr_g4", "avian_wag_g4", "replicase_m3"])
def test_root_integration_equals_edge_integration(emu, oracle, golden_dir, case):
    ec.check_root_equals_edge(emu, oracle, division_from_golden(golden_dir, case))


@pytest.mark.parametrize("kind,ntaxa,npat", [("gtr", 40, 700), ("wag", 20, 300)])
def test_patterns_sharded_inside_one_instance(emu, oracle, golden_dir, monkeypatch, kind, ntaxa, npat):
    div = synthetic_division(kind, ntaxa, npat, seed=81, tree_seed=82, p_gap=0.03, golden_dir=golden_dir)
    ec.check_sharded_instance(emu, oracle, div, monkeypatch, shards=3)


@pytest.mark.parametrize("double_precision", [False, True])
@pytest.mark.parametrize("kind", ["gtr", "wag"])
def test_multi_partition_instance(emu, oracle, golden_dir, kind, double_precision):
    a = synthetic_division(kind, 24, 330, seed=91, tree_seed=92, p_gap=0.02, golden_dir=golden_dir, alpha=0.5)
    b = synthetic_division(kind, 24, 150, seed=93, tree_seed=94, p_gap=0.02, golden_dir=golden_dir, alpha=1.7, brlen=0.11)
    ec.check_multi_partition_instance(emu, oracle, a, b, double_precision)


@pytest.mark.parametrize("ntaxa,npat,nstates,words", [(12, 100, 4, 1), (30, 333, 4, 1), (9, 64, 20, 1), (8, 70, 61, 1), (7, 65, 64, 1),
                                                       (10, 50, 10, 1), (6, 40, 70, 2),
                                                       (130, 200, 4, 1), (45, 100, 20, 1), (100, 70, 61, 1)])
def test_parsimony(emu, ntaxa, npat, nstates, words):
    """Device Fitch parsimony (mbamdPars*) against the oracle: u8 / u16 / u32 / u64 / 2 x u64 sets."""
    ec.check_parsimony(emu, ntaxa, npat, nstates, words=words)


@pytest.mark.parametrize("waves", [1, 2, 4, 8])
def test_parsimony_waves_per_workgroup(emu, monkeypatch, waves):
    """The parsimony walk with the tree cut over 1 / 2 / 4 / 8 waves of a workgroup (phases separated by barriers, ParsInstance::flush):
    every setting gives the oracle's sets and lengths exactly."""
    monkeypatch.setenv("MBAMD_PARS_WAVES", str(waves))
    ec.check_parsimony(emu, 70, 130, 4, seed=5)
    if waves == 8:                       # ... and with a program cut into launches of two phases each
        monkeypatch.setenv("MBAMD_PARS_PHASE_LIMIT", "2")
        ec.check_parsimony(emu, 70, 130, 4, seed=8)
        monkeypatch.delenv("MBAMD_PARS_PHASE_LIMIT")
    ec.check_parsimony(emu, 33, 64, 20, seed=6)


@pytest.mark.parametrize("case", ["primates_gtr_g4", "primates_gtr_ig4", "avian_wag_g4", "replicase_m3", "synth_dna_gaps"])
def test_double_precision(emu, golden_dir, case):
    """BEAGLE_FLAG_PRECISION_DOUBLE: the fp64 engine (mbamd_f64.h) against the reference's double build."""
    ec.check_double_precision(emu, golden_dir, case)


@pytest.mark.parametrize("case", ["primates_gtr_g4", "synth_dna_gaps"])
def test_double_precision_walk_equals_levels(emu, golden_dir, case, monkeypatch):
    """fp64, four states: k64_walk4 (one launch per operation list) against the level kernels, bit for bit."""
    ec.check_double_precision_walk(emu, golden_dir, case, monkeypatch)


def test_double_precision_walk_category_counts(emu, monkeypatch):
    """fp64, four states: every category count 1 ... 8 on the walk (categories in the lanes of one wave) against the level kernels."""
    ec.check_double_precision_walk_categories(emu, monkeypatch)


@pytest.mark.parametrize("nstates", [4, 20])
def test_double_precision_queued_lists(emu, nstates):
    """fp64: operation lists are queued and run together; errors are reported by the call that brought the list."""
    ec.check_double_precision_queue(emu, nstates=nstates)


@pytest.mark.parametrize("nstates", [4, 20, 61])
def test_double_precision_queued_matrix_updates(emu, monkeypatch, nstates):
    """fp64: transition-matrix updates of several calls run as one launch; staging ring; unchanged weights / frequencies are not re-sent."""
    ec.check_double_precision_matrix_queue(emu, monkeypatch, nstates=nstates)


def test_parsimony_model_golden(emu, golden_dir):
    """device Fitch lengths == the reference's own parsimony-model likelihood (golden vectors from oracle/_ref/mb)"""
    ec.check_parsimony_model_golden(emu, golden_dir)


def test_general_state_path_kernel(emu, oracle, golden_dir, monkeypatch):
    """A move's root-ward path at 20 / 61 states: k_pathg == k_walkg bit for bit, both == the oracle."""
    ec.check_general_state_path_kernel(emu, oracle, golden_dir, monkeypatch)


def test_path_and_log_likelihood_in_one_launch(emu, oracle, monkeypatch):
    """k_path4_lnl == k_path4 + k_integrate_lnl_s4 bit for bit (a fixed-topology generation is one launch behind its matrices)."""
    ec.check_fused_path_and_likelihood(emu, oracle, monkeypatch)


@pytest.mark.parametrize("ncat", [4, 1, 5, 16])
def test_paths_that_join_run_as_arms(emu, oracle, monkeypatch, ncat):
    """The lists of topology moves (two dirty branches: two root-ward paths and their common stem) on the path kernel == the same
    lists on the tree-walk kernel, bit for bit; three dirty branches fall back to the walk where the arms nest.  Any category count
    (beyond eight the path runs on k_path4 and the log-likelihood on its own kernel)."""
    ec.check_forked_paths(emu, oracle, monkeypatch, ntaxa=60 if ncat == 4 else 40, npat=700 if ncat == 4 else 200, ncat=ncat)

