"""Pin the CPU oracle (oracle/mb_oracle.c + mrbayes_amd.model host numerics) against
log-likelihoods produced by the REAL reference binaries (tests/golden/*.json, written by
tools/gen_golden.py from oracle/_ref/mb, mb_scalar, mb_fp64)."""
import json
import os

import pytest

from mrbayes_amd.division import division_from_golden

SMALL = ["primates_gtr_g4", "primates_gtr_ig4", "primates_gtr_equal", "avian_wag_g4", "replicase_m3",
         "synth_dna_gaps", "synth_aa_wag", "synth_codon_m3"]


def _gold(golden_dir, case):
    with open(os.path.join(golden_dir, case + ".json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("case", SMALL)
def test_oracle_matches_reference(oracle, golden_dir, case):
    g = _gold(golden_dir, case)
    div = division_from_golden(golden_dir, case)
    assert div.npatterns == g["npatterns"]
    lnl = oracle.tree_loglike(div, use_shortcuts=True)
    ref_scalar, ref_fp64, ref_fma = g["lnL"]["scalar"], g["lnL"]["fp64"], g["lnL"]["fma"]
    # the oracle restates the *scalar* kernels: same arithmetic, so it must sit inside the
    # reference's own fp32 self-consistency band (|fma - fp64| is that band on this data set)
    band = max(abs(ref_fma - ref_fp64), abs(ref_scalar - ref_fp64), 1e-7 * abs(ref_fp64))
    assert abs(lnl - ref_scalar) <= 2.0 * band, (lnl, ref_scalar, band)
    assert abs(lnl - ref_fp64) / abs(ref_fp64) < 2e-6
    # dense-tip evaluation (what the SIMD reference kernels do) stays in the same band
    lnl_dense = oracle.tree_loglike(div, use_shortcuts=False)
    assert abs(lnl_dense - ref_fp64) / abs(ref_fp64) < 2e-6


def test_oracle_big_dna(oracle, golden_dir):
    case = "synth_dna_500x20k"
    if not os.path.exists(os.path.join(golden_dir, case + ".json")):
        pytest.skip("fixture not generated")
    g = _gold(golden_dir, case)
    div = division_from_golden(golden_dir, case)
    assert div.npatterns == g["npatterns"]
    lnl = oracle.tree_loglike(div)
    assert abs(lnl - g["lnL"]["fp64"]) / abs(g["lnL"]["fp64"]) < 2e-6
    assert abs(lnl - g["lnL"]["fma"]) / abs(g["lnL"]["fma"]) < 1e-5


def test_parsimony_oracle_against_the_reference_parsimony_model(golden_dir):
    """oracle/pars_oracle.c (GetFitchPartials restated) against golden values from the reference's own Likelihood_Pars
    (tests/golden/parsmodel.json, written by tools/gen_golden_pars.py from oracle/_ref/mb with `lset parsmodel=yes`):
    lnL = -(tree length + number of characters) ln 4, to the printed digits."""
    import math
    import numpy as np
    from mrbayes_amd import data as mbdata
    from mrbayes_amd import parsimony as mp
    from mrbayes_amd import tree as mbtree
    from tests import oracle_lib as ol
    with open(os.path.join(golden_dir, "parsmodel.json")) as fh:
        cases = json.load(fh)["cases"]
    for case in cases:
        st = mbdata.synthetic_states(case["ntaxa"], case["nsites"], case["nstates"], case["seed"], 0.15, case["p_gap"])
        tr = mbtree.random_tree(case["ntaxa"], case["tree_seed"], brlen=0.05)
        sets = np.zeros((tr.n_nodes, st.shape[1]), dtype=np.uint64)
        sets[:st.shape[0]] = mp.tip_sets(st, case["nstates"])
        w = np.ones(st.shape[1], dtype=np.float32)
        total, _ = ol.pars_down(sets, mp.down_pass_ops(tr, tr.root_left), w)
        total += ol.pars_score(sets, [[tr.root_left, -1, tr.root, -1]], w)[0]          # the branch to the calculation root
        lnl = -(total + st.shape[1]) * math.log(case["nstates"])
        # (the reference adds the tree length up in float, src/likelihood.c:7597-7672: beyond 2^24 steps -- the configs[3] shape -- its own
        #  sum is rounded; there the golden pins to the float's resolution, as in tests/engine_checks.py::check_parsimony_model_golden)
        tol = 1e-6 if total < 2 ** 24 else 4e-6 * abs(lnl)
        assert abs(lnl - case["lnL_reference"]) <= tol, (case["name"], lnl, case["lnL_reference"])
