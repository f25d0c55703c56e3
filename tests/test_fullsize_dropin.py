"""The SURVEY 8(f) analyses at bench-like sizes on the MI355X, against goldens the real reference wrote in the build container
(tools/gen_golden_fullsize.py -> tests/golden/fullsize.json, fullsize_anc.npz, parsmodel.json): covarion DNA 200 x 10 000 and covarion
protein 100 x 3 000 (8- / 40-state tree walks at a size that fills the chip), the final pass + ancestral states of three constrained
nodes on the configs[1] shape (500 x 20 000), a BEAGLE v3 multi-partition instance of 2 x (200 x 10 000), and the device Fitch
down-pass on the configs[3] shape (1000 x 50 000, in tests/test_engine_gpu.py::test_parsimony_model_golden).  Both scaling schemes."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import test_mrbayes_dropin as dropin
from tests import test_reports_dropin as rep
from tools import refrun

pytestmark = pytest.mark.gpu
GOLD = os.path.join(refrun.ROOT, "tests", "golden")


def _gold():
    path = os.path.join(GOLD, "fullsize.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/fullsize.json not generated")
    with open(path) as fh:
        return json.load(fh)


@pytest.mark.parametrize("case", ["covarion_dna_200x10000", "covarion_protein_100x3000"])
def test_covarion_at_bench_size(case):
    if not os.path.exists(rep.REF_AMD_REPORTS):
        pytest.skip("oracle/_ref/mb_amd_reports was not built")
    g = _gold()[case]
    for scaling in ("dynamic", "always"):
        out, h, r = rep._row0(rep.REF_AMD_REPORTS, case, scaling)
        assert "mbamd HIP gfx950" in out and "tree-walk" in out and "Non-beagle version" not in out, out[-1500:]
        got = r[h.index("lnLike")]
        assert abs(got - g["lnLike"]) <= rep.TOL_LNL_REL * abs(g["lnLike"]), (case, scaling, got, g["lnLike"])


def test_ancestral_states_on_the_config1_shape():
    if not os.path.exists(rep.REF_AMD_REPORTS):
        pytest.skip("oracle/_ref/mb_amd_reports was not built")
    case = "dna_anc_500x20000"
    g = _gold()[case]
    want = np.load(os.path.join(GOLD, "fullsize_anc.npz"))["row"].astype(np.float64)
    for scaling in ("dynamic", "always"):
        out, h, r = rep._row0(rep.REF_AMD_REPORTS, case, scaling, env={"MBAMD_REPORTS_CHECK": "1"})
        assert "mbamd HIP gfx950" in out and "Non-beagle version" not in out, out[-1500:]
        assert hashlib.sha1("\t".join(h).encode()).hexdigest() == g["header_sha1"] and len(r) == g["columns"] == len(want)
        i = g["lnLike_column"]
        assert abs(r[i] - g["lnLike"]) <= rep.TOL_LNL_REL * abs(g["lnLike"]), (scaling, r[i], g["lnLike"])
        got = np.asarray(r)
        cols = np.array([j for j in range(len(h)) if j != i])
        worst = np.abs(got[cols] - want[cols]).max()
        # state probabilities, fp32 conditional likelihoods on both sides through ~1000 node products (500 taxa): 5e-5 absolute
        # (measured 1.8e-5; the 12-taxon cases of tests/test_reports_dropin.py hold 1e-5), + the float32 the golden row is stored in
        assert worst <= 5e-5, (scaling, worst)
        assert g["state_probability_columns"] > 100000


def test_multi_partition_instance_at_bench_size():
    if not os.path.exists(refrun.REF_MB_AMD_V3):
        pytest.skip("oracle/_ref/mb_amd_v3 was not built")
    g = _gold()["partitioned_2x200x10000"]
    for scaling in ("dynamic", "always"):
        out, _ = refrun.run_mb(refrun.REF_MB_AMD_V3, dropin._partitioned_nexus(scaling, **g["args"]))
        assert "for 2 divisions" in out and "mbamd HIP gfx950" in out, out[-2500:]
        ours = refrun.initial_lnl(out)
        assert abs(ours - g["initial_lnL_native"]) <= 1e-5 * abs(g["initial_lnL_native"]), (scaling, ours, g)
