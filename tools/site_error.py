"""Per-site log-likelihood error of the single-precision engine against the double-precision engine (itself pinned to 1e-11 of the
reference's fp64 build) on the bench workloads: max and RMS over the site patterns, the weighted total, and the reference's own
fp32 builds beside it.  Run once per library (MBAMD_LIBRARY=... selects the one under test):

    python tools/site_error.py c3 c5 [--scaling always|dynamic]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrbayes_amd import beagle as bg                     # noqa: E402
from mrbayes_amd import likelihood as lk                 # noqa: E402
from mrbayes_amd.division import division_from_golden    # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    cases = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c3", "c5"]
    scaling = lk.MB_BEAGLE_SCALE_DYNAMIC if "dynamic" in sys.argv else lk.MB_BEAGLE_SCALE_ALWAYS
    lib = bg.library()
    out = {"library": os.path.relpath(lib.path, ROOT), "cases": {}}
    for c in cases:
        case = "bench_" + c
        with open(os.path.join(GOLD, case + ".json")) as fh:
            gold = json.load(fh)["lnL"]
        div = division_from_golden(GOLD, case)
        vals = {}
        for dp in (True, False):
            bd = lk.BeagleDivision(div, lib, scaling=scaling, double_precision=dp)
            lnl = bd.LogLike(0)
            site = np.array(bd.inst.get_site_log_likelihoods(), dtype=np.float64)
            bd.finalize()
            vals[dp] = (lnl, site)
        d = vals[False][1] - vals[True][1]
        out["cases"][c] = {
            "patterns": int(d.size),
            "site_error_max": float(np.abs(d).max()), "site_error_rms": float(np.sqrt((d * d).mean())), "site_error_mean": float(d.mean()),
            "site_relative_error_rms": float(np.sqrt(((d / vals[True][1]) ** 2).mean())),
            "lnL_fp32_engine": vals[False][0], "lnL_fp64_engine": vals[True][0], "lnL_error": vals[False][0] - vals[True][0],
            "reference_fp64_build": gold["fp64"], "reference_fma_build_error": gold["fma"] - gold["fp64"],
            "reference_scalar_build_error": gold["scalar"] - gold["fp64"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
