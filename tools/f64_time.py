#!/usr/bin/env python3
"""Full-tree evaluations through the double-precision engine (BEAGLE_FLAG_PRECISION_DOUBLE, mbamd_f64.h) next to the fp32
engine on the bench workloads: wall time per evaluation through the Python twin (so both include the same host overhead),
log-likelihoods against the reference's double build.   python tools/f64_time.py [c2 c3 c5 ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrbayes_amd import beagle as bg                         # noqa: E402
from mrbayes_amd import likelihood as lk                     # noqa: E402
from mrbayes_amd.division import division_from_golden        # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    lib = bg.library()
    for cfg in (sys.argv[1:] or ["c2", "c3", "c5"]):
        case = "bench_" + cfg
        with open(os.path.join(GOLD, case + ".json")) as fh:
            ref = json.load(fh)["lnL"]["fp64"]
        div = division_from_golden(GOLD, case)
        row = []
        for dbl, levels in ((False, False), (True, False), (True, True)):
            if levels:                                       # round 2's kernels: no four-state walk, no fp64 matrix cores
                os.environ["MBAMD_F64_NO_WALK"] = "1"        # (read when the instance is created)
                os.environ["MBAMD_F64_NO_MFMA"] = "1"
            else:
                os.environ.pop("MBAMD_F64_NO_WALK", None)
                os.environ.pop("MBAMD_F64_NO_MFMA", None)
            bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_ALWAYS, double_precision=dbl)
            lnl = bd.LogLike(0)
            bd.AcceptMove(0)
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                bd.TouchAllTreeNodes(0)
                bd.LogLike(0)
                bd.AcceptMove(0)
            dt = (time.perf_counter() - t0) / reps
            row.append((lnl, dt))
            bd.finalize()
        nodes = div.tree.n_int_nodes * div.npatterns
        print("%s  %d taxa x %d patterns, %d states x %d categories" % (case, div.tree.ntaxa, div.npatterns, div.nstates, div.ncat))
        print("   reference fp64 build  lnL %.8f" % ref)
        os.environ.pop("MBAMD_F64_NO_WALK", None)
        os.environ.pop("MBAMD_F64_NO_MFMA", None)
        for name, (lnl, dt) in zip(("fp32 engine", "fp64 engine", "fp64 round 2"), row):
            print("   %-12s lnL %.8f  |diff| %.3g (rel %.2g)   %.2f ms per evaluation incl. the Python host  (%.3g node-pattern updates/s)"
                  % (name, lnl, abs(lnl - ref), abs(lnl - ref) / abs(ref), dt * 1e3, nodes / dt))


if __name__ == "__main__":
    main()
