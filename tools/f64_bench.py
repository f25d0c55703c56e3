#!/usr/bin/env python3
"""bench.py's double-precision line alone: replayed full-tree evaluations through the fp64 engine, wall time per evaluation and
the log-likelihood's distance from the reference's double build (not asserted here: ablation builds may be measured).
    [MBAMD_LIBRARY=build_x/...so] python tools/f64_bench.py [c4 c2 c5 c3]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                  # noqa: E402
from mrbayes_amd import beagle as bg                         # noqa: E402
from mrbayes_amd import likelihood as lk                     # noqa: E402
from mrbayes_amd.division import division_from_golden        # noqa: E402

if __name__ == "__main__":
    lib = bg.library()
    steps = int(os.environ.get("F64_STEPS", "50"))
    for cfg in (sys.argv[1:] or ["c4", "c5"]):
        case = bench.CONFIGS[cfg][0]
        with open(os.path.join(bench.GOLD, case + ".json")) as fh:
            ref = json.load(fh)["lnL"]["fp64"]
        div = division_from_golden(bench.GOLD, case)
        bd = lk.BeagleDivision(div, lib, nchains=1, scaling=lk.MB_BEAGLE_SCALE_ALWAYS, double_precision=True)
        lnl = bd.LogLike(0)
        bd.AcceptMove(0)
        evals = [lk.record_evaluation(bd), lk.record_evaluation(bd)]
        for i in range(5):
            evals[i & 1].run()
        t0 = time.perf_counter()
        for i in range(steps):
            rc, lnl = evals[i & 1].run()
        dt = time.perf_counter() - t0
        bd.finalize()
        print(json.dumps({"config": cfg, "library": os.path.basename(lib.path), "ms_per_step": round(dt / steps * 1e3, 4), "abs_diff": abs(lnl - ref)}))
