"""Device time of a full-tree evaluation of the 4-state walk under the two rescaling schemes: `always` (every operation writes its
node's exponents: SCALE_WRITE entries) and `dynamic` between rescalings (every operation divides by the exponents stored at the last
rescaling: SCALE_READ entries -- what most evaluations of a real chain with beaglescaling=dynamic are).  usage: scale_read_time.py [case] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from tests.engine_checks import division_from_golden
case = sys.argv[1] if len(sys.argv) > 1 else "bench_c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lib = bg.BeagleLibrary()
div = division_from_golden(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"), case)
for name, scaling in (("always", lk.MB_BEAGLE_SCALE_ALWAYS), ("dynamic", lk.MB_BEAGLE_SCALE_DYNAMIC)):
    bd = lk.BeagleDivision(div, lib, scaling=scaling)
    try:
        first = bd.LogLike(0)
        bd.AcceptMove(0)
        bd.inst.kernel_timing(True)
        for warm in range(2):
            vals = []
            t0 = time.perf_counter()
            for i in range(steps):
                bd.TouchAllTreeNodes(0)
                vals.append(bd.LogLike(0))
                bd.AcceptMove(0)
            wall = (time.perf_counter() - t0) / steps
            kms, kn = bd.inst.get_kernel_timing()
            sms, sn = bd.inst.get_step_timing()
        print("%-8s lnL %.6f (first %.6f)  partials kernel %.4f ms (%d launches)  all kernels %.4f ms  wall per evaluation incl. the python twin %.3f ms  lists %s" %
              (name, vals[-1], first, kms / max(kn, 1), kn, sms / max(sn, 1), wall * 1e3, bd.inst.get_list_counts()))
    finally:
        bd.finalize()
