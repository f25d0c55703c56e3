"""Kernel time of the general-state tree walk per full evaluation, for MBAMD_LIBRARY (a variant built by tools/build_variants.py,
or round 3's library) or the product, under whatever MBAMD_* switches the environment holds: python tools/ablate_walkg.py c3|c5"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import division_from_golden
case = {"c3": "bench_c3", "c5": "bench_c5"}[sys.argv[1] if len(sys.argv) > 1 else "c3"]
div = division_from_golden(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"), case)
lib = bg.library()
bd = lk.BeagleDivision(div, lib, nchains=1, scaling=lk.MB_BEAGLE_SCALE_ALWAYS)
try:
    bd.LogLike(0)
except Exception as e:
    print("first evaluation:", e)
bd.AcceptMove(0)
evals = [lk.record_evaluation(bd), lk.record_evaluation(bd)]
for i in range(5):
    evals[i & 1].run()
bd.inst.kernel_timing(True)
bd.inst.get_kernel_timing(reset=True)
t0 = time.perf_counter()
n = 50
for i in range(n):
    evals[i & 1].run()
dt = time.perf_counter() - t0
kms, kl = bd.inst.get_kernel_timing(reset=True)
print("%s %s: %.4f ms per evaluation, kernel %.4f ms, %d launches" % (case, os.environ.get("MBAMD_LIBRARY", "product"), dt / n * 1e3, kms / n, kl / n))
