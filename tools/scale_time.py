"""How the roofline fraction of the general-state walk depends on the size of the alignment: full-tree evaluations of synthetic
protein / codon workloads with growing pattern counts (same tree shape as the bench cases), device time of all kernels and of the
partials kernel (HIP events), the issued matrix-core flops and HBM bytes scaled from the PMC counts of the bench case.
usage: scale_time.py wag|m3 ntaxa npat [npat ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import synthetic_division

kind, ntaxa = sys.argv[1], int(sys.argv[2])
base = {"wag": ("c3", 10000, 200), "m3": ("c5", 5000, 100)}[kind]
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")) as fh:
    pmc = json.load(fh)[base[0]]
lib = bg.library()
for npat in [int(x) for x in sys.argv[3:]]:
    div = synthetic_division(kind, ntaxa, npat, seed=7, tree_seed=3)
    bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_ALWAYS)
    bd.LogLike(0); bd.AcceptMove(0)
    evals = [lk.record_evaluation(bd), lk.record_evaluation(bd)]
    for i in range(10):
        evals[i & 1].run()
    bd.inst.kernel_timing(True); bd.inst.get_kernel_timing(reset=True); bd.inst.get_step_timing(reset=True)
    n = 60
    for i in range(n):
        evals[i & 1].run()
    kms, _ = bd.inst.get_kernel_timing(reset=True)
    ams, spans = bd.inst.get_step_timing(reset=True)
    scale = (npat / base[1]) * ((ntaxa - 2) / (base[2] - 2))
    flops = pmc.get("mfma_issued_gflop", 0.0) * 1e9 * scale
    byts = pmc["traffic_bytes"] * scale
    t_all, t_k = ams / max(spans, 1) * 1e-3, kms / n * 1e-3
    print("%s %d x %d: all kernels %.3f ms, partials kernel %.3f ms; issued MFMA %.1f TFLOP/s = %.2f of 157.3 (partials kernel alone %.2f); HBM %.2f TB/s = %.2f of 8"
          % (kind, ntaxa, npat, t_all * 1e3, t_k * 1e3, flops / t_all / 1e12, flops / t_all / 157.3e12, flops / t_k / 157.3e12, byts / t_all / 1e12, byts / t_all / 8e12))
    bd.finalize()
