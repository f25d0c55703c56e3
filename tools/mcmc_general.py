"""Timing experiment: whole MCMC generations per second of the UNMODIFIED MrBayes binary for the general-state
models (protein WAG+G4 / codon M3), fixed topology (every move re-evaluates a root-ward path or the whole tree
through the likelihood path), engine (oracle/_ref/mb_amd) vs the same binary's native kernels (oracle/_ref/mb).
Two-point differencing of "Analysis used X seconds of CPU time"/wall so that set-up cancels.
usage: mcmc_general.py wag|m3 ntaxa nsites [engine_lo engine_hi cpu_lo cpu_hi]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun

kind, ntaxa, nsites = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
elo, ehi, clo, chi = (int(x) for x in sys.argv[4:8]) if len(sys.argv) >= 8 else (500, 3000, 10, 40)
nstates = {"wag": 20, "m3": 61}[kind]
st = mbdata.synthetic_states(ntaxa, nsites, nstates, 7, 0.15, 0.0)
tr = mbtree.random_tree(ntaxa, 3, brlen=0.05)


def rate(binary, lo, hi, beagle):
    t = {}
    for n in (lo, hi):
        t0 = time.time()
        out, _ = refrun.run_mb(binary, refrun.model_nexus(kind, st, tr, ngen=n, beagle=beagle, fixed_topology=True))
        assert "Analysis completed" in out, out[-1500:]
        t[n] = time.time() - t0
    return (hi - lo) / (t[hi] - t[lo])


res = {"model": kind, "ntaxa": ntaxa, "nsites": nsites}
if os.path.exists(refrun.REF_MB_AMD):
    res["engine_gen_per_s"] = rate(refrun.REF_MB_AMD, elo, ehi, "dynamic")
if os.path.exists(refrun.REF_MB) and "--no-cpu" not in sys.argv:
    res["cpu_gen_per_s"] = rate(refrun.REF_MB, clo, chi, None)
if "engine_gen_per_s" in res and "cpu_gen_per_s" in res:
    res["ratio"] = res["engine_gen_per_s"] / res["cpu_gen_per_s"]
print(res)
