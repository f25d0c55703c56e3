"""Soak test: long default-move MCMC runs of the unmodified MrBayes binary on the engine (topology moves ->
ever-changing operation lists, plan-cache churn, accept/reject buffer flips, dynamic rescaling), checked for
completion and for a final lnL close to what the native kernels reach from the same seed.
usage: [MB_BINARY=...] [MB_PRECISION=double] soak_mcmc.py gtr|wag|m3 ntaxa nsites ngen nchains"""
import os, sys, re
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun

kind, ntaxa, nsites, ngen, nchains = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
nstates = {"gtr": 4, "wag": 20, "m3": 61}[kind]
st = mbdata.synthetic_states(ntaxa, nsites, nstates, 21, 0.15, 0.02)
tr = mbtree.random_tree(ntaxa, 22, brlen=0.05)
if kind == "gtr":
    nex = refrun.mcmc_nexus(st, tr, ngen, beagle="dynamic", nchains=nchains)
else:
    nex = refrun.model_nexus(kind, st, tr, ngen=ngen, beagle="dynamic").replace("nchains=1", "nchains=%d" % nchains)
if os.environ.get("MB_PRECISION") == "double":                      # `set beagleprecision=double`: the fp64 engine
    nex = nex.replace("beagleprecision=single", "beagleprecision=double")
    assert "beagleprecision=double" in nex
binary = os.environ.get("MB_BINARY", refrun.REF_MB_AMD)            # e.g. oracle/_ref/mb_amd_pars: with the device-parsimony binding
out, wall = refrun.run_mb(binary, nex, timeout=3000, env={"MBAMD_STATS": "1"})
ok = "Analysis completed" in out
last = [l for l in out.splitlines() if re.match(r"\s+%d -- " % ngen, l)]
print("completed" if ok else "FAILED", "wall %.1f s" % wall)
print("\n".join(last[:2]))
print("\n".join(l for l in out.splitlines() if "plan cache" in l or "rror" in l)[:600])
if not ok:
    print(out[-2000:])
    sys.exit(1)
