"""Timing experiment: run the unmodified MrBayes binary on the engine for N generations with MBAMD_STATS=1."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
ntaxa, npat, ngen = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scaling = sys.argv[4] if len(sys.argv) > 4 else "dynamic"
fixed = len(sys.argv) > 5 and sys.argv[5] == "fixed"
st = mbdata.synthetic_states(ntaxa, npat, 4, 7, 0.15, 0.0)
tr = mbtree.random_tree(ntaxa, 3, brlen=0.05)
out, wall = refrun.run_mb(refrun.REF_MB_AMD, refrun.mcmc_nexus(st, tr, ngen, beagle=scaling, fixed_topology=fixed), env={"MBAMD_STATS": "1"})
print("wall", wall)
print("\n".join(l for l in out.splitlines() if "mbamd" in l or "Analysis" in l or "rescal" in l.lower())[:4000])
