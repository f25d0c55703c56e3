import json, os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
with open(os.path.join(bench.GOLD, bench.CONFIGS["c2"][0] + ".json")) as fh:
    gold = json.load(fh)
sy = gold["synthetic"]
st = mbdata.synthetic_states(sy["ntaxa"], sy["nsites"], 4, sy["seed"], sy["p_mut"], sy["p_gap"])
tr = mbtree.parse_newick(gold["newick"])
for ngen in (1000, 6000):
    out, wall = refrun.run_mb(refrun.REF_MB_AMD_PARS, refrun.mcmc_nexus(st, tr, ngen, beagle="dynamic"), env={"MBAMD_STATS": "1"})
    print(ngen, wall)
    print("\n".join(l for l in out.split("\n") if "[mbamd]" in l))
# acceptance table of the moves
print("\n".join(l for l in out.split("\n") if "%" in l and ("Pars" in l or "NNI" in l or "TBR" in l or "SPR" in l or "Multiplier" in l or "Slider" in l or "Dirichlet" in l))[:3000])
