import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
kind, ntaxa, nsites, ngen = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
st = mbdata.synthetic_states(ntaxa, nsites, {"wag":20,"m3":61}[kind], 7, 0.15, 0.0)
tr = mbtree.random_tree(ntaxa, 3, brlen=0.05)
out, wall = refrun.run_mb(refrun.REF_MB_AMD, refrun.model_nexus(kind, st, tr, ngen=ngen, beagle="dynamic", fixed_topology=True), env={"MBAMD_STATS": "1"})
print("wall", wall)
print("\n".join(l for l in out.splitlines() if "mbamd" in l or "Analysis" in l or "unique site" in l or "Chain 1" in l)[:3000])
